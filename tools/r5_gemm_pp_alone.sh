#!/bin/bash
# Round 5, GPU call 6: staggered block starts of the ping-pong GEMM (epilogue bursts against HBM write bandwidth), tile time stamps
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r5/gemm_pp_alone
mkdir -p $O
export RSP_DEV_BUILD=1
t0=$(date +%s)
timeout 600 python -m rsprompter_amd.build > $O/dev_build.log 2>&1; echo "[dev build] rc=$? $(( $(date +%s) - t0 )) s"
timeout 400 python tools/gemm_pp_exp.py ablate huge alone > $O/pp_alone.txt 2>&1; echo "[ablate] rc=$? $(( $(date +%s) - t0 )) s"; grep -v amdgpu.ids $O/pp_alone.txt | cut -c1-900
echo "[done] $(( $(date +%s) - t0 )) s"
