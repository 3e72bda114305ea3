#!/bin/bash
# Round 5, GPU call 3: where the ping-pong GEMM's time goes -- development build: ablations, per-tile time stamps, PMC
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r5/gemm_pp_ablate
mkdir -p $O
export RSP_DEV_BUILD=1
t0=$(date +%s)
timeout 300 python tools/gemm_pp_exp.py ablate huge > $O/pp_ablate.txt 2>&1; echo "[ablate] rc=$? $(( $(date +%s) - t0 )) s"; cat $O/pp_ablate.txt | cut -c1-600
bash tools/gpu_job.sh r5/gemm_pp_ablate "m:pp:python tools/gemm_pp_exp.py huge 8"
echo "[done] $(( $(date +%s) - t0 )) s"
