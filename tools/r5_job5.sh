#!/bin/bash
# Round 5, GPU call 5: the 128 x 256 tile of the ping-pong GEMM: parity tests, micro-benchmark on the three architectures, step
# A/B, then (development build) ablations and tile time stamps
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r5/job5
mkdir -p $O
t0=$(date +%s)
timeout 300 python -m pytest -m gpu -q -x tests/test_gpu_gemm_s2.py > $O/gemm_tests.log 2>&1
echo "[gemm tests] rc=$? $(( $(date +%s) - t0 )) s: $(tail -n 1 $O/gemm_tests.log)"; grep -E "Error|assert|FAILED" $O/gemm_tests.log | head -n 10
for a in "huge 8" "base 8" "large 16"; do
  n=${a%% *}
  timeout 300 python tools/gemm_pp_exp.py $a > $O/pp_$n.txt 2>&1; echo "[pp $n] rc=$? $(( $(date +%s) - t0 )) s"; grep -v amdgpu.ids $O/pp_$n.txt | cut -c1-420
done
timeout 400 python tools/ab_bench.py --kernels --steps 4 --rounds 3 "s2:ops.PP_AUTO=False" "pp:ops.PP_AUTO=True" > $O/ab_huge.txt 2> $O/ab_huge.err
echo "[ab huge] rc=$? $(( $(date +%s) - t0 )) s"; cat $O/ab_huge.txt | cut -c1-220; tail -n 3 $O/ab_huge.err
timeout 400 python tools/ab_bench.py --arch base --kernels --steps 6 --rounds 3 "s2:ops.PP_AUTO=False" "pp:ops.PP_AUTO=True" > $O/ab_base.txt 2> $O/ab_base.err
echo "[ab base] rc=$? $(( $(date +%s) - t0 )) s"; cat $O/ab_base.txt | cut -c1-220; tail -n 3 $O/ab_base.err
export RSP_DEV_BUILD=1
timeout 600 python -m rsprompter_amd.build > $O/dev_build.log 2>&1; echo "[dev build] rc=$? $(( $(date +%s) - t0 )) s"
timeout 300 python tools/gemm_pp_exp.py ablate huge > $O/pp_ablate.txt 2>&1; echo "[ablate] rc=$? $(( $(date +%s) - t0 )) s"; grep -v amdgpu.ids $O/pp_ablate.txt | cut -c1-700
echo "[done] $(( $(date +%s) - t0 )) s"
