#!/bin/bash
# Round 5, GPU call 2: the ping-pong GEMM's first run on an MI355X: parity tests, micro-benchmark against gemm_s2 on the
# encoder shapes, A/B of the whole step.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r5/gemm_pp_first
mkdir -p $O
t0=$(date +%s)
timeout 300 python -m pytest -m gpu -q -x tests/test_gpu_gemm_s2.py > $O/gemm_tests.log 2>&1
echo "[gemm tests] rc=$? $(( $(date +%s) - t0 )) s: $(tail -n 1 $O/gemm_tests.log)"; grep -E "Error|assert|FAILED" $O/gemm_tests.log | head -n 10
timeout 300 python tools/gemm_pp_exp.py huge 8 > $O/pp_huge.txt 2>&1; echo "[pp huge] rc=$? $(( $(date +%s) - t0 )) s"; cat $O/pp_huge.txt | cut -c1-400
timeout 300 python tools/gemm_pp_exp.py base 8 > $O/pp_base.txt 2>&1; echo "[pp base] rc=$? $(( $(date +%s) - t0 )) s"; cat $O/pp_base.txt | cut -c1-400
timeout 300 python tools/gemm_pp_exp.py large 16 > $O/pp_large.txt 2>&1; echo "[pp large] rc=$? $(( $(date +%s) - t0 )) s"; cat $O/pp_large.txt | cut -c1-400
timeout 400 python tools/ab_bench.py --kernels --steps 4 --rounds 3 "s2:ops.PP_AUTO=False" "pp:ops.PP_AUTO=True" > $O/ab_huge.txt 2> $O/ab_huge.err
echo "[ab huge] rc=$? $(( $(date +%s) - t0 )) s"; cat $O/ab_huge.txt | cut -c1-220; tail -n 3 $O/ab_huge.err
echo "[done] $(( $(date +%s) - t0 )) s"
