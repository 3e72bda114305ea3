#!/bin/bash
# round-3 GPU job 1: new two-blocks-per-CU GEMM: bit-exactness vs the round-2 kernels, timing on the ViT-H shapes,
# and the round-2 tree's bench line on this box as the baseline of the round
mkdir -p gpurun_out/r3
cd "$(dirname "$0")/.."
timeout 900 python tools/gemm_s2_exp.py both > gpurun_out/r3/s2_exp1.log 2>&1
echo "exit $?" >> gpurun_out/r3/s2_exp1.log
timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/r3/bench_base.json 2> gpurun_out/r3/bench_base.err
tail -5 gpurun_out/r3/s2_exp1.log
