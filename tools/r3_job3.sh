#!/bin/bash
# round-3 GPU job 3: gemm_s2 with specialised epilogues + tile tickets: bit-exactness, timing, time stamps
cd "$(dirname "$0")/.."
O=gpurun_out/r3
mkdir -p $O
timeout 900 python tools/gemm_s2_exp.py both > $O/s2_exp4.log 2>&1
timeout 300 python tools/gemm_s2_exp.py trace > $O/s2_trace4.log 2>&1
grep -c OK $O/s2_exp4.log; grep -E "FAIL|ALL|SOME" $O/s2_exp4.log; tail -8 $O/s2_exp4.log
