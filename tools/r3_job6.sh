#!/bin/bash
# round-3 GPU job 6: gemm_s2 after the store-data fix + epilogue priority: check, time, trace
cd "$(dirname "$0")/.."
O=gpurun_out/r3
mkdir -p $O
timeout 600 python tools/gemm_s2_exp.py check > $O/s2_check6.log 2>&1
grep -c "^OK" $O/s2_check6.log; grep -E "FAIL|ALL|SOME|part|fault" $O/s2_check6.log | head -20
timeout 600 python tools/gemm_s2_exp.py time > $O/s2_time6.log 2>&1
tail -8 $O/s2_time6.log
timeout 300 python tools/gemm_s2_exp.py trace > $O/s2_trace6.log 2>&1
grep -E "tiles|histogram" $O/s2_trace6.log
