#!/bin/bash
# round-3 GPU job 8: gemm_s2 with the per-wave LDS-transposed epilogue: check, time, ablation, trace
cd "$(dirname "$0")/.."
O=gpurun_out/r3; mkdir -p $O
timeout 600 python tools/gemm_s2_exp.py check > $O/s2_check8.log 2>&1
grep -c "^OK" $O/s2_check8.log; grep -E "FAIL|ALL|SOME|part|fault" $O/s2_check8.log | head -20
timeout 600 python tools/gemm_s2_exp.py time > $O/s2_time8.log 2>&1
tail -8 $O/s2_time8.log
timeout 300 python tools/gemm_epi_ablate.py > $O/epi_ablate8.log 2>&1
cat $O/epi_ablate8.log
timeout 300 python tools/gemm_s2_exp.py trace > $O/s2_trace8.log 2>&1
grep -E "tiles|histogram" $O/s2_trace8.log
