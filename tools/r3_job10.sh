#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3; mkdir -p $O
timeout 600 python tools/debug_c4.py base > $O/debug_c4.log 2>&1
cat $O/debug_c4.log | tail -20
RSP_GEMM_RULE=r2 timeout 600 python tools/debug_c4.py base > $O/debug_c4_r2.log 2>&1
tail -8 $O/debug_c4_r2.log
