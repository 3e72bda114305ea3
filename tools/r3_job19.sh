#!/bin/bash
# s2 GEMM: result stores with the non-temporal hint (variant 1) against the product kernel, same process
O=gpurun_out/r3; mkdir -p $O
timeout 600 python tools/gemm_s2_exp.py check > $O/s2_check19.log 2>&1; echo "check rc=$?"; grep -c "^OK" $O/s2_check19.log; grep -v "^OK" $O/s2_check19.log | tail -5
timeout 600 python tools/gemm_s2_exp.py time > $O/s2_time19.log 2>&1; echo "time rc=$?"; grep -v amdgpu $O/s2_time19.log | cut -c1-420
