#!/bin/bash
O=gpurun_out/r3; mkdir -p $O
timeout 300 python tools/gemm_s2_group_sweep.py > $O/s2_group24.log 2>&1; echo "rc=$?"; grep -v amdgpu $O/s2_group24.log
