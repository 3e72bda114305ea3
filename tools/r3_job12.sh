#!/bin/bash
# round-3 GPU job 12: MFMA form of rsp_sam_i2t_fused: unit test vs the fp64 composition, timing against the VALU form
cd "$(dirname "$0")/.."
O=gpurun_out/r3; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "i2t" -s > $O/i2t_test12.log 2>&1
grep -E "passed|failed|err|Error" $O/i2t_test12.log | tail -12
timeout 300 python tools/i2t_micro.py > $O/i2t_micro12.log 2>&1
RSP_I2T_VALU=1 timeout 300 python tools/i2t_micro.py >> $O/i2t_micro12.log 2>&1
grep -v amdgpu $O/i2t_micro12.log
