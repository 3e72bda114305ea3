#!/bin/bash
# MFMA form of rsp_sam_i2t_fused, second arrangement (wave = channel block in the product phase): test + timing
cd "$(dirname "$0")/.."
O=gpurun_out/r3; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "i2t" -s > $O/i2t_test16.log 2>&1
grep -E "passed|failed|err|Error" $O/i2t_test16.log | tail -12
RSP_I2T_MFMA=1 timeout 300 python tools/i2t_micro.py > $O/i2t_micro16.log 2>&1
RSP_I2T_MFMA=1 timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench16.json 2> $O/bench16.err; python -c "import json; d=json.loads(open('gpurun_out/r3/bench16.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['parity_canary']); print({k:v for k,v in d['kernels'].items() if 'i2t' in k})"
timeout 300 python tools/i2t_micro.py >> $O/i2t_micro16.log 2>&1
grep -v amdgpu $O/i2t_micro16.log
