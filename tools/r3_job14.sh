#!/bin/bash
# windowed layers without padded GEMM rows (qkv scatters, proj gathers): encoder parity + bench
O=gpurun_out/r3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_kernels.py -x -q -m gpu > $O/enc_tests14.log 2>&1; echo "encoder+kernel tests rc=$?"
tail -4 $O/enc_tests14.log
timeout 600 python bench.py --steps 6 --warmup 2 > $O/bench14.json 2> $O/bench14.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/bench14.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d.get('parity_canary'))
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms'])[:14]:
    print(f"{k:45s} {v['ms']:8.3f} ms {v['calls']:5d} calls  {v.get('tflops')} TF")
PY
