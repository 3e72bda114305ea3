#!/bin/bash
# rel-pos of the windowed layers for the real tokens only, 4 query groups per block; q fill of padded rows dropped
O=gpurun_out/r3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "vit_attention" > $O/attn_test23.log 2>&1; echo "tests rc=$?"; tail -3 $O/attn_test23.log
timeout 900 python -m pytest tests/test_gpu_encoder.py -m gpu -q > $O/enc_test23.log 2>&1; echo "encoder tests rc=$?"; tail -2 $O/enc_test23.log
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench23.json 2> $O/bench23.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/bench23.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['parity_canary']['image_embedding_max_abs_err'], d['parity_canary']['mask_logit_max_abs_err'], d['parity_canary']['ok'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms'])[:8]:
    print(f"{k:45s} {v['ms']:8.3f} ms {v['calls']:5d} calls  {v.get('tflops')} TF")
print({k:v['ms'] for k,v in d['kernels'].items() if 'fill' in k or 'relpos' in k})
PY
