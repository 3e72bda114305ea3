#!/bin/bash
# window attention without the padded queries: kernel test, encoder parity, bench
O=gpurun_out/r3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "vit_attention" > $O/attn_test21.log 2>&1; echo "tests rc=$?"; tail -3 $O/attn_test21.log
timeout 900 python -m pytest tests/test_gpu_encoder.py -m gpu -q > $O/enc_test21.log 2>&1; echo "encoder tests rc=$?"; tail -2 $O/enc_test21.log
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench21.json 2> $O/bench21.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/bench21.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['parity_canary']['image_embedding_max_abs_err'], d['parity_canary']['mask_logit_max_abs_err'], d['parity_canary']['ok'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms'])[:10]:
    print(f"{k:45s} {v['ms']:8.3f} ms {v['calls']:5d} calls  {v.get('tflops')} TF")
PY
