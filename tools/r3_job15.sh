#!/bin/bash
# qkv scatter form on the s2 kernel: bit-exactness vs the gather form, timing, encoder parity, bench
O=gpurun_out/r3; mkdir -p $O
timeout 600 python tools/gemm_s2_exp.py check > $O/s2_check15.log 2>&1; echo "check rc=$?"; grep -c "^OK" $O/s2_check15.log; grep -v "^OK" $O/s2_check15.log | tail -8
timeout 600 python tools/gemm_s2_exp.py time > $O/s2_time15.log 2>&1; echo "time rc=$?"; head -8 $O/s2_time15.log
timeout 900 python -m pytest tests/test_gpu_encoder.py -x -q -m gpu > $O/enc_tests15.log 2>&1; echo "encoder tests rc=$?"; tail -2 $O/enc_tests15.log
timeout 600 python bench.py --steps 6 --warmup 2 > $O/bench15.json 2> $O/bench15.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/bench15.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['parity_canary']['image_embedding_max_abs_err'], d['parity_canary']['ok'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms'])[:8]:
    print(f"{k:45s} {v['ms']:8.3f} ms {v['calls']:5d} calls  {v.get('tflops')} TF")
PY
