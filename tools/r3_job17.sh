#!/bin/bash
# s2 GEMM: next tile's ticket drawn behind the last DMA (atomic latency under the last three steps)
O=gpurun_out/r3; mkdir -p $O
timeout 600 python tools/gemm_s2_exp.py check > $O/s2_check17.log 2>&1; echo "check rc=$?"; grep -c "^OK" $O/s2_check17.log; grep -v "^OK" $O/s2_check17.log | tail -5
timeout 600 python tools/gemm_s2_exp.py time > $O/s2_time17.log 2>&1; echo "time rc=$?"; grep -v amdgpu $O/s2_time17.log | cut -c1-400
timeout 600 python bench.py --steps 6 --warmup 2 > $O/bench17.json 2> $O/bench17.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/bench17.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['parity_canary']['image_embedding_max_abs_err'], d['parity_canary']['ok'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms'])[:8]:
    print(f"{k:45s} {v['ms']:8.3f} ms {v['calls']:5d} calls  {v.get('tflops')} TF")
PY
