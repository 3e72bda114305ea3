#!/bin/bash
# round-3 GPU job 9: whole GPU suite with gemm_s2 as the product GEMM + the round's new tests; bench line
cd "$(dirname "$0")/.."
O=gpurun_out/r3; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=900 > $O/suite9.log 2>&1
tail -15 $O/suite9.log
timeout 400 python bench.py --no-cpu-baseline --steps 10 --warmup 2 > $O/bench9.json 2> $O/bench9.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/bench9.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'], d['parity_canary'])
for k,v in list(d['kernels'].items())[:14]: print(k, v)
PY
