#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_final; mkdir -p $O
timeout 600 python bench.py > $O/bench_default2.json 2> $O/bench_default2.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_final/bench_default2.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'])
PY
