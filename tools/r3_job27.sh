#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3; mkdir -p $O
timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench27_dev.json 2> $O/bench27_dev.err; echo "rc=$?"
timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --host-inputs > $O/bench27_host.json 2> $O/bench27_host.err; echo "rc=$?"
python - <<'PY'
import json
for f in ('bench27_dev', 'bench27_host'):
    d=json.loads(open(f'gpurun_out/r3/{f}.json').read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['data'])
PY
