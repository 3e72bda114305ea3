#!/bin/bash
# Round 4, last GPU call (4 minutes of budget): the kernels written after the closing whole-suite run -- box coder
# branches, NMS above 16384 candidates, MSDeformAttn level counts, the folded token -> image attention -- and one A/B of the
# bench with that attention folded / projected.  Every step has its own timeout; results land in gpurun_out/r4/last/.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r4/last
mkdir -p $O
t0=$(date +%s)
timeout 90 python -m pytest -m gpu -q -s tests/test_gpu_kernels.py tests/test_gpu_samdet.py tests/test_gpu_query.py tests/test_gpu_baseline_configs.py \
  -k "t2i_fold or batched_nms or box_coder or many_classes or level_counts or folded_token" > $O/new_kernels.log 2>&1
echo "[new kernels] rc=$? $(( $(date +%s) - t0 )) s: $(tail -n 1 $O/new_kernels.log)"
RSP_T2I_FOLD=1 timeout 60 python bench.py --no-cpu-baseline > $O/bench_fold_on.json 2> $O/bench_fold_on.err
echo "[bench fold on] rc=$? $(( $(date +%s) - t0 )) s: $(python -c "import json,sys; r=json.loads(open('$O/bench_fold_on.json').read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], (r.get('parity_canary') or {}).get('ok'), {k: v['ms'] for k, v in r['kernels'].items() if 't2i' in k or 'i2t' in k})" 2>&1 | tail -n 1)"
timeout 60 python bench.py --no-cpu-baseline > $O/bench_fold_off.json 2> $O/bench_fold_off.err
echo "[bench fold off] rc=$? $(( $(date +%s) - t0 )) s: $(python -c "import json,sys; r=json.loads(open('$O/bench_fold_off.json').read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], (r.get('parity_canary') or {}).get('ok'), {k: v['ms'] for k, v in r['kernels'].items() if 't2i' in k or 'i2t' in k})" 2>&1 | tail -n 1)"
RSP_T2I_FOLD=1 timeout 80 python -m pytest -m gpu -q tests/test_gpu_anchor.py > $O/anchor_fold_on.log 2>&1
echo "[anchor suite, fold on] rc=$? $(( $(date +%s) - t0 )) s: $(tail -n 1 $O/anchor_fold_on.log)"
timeout 70 python -m pytest -m gpu -q tests/test_gpu_baseline_configs.py -k "option_branches and levels" > $O/levels.log 2>&1
echo "[query head level counts] rc=$? $(( $(date +%s) - t0 )) s: $(tail -n 1 $O/levels.log)"
