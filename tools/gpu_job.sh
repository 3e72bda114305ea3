#!/bin/bash
# One parametrised runner for the GPU jobs of a round (replaces the per-job scripts of rounds 2-3).
#
#   gpurun --timeout 900 -- 'bash tools/gpu_job.sh r4/job3 "t:s2:tests/test_gpu_gemm_s2.py" "b:c4:--model query --arch huge --batch 4 --lora"'
#
# Every step writes into gpurun_out/<out-subdir>/ and prints one status line; a failing step does not stop the next.
#   t:<name>:<pytest args>     python -m pytest -m gpu -q <args>                -> <name>.log      (timeout 1500 s)
#   b:<name>:<bench args>      python bench.py <args>                           -> bench_<name>.json / .err
#   p:<name>:<bench args>      rocprofv3 --kernel-trace --stats -- bench.py ... -> prof_<name>/ + prof_<name>_kernel_stats.csv
#   x:<name>:<command>         bash -c <command>                                -> <name>.log      (timeout 900 s)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/$1; shift
mkdir -p "$O"
export TMPDIR=/tmp
for step in "$@"; do
  kind=${step%%:*}; rest=${step#*:}; name=${rest%%:*}; args=${rest#*:}
  t0=$(date +%s)
  case $kind in
    t) eval "timeout 1500 python -m pytest -m gpu -q $args" > "$O/$name.log" 2>&1; rc=$?
       tail -n 3 "$O/$name.log" | tr '\n' ' ' ;;
    b) timeout 900 python bench.py $args > "$O/bench_$name.json" 2> "$O/bench_$name.err"; rc=$?
       python - "$O/bench_$name.json" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c = r.get('parity_canary') or {}
    print(f"{r['value']} {r['unit']}, {r['ms_per_step']} ms/step; roofline {r['roofline']['kernel']} {r['roofline']['achieved']} TF/s "
          f"frac {r['roofline']['frac']}; attention {r['roofline_attention']['achieved']} TF/s; canary ok={c.get('ok')} "
          f"emb {c.get('image_embedding_max_abs_err')} logits {c.get('mask_logit_max_abs_err')}; cpu {(r.get('cpu_baseline') or {}).get('value')}", end=' ')
except Exception as e:
    print('no bench line:', e, end=' ')
PY
       ;;
    p) rm -rf "$O/prof_$name"
       (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$O/prof_$name" -o run -- python "$OLDPWD/bench.py" $args --no-cpu-baseline > "$OLDPWD/$O/prof_$name.json" 2> "$OLDPWD/$O/prof_$name.err"); rc=$?
       f=$(find "$O/prof_$name" -name '*kernel_stats.csv' | head -n 1)
       [ -n "$f" ] && cp "$f" "$O/prof_${name}_kernel_stats.csv" && head -n 6 "$f" | cut -c1-160
       # the trace itself is large: keep the summary only
       find "$O/prof_$name" -name '*kernel_trace.csv' -delete ;;
    x) timeout 900 bash -c "$args" > "$O/$name.log" 2>&1; rc=$?
       tail -n 4 "$O/$name.log" | cut -c1-300 ;;
    *) echo "unknown step $step"; rc=99 ;;
  esac
  echo "[$kind:$name] rc=$rc $(( $(date +%s) - t0 )) s"
done
