#!/bin/bash
# One parametrised runner for the GPU jobs of a round (replaces the per-job scripts of rounds 2-3).
#
#   gpurun --timeout 900 -- 'bash tools/gpu_job.sh r4/job3 "t:s2:tests/test_gpu_gemm_s2.py" "b:c4:--model query --arch huge --batch 4 --lora"'
#
# Every step writes into gpurun_out/<out-subdir>/ and prints one status line; a failing step does not stop the next.
#   t:<name>:<pytest args>     python -m pytest -m gpu -q <args>                -> <name>.log      (timeout 1500 s)
#   n:<name>:<pytest args>     the same under RSP_POISON_EMPTY=1 (tests/conftest.py: torch.empty starts as NaN bytes)
#   b:<name>:<bench args>      python bench.py <args>                           -> bench_<name>.json / .err
#   p:<name>:<bench args>      rocprofv3 --kernel-trace --stats -- bench.py ... -> prof_<name>/ + prof_<name>_kernel_stats.csv
#   x:<name>:<command>         bash -c <command>                                -> <name>.log      (timeout 900 s)
#   c:<name>:<command>         rocprofv3 --pmc passes of <command> (SQ groups; counters only, one group per pass)
#   m:<name>:<command>         ... plus the memory-side passes (FETCH_SIZE, WRITE_SIZE, TCC hit / miss)
#                              -> pmc_<name>/ + pmc_<name>_report.{txt,json} (tools/pmc_report.py)
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
    n) eval "RSP_POISON_EMPTY=1 timeout 1500 python -m pytest -m gpu -q $args" > "$O/$name.log" 2>&1; rc=$?
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
       (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$O/prof_$name" -o run -- python "$OLDPWD/bench.py" $args --no-cpu-baseline > "$OLDPWD/$O/prof_$name.json" 2> "$OLDPWD/$O/prof_$name.err"); rc=$?
       f=$(find "$O/prof_$name" -name '*kernel_stats.csv' | head -n 1)
       [ -n "$f" ] && cp "$f" "$O/prof_${name}_kernel_stats.csv" && head -n 6 "$f" | cut -c1-160
       # the trace / database files are large (gpurun copies back at most 64 MiB): keep the summary only
       rm -rf "$O/prof_$name" ;;
    x) timeout 900 bash -c "$args" > "$O/$name.log" 2>&1; rc=$?
       tail -n 4 "$O/$name.log" | cut -c1-300 ;;
    c|m) P="$O/pmc_$name"; rm -rf "$P"; mkdir -p "$P"; rc=0
       pmc_pass() { n=$1; shift
         (cd /tmp && timeout 300 rocprofv3 --pmc "$@" --output-format csv -d "$OLDPWD/$P/$n" -o r -- bash -c "cd $OLDPWD && $args" > "$OLDPWD/$P/$n.log" 2>&1) || rc=$?; }
       pmc_pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU
       pmc_pass sq3 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU
       pmc_pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC
       if [ "$kind" = m ]; then
         pmc_pass fetch FETCH_SIZE
         pmc_pass write WRITE_SIZE
         pmc_pass tcc TCC_HIT_sum TCC_MISS_sum
       fi
       python tools/pmc_report.py "$P" --json "$O/pmc_${name}_report.json" > "$O/pmc_${name}_report.txt" 2>&1
       find "$P" -name "*agent_info.csv" -delete
       find "$P" -type f ! -name "*counter_collection.csv" ! -name "*.log" -delete
       find "$P" -type f -size +4M -delete
       grep -E "^[a-z_A-Z0-9<>, ]+grid=|mfma_busy|pct_of_wave|per_mfma" "$O/pmc_${name}_report.txt" | head -n 40 | cut -c1-150 ;;
    *) echo "unknown step $step"; rc=99 ;;
  esac
  echo "[$kind:$name] rc=$rc $(( $(date +%s) - t0 )) s"
done
