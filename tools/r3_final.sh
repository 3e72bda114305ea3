# round-3 evidence run at HEAD: default bench line (with the CPU baseline), rocprofv3 kernel statistics of the same
# command, and the PMC passes of the dominant GEMM (gemm_f16x3_s2_kernel on the lin1 shape: one counter group per pass,
# counters only).  usage (GPU box): bash tools/r3_final.sh [nopmc]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_final
mkdir -p $O
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof -o r -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/rocprof_bench.json 2> $O/rocprof.err; echo "rocprof rc=$?"
find $O/rocprof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/rocprof
if [ "$1" != "nopmc" ]; then
  P=$O/pmc; mkdir -p $P
  for shape in lin1_gelu_planes; do
    timeout 300 python tools/pmc_suite.py --what gemm --only $shape > $P/manifest_$shape.jsonl 2> $P/manifest_$shape.err
    pass() { n=$1; shift
      timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $P/$shape/$n -o r -- python tools/pmc_suite.py --what gemm --only $shape --iters 2 > $P/${shape}_$n.log 2>&1
      echo "pass $shape $n rc=$?"; }
    pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU
    pass sq3 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
    pass fetch FETCH_SIZE
    pass write WRITE_SIZE
    pass tcc TCC_HIT_sum TCC_MISS_sum
    python tools/pmc_report.py $P/$shape --json $P/report_$shape.json > $P/report_$shape.txt 2>&1
    find $P/$shape -name "*agent_info.csv" -delete
  done
fi
if [ "$1" != "nopmc" ]; then
  # the two ViT attention kernels (+ rel-pos), same counters
  P=$O/pmc; mkdir -p $P
  timeout 300 python tools/pmc_suite.py --what attn > $P/manifest_attn.jsonl 2> $P/manifest_attn.err
  for n in sq1 sq3; do
    if [ $n = sq1 ]; then C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"; else C="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU"; fi
    timeout 300 rocprofv3 --pmc $C --output-format csv -d $P/attn/$n -o r -- python tools/pmc_suite.py --what attn --iters 2 > $P/attn_$n.log 2>&1
    echo "pass attn $n rc=$?"
  done
  python tools/pmc_report.py $P/attn --json $P/report_attn.json > $P/report_attn.txt 2>&1
  find $P/attn -name "*agent_info.csv" -delete
fi
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_final/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d.get('cpu_baseline'))
PY
tail -30 gpurun_out/r3_final/pmc/report_lin1_gelu_planes.txt
