# round-2 evidence run: default bench line (with the CPU baseline), rocprofv3 kernel statistics of the same command,
# the per-shape kernel table and the opt-in --f8corr line.  usage (GPU box): bash tools/r2_final.sh [pmc]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_final
mkdir -p $O
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --shapes > $O/bench_shapes.json 2> $O/bench_shapes.err
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --f8corr > $O/bench_f8corr.json 2> $O/bench_f8corr.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof -o r -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/rocprof_bench.json 2> $O/rocprof.err
find $O/rocprof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/rocprof
if [ "$1" = "pmc" ]; then
  bash tools/pmc_round2.sh r2_final > $O/pmc.log 2>&1
  tail -3 $O/pmc.log
fi
