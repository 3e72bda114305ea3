cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 300 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_r1.json 2> gpurun_out/bench_r1.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
find gpurun_out/prof_r1 -name "*kernel_stats.csv" | head -3
tail -c 400 gpurun_out/bench_r1.json
