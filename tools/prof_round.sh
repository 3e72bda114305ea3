# round evidence: the official bench line (with cpu_baseline), the rocprofv3 kernel stats of the same command,
# and the other architectures' bench lines
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_r1.json 2> gpurun_out/bench_r1.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
timeout 200 python bench.py --arch large --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_r1_large.json 2>/dev/null
timeout 200 python bench.py --arch huge --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_r1_huge.json 2>/dev/null
find gpurun_out/prof_r1 -name "*kernel_stats.csv" | head -3
