# HBM traffic of the dominant GEMM (ViT-B lin1 shape of the bench workload) from the L2 memory-side counters.
# Separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass); no trace domains besides the counters.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_')
  timeout 120 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_$n -- python tools/gemm_micro.py 32768 3072 768 0 6 > gpurun_out/pmc_$n.log 2>&1
done
find gpurun_out -name "*counter_collection.csv" | head
