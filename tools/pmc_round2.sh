# Round-2 PMC evidence for the dominant kernels on the ViT-H shapes.  One counter group per pass
# (SQ has 8 slots, FETCH_SIZE / WRITE_SIZE do not share a pass), counters only -- no trace domains.
#   usage: bash tools/pmc_round2.sh <tag> [extra pmc_suite args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
TAG=${1:-r2}; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
rocprofv3 -L > $OUT/counters_available.txt 2>&1
timeout 300 python tools/pmc_suite.py "$@" > $OUT/manifest.jsonl 2> $OUT/manifest.err
pass() {  # name counters...
  n=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$n -o r -- python tools/pmc_suite.py --iters 2 "${EXTRA[@]}" > $OUT/$n.log 2>&1
  echo "pass $n rc=$?"
}
EXTRA=("$@")
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass sq3 SQ_INSTS_VALU_MFMA_F16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_LDS_UNALIGNED_STALL SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum
find $OUT -name "*counter_collection.csv" | head -20
cat $OUT/manifest.jsonl
