# PMC passes for the attention kernels only (round 2): bash tools/pmc_attn.sh <tag>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-attn}
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
timeout 200 python tools/pmc_suite.py --what attn > $OUT/manifest.jsonl 2> $OUT/manifest.err
pass() { n=$1; shift; timeout 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$n -o r -- python tools/pmc_suite.py --iters 2 --what attn > $OUT/$n.log 2>&1; echo "pass $n rc=$?"; }
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass sq3 SQ_INSTS_VALU_MFMA_F16 SQ_INSTS_MFMA SQ_WAVES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU
cat $OUT/manifest.jsonl
