cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 100 python tools/attn_micro.py 8 64 12 64 5 > gpurun_out/attn_micro.log 2>&1
timeout 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_attn1 -- python tools/attn_micro.py 8 64 12 64 3 > gpurun_out/pmc_attn1.log 2>&1
timeout 120 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_attn2 -- python tools/attn_micro.py 8 64 12 64 3 > gpurun_out/pmc_attn2.log 2>&1
cat gpurun_out/attn_micro.log
