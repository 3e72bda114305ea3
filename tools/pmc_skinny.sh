cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_sk1 -- python tools/gemm_micro.py 3276800 256 256 0 3 > gpurun_out/pmc_sk1.log 2>&1
timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sk2 -- python tools/gemm_micro.py 3276800 256 256 0 3 > gpurun_out/pmc_sk2.log 2>&1
