#!/bin/bash
# round-3 GPU job 2: what bounds the GEMM main loops (DMA issue / LDS write port / memory latency) -- ablations, time
# stamps of the persistent blocks, PMC passes (counters only, no trace domains)
cd "$(dirname "$0")/.."
R=$PWD
O=gpurun_out/r3
mkdir -p $O
timeout 600 python tools/gemm_s2_exp.py time > $O/s2_exp2.log 2>&1
timeout 300 python tools/gemm_s2_exp.py trace > $O/s2_trace.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd $R
pass() { n=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_gemm/$n -o r -- python tools/gemm_pmc_micro.py 2 > $O/pmc_gemm_$n.log 2>&1
  echo "pass $n rc=$?"; }
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
pass sq2 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE
pass ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_BUFFER_READ_LDS_WAVEFRONTS_sum GRBM_GUI_ACTIVE
pass tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
pass tcc TCC_HIT_sum TCC_MISS_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
tail -3 $O/s2_exp2.log
