#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3; mkdir -p $O
timeout 600 python tools/gemm_epi_ablate.py > $O/epi_ablate7.log 2>&1
cat $O/epi_ablate7.log
