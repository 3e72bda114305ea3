#!/bin/bash
# Round 5, closing call: the four bench lines + rocprofv3 kernel statistics again after the round's last code changes (RPN
# top-k scan, aggregator branches on side streams); the PMC passes of tools/r5_final.sh are not repeated (the GEMM and
# attention kernels did not change).  Results: gpurun_out/r5/final2/
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out/r5/final2
bash tools/gpu_job.sh r5/final2 \
  "b:config3_anchor_vith_b8:" "p:config3_anchor_vith_b8:" \
  "b:config1_anchor_vitb_b8:--arch base" "p:config1_anchor_vitb_b8:--arch base" \
  "b:config2_query_vitl_b16:--model query --arch large --batch 16" "p:config2_query_vitl_b16:--model query --arch large --batch 16" \
  "b:config4_query_vith_lora_b4:--model query --arch huge --batch 4 --lora" "p:config4_query_vith_lora_b4:--model query --arch huge --batch 4 --lora"
python __graft_entry__.py smoke > gpurun_out/r5/final2/smoke.log 2>&1; echo "[smoke] rc=$? $(tail -n 1 gpurun_out/r5/final2/smoke.log)"
