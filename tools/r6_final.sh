#!/bin/bash
# Round 6 closing run on ONE box: the whole GPU suite at HEAD, the multimask loops (third box of DESIGN 9.1), the four
# single-GPU bench lines of BASELINE.json with cpu_baseline, rocprofv3 kernel statistics of the same commands, the PMC passes of
# the dominant GEMM shape (memory-side traffic for bench.py's roofline.traffic) and of the attention kernels, smoke.
# Results: gpurun_out/r6/final/ (copied into profiles/ by hand afterwards).   gpurun --timeout 3300 -- 'bash tools/r6_final.sh'
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r6/final
mkdir -p $O
bash tools/gpu_job.sh r6/final \
  "t:suite:tests --durations=30" \
  "x:loop_mm:python tools/multimask_loop.py --iters 3000" \
  "x:loop_chain:python tools/multimask_loop.py --iters 3000 --multimask 0 --fused 0" \
  "b:config3_anchor_vith_b8:" "p:config3_anchor_vith_b8:" \
  "b:config1_anchor_vitb_b8:--arch base" "p:config1_anchor_vitb_b8:--arch base" \
  "b:config2_query_vitl_b16:--model query --arch large --batch 16" "p:config2_query_vitl_b16:--model query --arch large --batch 16" \
  "b:config4_query_vith_lora_b4:--model query --arch huge --batch 4 --lora" "p:config4_query_vith_lora_b4:--model query --arch huge --batch 4 --lora" \
  "m:gemm_huge:python tools/pmc_suite.py --arch huge --batch 8 --what gemm --only lin1_gelu_planes" \
  "c:attn:python tools/pmc_suite.py --arch huge --batch 8 --what attn"
python tools/pmc_traffic.py $O/pmc_gemm_huge_report.json huge 8 $O/gemm_traffic_huge.json > /dev/null 2>&1 || echo "no traffic file for huge"
python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "[smoke] rc=$? $(tail -n 1 $O/smoke.log)"
