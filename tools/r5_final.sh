#!/bin/bash
# Round 5 evidence run: the four single-GPU bench lines of BASELINE.json (configs[1], [2], the per-GPU slices of [3] and [4])
# with cpu_baseline, rocprofv3 kernel statistics of the same commands, PMC passes of the dominant GEMM shape per architecture
# (memory-side traffic for bench.py's roofline.traffic) and of the two ViT attention kernels.  Results: gpurun_out/r5/final/
# (copied into profiles/ by hand afterwards).   gpurun --timeout 2400 -- 'bash tools/r5_final.sh'
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r5/final
mkdir -p $O
bash tools/gpu_job.sh r5/final \
  "b:config3_anchor_vith_b8:" "p:config3_anchor_vith_b8:" \
  "b:config1_anchor_vitb_b8:--arch base" "p:config1_anchor_vitb_b8:--arch base" \
  "b:config2_query_vitl_b16:--model query --arch large --batch 16" "p:config2_query_vitl_b16:--model query --arch large --batch 16" \
  "b:config4_query_vith_lora_b4:--model query --arch huge --batch 4 --lora" "p:config4_query_vith_lora_b4:--model query --arch huge --batch 4 --lora" \
  "m:gemm_huge:python tools/pmc_suite.py --arch huge --batch 8 --what gemm --only lin1_gelu_planes" \
  "m:gemm_large:python tools/pmc_suite.py --arch large --batch 16 --what gemm --only lin1_gelu_planes" \
  "m:gemm_base:python tools/pmc_suite.py --arch base --batch 8 --what gemm --only lin1_gelu_planes" \
  "c:attn:python tools/pmc_suite.py --arch huge --batch 8 --what attn"
for a in "huge 8" "large 16" "base 8"; do
  set -- $a
  python tools/pmc_traffic.py $O/pmc_gemm_$1_report.json $1 $2 $O/gemm_traffic_$1.json > /dev/null 2>&1 || echo "no traffic file for $1"
done
python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "[smoke] rc=$? $(tail -n 1 $O/smoke.log)"
