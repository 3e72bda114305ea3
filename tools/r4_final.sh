#!/bin/bash
# Round-4 evidence run at HEAD, one gpurun call (about 35 GPU-minutes):
#   whole GPU suite, smoke(), the four bench lines of BASELINE.json's single-GPU configurations (with roofline, parity
#   canary and cpu_baseline), rocprofv3 --kernel-trace --stats of the same commands, the --pmc passes of the dominant
#   GEMM shape (SQ groups + FETCH / WRITE / TCC) and of the two ViT attention kernels.
# usage (GPU box): bash tools/r4_final.sh [nosuite]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
J=r4/final
STEPS=()
if [ "$1" != "nosuite" ]; then
  STEPS+=("t:gpu_suite:tests")
  STEPS+=("x:smoke:python -c 'import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")'")
fi
STEPS+=("b:config3_anchor_vith_b8:"
        "b:config1_anchor_vitb_b8:--arch base"
        "b:config2_query_vitl_b16:--model query --arch large --batch 16"
        "b:config4_query_vith_lora_b4:--model query --arch huge --batch 4 --lora"
        "p:config3_anchor_vith_b8:--steps 3 --warmup 1"
        "p:config2_query_vitl_b16:--model query --arch large --batch 16 --steps 3 --warmup 1"
        "p:config4_query_vith_lora_b4:--model query --arch huge --batch 4 --lora --steps 3 --warmup 1"
        "p:config1_anchor_vitb_b8:--arch base --steps 3 --warmup 1"
        "m:gemm_lin1:python tools/pmc_suite.py --what gemm --only lin1_gelu_planes --iters 2"
        "c:attn:python tools/pmc_suite.py --what attn --iters 2")
bash tools/gpu_job.sh $J "${STEPS[@]}"
