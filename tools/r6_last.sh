#!/bin/bash
# The round's last gpurun call, at HEAD: decoder loops, the bit-compared soak of whole bench steps, the four
# bench lines with their cpu_baseline legs (+ rocprofv3 statistics of the default one), smoke and, LAST, the plain GPU suite exactly
# as the driver runs it.  `poisoned` as first argument: also the suite with every torch.empty poisoned, in front of the plain one.
#   gpurun --timeout 3000 -- 'bash tools/r6_last.sh [poisoned]'
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
P=(); [ "$1" = poisoned ] && P=("n:suite_poisoned:tests -x")
bash tools/gpu_job.sh r6/last \
  "x:loop_mm:python tools/multimask_loop.py --iters 3000" \
  "x:loop_chain:python tools/multimask_loop.py --iters 3000 --multimask 0 --fused 0" \
  "x:loop_product:python tools/multimask_loop.py --iters 3000 --multimask 0" \
  "x:soak_vith:python tools/determinism_soak.py --arch huge --steps 60" \
  "x:soak_vitb:python tools/determinism_soak.py --arch base --steps 200" \
  "x:soak_query:python tools/determinism_soak.py --arch large --model query --batch 16 --steps 40" \
  "b:config3_anchor_vith_b8:" "p:config3_anchor_vith_b8:" \
  "b:config1_anchor_vitb_b8:--arch base" \
  "b:config2_query_vitl_b16:--model query --arch large --batch 16" \
  "b:config4_query_vith_lora_b4:--model query --arch huge --batch 4 --lora" \
  "x:smoke:python __graft_entry__.py smoke" \
  "${P[@]}" \
  "t:suite:tests -x --durations=25"
