#!/bin/bash
# Round 6: the round-5 sam_upscale2_kernel -- the form that gives rare wrong quarter-wave sums -- with s_nops behind
# ONE instruction class per variant (tools/probes/up2_isa_bisect.py builds rsprompter_amd/variants/librsp_hip_<v>.so here in
# the container), each looped on the failing chain on ONE box.  base first and last: the box's own failure rate.
#   gpurun --timeout 1500 -- 'bash tools/r6_isa_bisect.sh 1500 base pk acc trans mov mfma valu lds vmem base'
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r6/isa_bisect; mkdir -p $O
N=$1; shift
cp rsprompter_amd/librsp_hip.so /tmp/librsp_hip_product.so
i=0
for v in "$@"; do
  i=$((i+1))
  cp rsprompter_amd/variants/librsp_hip_$v.so rsprompter_amd/librsp_hip.so
  timeout 600 python tools/multimask_loop.py --iters $N > $O/${i}_$v.log 2>&1
  echo "[$i $v] rc=$? $(tail -n 1 $O/${i}_$v.log | cut -c1-220)"
done
cp /tmp/librsp_hip_product.so rsprompter_amd/librsp_hip.so
