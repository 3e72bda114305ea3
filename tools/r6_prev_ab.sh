#!/bin/bash
# Round 6: the library of the working tree against rsprompter_amd/librsp_hip_prev.so (the same sources with SOME files taken from
# an earlier commit, built by hand before the call; not tracked) on ONE box: bench lines new / prev / new, per-kernel table.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r6/${1:-prev_ab}
mkdir -p $O
L=rsprompter_amd/librsp_hip.so
run() {
  n=$1; shift
  timeout 400 python bench.py --steps 8 --warmup 2 --no-cpu-baseline "$@" > $O/bench_$n.json 2> $O/bench_$n.err
  python - $O/bench_$n.json $n <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = j['kernels']
pick = lambda s: sum(v['ms'] for n, v in k.items() if s in n)
c = j['parity_canary']
print(f"[{sys.argv[2]}] {j['ms_per_step']:.2f} ms/step  gemm_pp {pick('gemm_f16x3_pp'):.2f}  attn global {pick('attn_stream'):.2f} window {pick('attn_win'):.2f}  "
      f"upscale {pick('sam_upscale_fused'):.2f}  i2t {pick('sam_i2t_fused'):.2f}  t2i_fold {pick('sam_t2i_fold'):.2f}  t2i {pick('sam_t2i_kernel'):.2f}  LN {pick('layernorm'):.2f}  "
      f"dma gemms {pick('gemm_f16x3_dma'):.2f}  canary ok={c['ok']} emb {c['image_embedding_max_abs_err']:.2e} logits {c['mask_logit_max_abs_err']:.2e} idx {c.get('indices_equal')}")
PY
}
run new1
cp $L /tmp/new.so; cp rsprompter_amd/librsp_hip_prev.so $L
run prev
cp /tmp/new.so $L
run new2
run new2_q --model query --arch large --batch 16
