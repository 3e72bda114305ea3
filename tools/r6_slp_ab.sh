#!/bin/bash
# Round 6: the library built with -fno-slp-vectorize (rsprompter_amd/librsp_hip_noslp.so, built by hand before the call; not
# tracked) against the default build on ONE box: bench lines new / noslp / new with the per-kernel table.  Why: the one
# kernel that ever produced a wrong answer (sam_upscale2_kernel) did so only in the builds whose sums
# hipcc had SLP-packed across independent accumulations; MI355X_MICROARCH.md calls compiler-packed fp32 beside MFMAs an
# anti-lever.  Results: gpurun_out/r6/slp/
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r6/slp
mkdir -p $O
L=rsprompter_amd/librsp_hip.so
run() {  # name, bench args
  n=$1; shift
  timeout 400 python bench.py --steps 8 --warmup 2 --no-cpu-baseline "$@" > $O/bench_$n.json 2> $O/bench_$n.err
  python - $O/bench_$n.json $n <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = j['kernels']
pick = lambda s: sum(v['ms'] for n, v in k.items() if s in n)
c = j['parity_canary']
print(f"[{sys.argv[2]}] {j['ms_per_step']:.2f} ms/step  gemm_pp {pick('gemm_f16x3_pp'):.2f}  attn global {pick('attn_stream'):.2f} window {pick('attn_win'):.2f}  "
      f"upscale {pick('sam_upscale_fused'):.2f}  i2t {pick('sam_i2t_fused'):.2f}  t2i_fold {pick('sam_t2i_fold'):.2f}  LN {pick('layernorm'):.2f}  "
      f"dma gemms {pick('gemm_f16x3_dma'):.2f}  canary ok={c['ok']} emb {c['image_embedding_max_abs_err']:.2e} logits {c['mask_logit_max_abs_err']:.2e}")
PY
}
run new1
run new1_b --arch base
cp $L /tmp/new.so
if [ -f rsprompter_amd/librsp_hip_prev.so ]; then   # the same sources with the parent commit's upscale.hip (fused upscaler before the round-6 rework)
  cp rsprompter_amd/librsp_hip_prev.so $L
  run prev_upscaler
fi
cp rsprompter_amd/librsp_hip_noslp.so $L
run noslp
run noslp_b --arch base
timeout 300 python -m pytest -m "gpu and quick" -q -x tests > $O/quick_noslp.log 2>&1
echo "[quick tier, noslp] rc=$? $(tail -n 1 $O/quick_noslp.log)"
cp /tmp/new.so $L
run new2
run new2_b --arch base
