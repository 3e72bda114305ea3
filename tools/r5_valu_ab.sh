#!/bin/bash
# Round 5: the VALU diet of the epilogues / softmax loops (new GELU form, packed fp32 operations, v_fma_mix_f32 remainders)
# against the previous library on ONE box: rsprompter_amd/librsp_hip_prev.so is the library built from the parent commit
# (built by hand before the call; not tracked).  Also the FP16_OVFL probe.  Results: gpurun_out/r5/valu/
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r5/valu
mkdir -p $O
t0=$(date +%s)
L=rsprompter_amd/librsp_hip.so
hipcc --offload-arch=gfx950 -O2 -o /tmp/f16_ovfl_probe tools/probes/f16_ovfl_probe.hip > /dev/null 2>&1 && /tmp/f16_ovfl_probe > $O/f16_ovfl_probe.txt 2>&1
echo "[probe] rc=$? $(( $(date +%s) - t0 )) s"; cat $O/f16_ovfl_probe.txt
timeout 400 python -m pytest -m "gpu and quick" -q -x tests > $O/quick.log 2>&1
echo "[quick tier] rc=$? $(( $(date +%s) - t0 )) s: $(tail -n 1 $O/quick.log)"; grep -E "Error|assert|FAILED" $O/quick.log | head -n 10
run() {  # name
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --shapes > $O/bench_$1.json 2> $O/bench_$1.err
  python - $O/bench_$1.json $1 <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = j['kernels']
pick = lambda s: sum(v['ms'] for n, v in k.items() if s in n)
print(f"[{sys.argv[2]}] {j['ms_per_step']:.2f} ms/step  pp lin1 {pick('N=5120 K=1280'):.2f}  lin2 {pick('N=1280 K=5120'):.2f}  qkv {pick('N=3840 K=1280'):.2f}  proj {pick('pp_kernel<256x256> M=32768 N=1280 K=1280'):.2f}"
      f"  attn global {pick('attn_stream'):.2f} window {pick('attn_win'):.2f}  upscale {pick('sam_upscale_fused'):.2f}  i2t {pick('sam_i2t_fused'):.2f}  t2i_fold {pick('sam_t2i_fold'):.2f}  LN {pick('layernorm'):.2f}"
      f"  canary emb {j['parity_canary']['image_embedding_max_abs_err']:.2e} logits {j['parity_canary']['mask_logit_max_abs_err']:.2e}")
PY
}
run new1
cp $L /tmp/new.so; cp rsprompter_amd/librsp_hip_prev.so $L
run old
cp /tmp/new.so $L
run new2
echo "[done] $(( $(date +%s) - t0 )) s"
