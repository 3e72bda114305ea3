#!/bin/bash
# rel-pos terms of the global layers on the matrix cores: tests + timing against the FMA kernel
O=gpurun_out/r3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "vit_attention" > $O/relpos_test20.log 2>&1; echo "tests rc=$?"; tail -3 $O/relpos_test20.log
python - <<'PY' > $O/relpos_time20.log 2>&1
import os, torch, sys
sys.path.insert(0, '.')
from rsprompter_amd import ops
dev = torch.device('cuda:0')
def timed(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (Bp, S, nh, dh) in ((8, 64, 16, 80), (8, 64, 12, 64), (8, 32, 12, 64)):
    D = nh * dh
    q = torch.randn(Bp * S * S, D, device=dev)
    rph, rpw = torch.randn(2 * S - 1, dh, device=dev) * 0.05, torch.randn(2 * S - 1, dh, device=dev) * 0.05
    os.environ.pop('RSP_RELPOS_FMA', None)
    a = ops.vit_relpos(q, rph, rpw, Bp, S, nh, dh, q_ld=D)
    t_m = timed(lambda: ops.vit_relpos(q, rph, rpw, Bp, S, nh, dh, q_ld=D))
    os.environ['RSP_RELPOS_FMA'] = '1'
    b = ops.vit_relpos(q, rph, rpw, Bp, S, nh, dh, q_ld=D)
    t_f = timed(lambda: ops.vit_relpos(q, rph, rpw, Bp, S, nh, dh, q_ld=D))
    os.environ.pop('RSP_RELPOS_FMA', None)
    gb = (q.numel() + a.numel()) * 4 / 1e9
    print(f'Bp={Bp} S={S} nh={nh} dh={dh}: MFMA {t_m:.3f} ms ({gb / t_m * 1e3:.0f} GB/s), FMA {t_f:.3f} ms; max |diff| {float((a - b).abs().max()):.2e} (range {float(b.abs().max()):.2f})')
PY
grep -v amdgpu $O/relpos_time20.log
