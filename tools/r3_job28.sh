#!/bin/bash
# window attention as two 4-wave blocks per (window, head) (RSP_ATTN_WIN4=1) against the 7-wave block
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3; mkdir -p $O
RSP_ATTN_WIN4=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "vit_attention" > $O/attn_test28.log 2>&1; echo "tests (4-wave) rc=$?"; tail -2 $O/attn_test28.log
python - <<'PY' > $O/attn_time28.log 2>&1
import os, sys, torch
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
for (Bp, nh, dh) in ((200, 16, 80), (200, 12, 64)):
    S, D = 14, nh * dh
    q = torch.randn(Bp * 196, D, device=dev)
    kv = ops.to_planes(torch.randn(Bp * 196, 2 * D, device=dev))
    rel = torch.randn(Bp * nh, 196, 28, device=dev) * 0.1
    for grid in (None, (5, 8)):
        res = {}
        for tag, env in (('7-wave', None), ('2x4-wave', '1')):
            if env: os.environ['RSP_ATTN_WIN4'] = env
            else: os.environ.pop('RSP_ATTN_WIN4', None)
            out = ops.vit_attention_planes(q, kv, rel, Bp, S, nh, dh, dh ** -0.5, planes=True, win_grid=grid)
            res[tag] = (timed(lambda: ops.vit_attention_planes(q, kv, rel, Bp, S, nh, dh, dh ** -0.5, planes=True, win_grid=grid)), out)
        os.environ.pop('RSP_ATTN_WIN4', None)
        same = torch.equal(res['7-wave'][1].hi, res['2x4-wave'][1].hi) if grid is None else 'n/a'
        print(f'nh={nh} dh={dh} grid={grid}: 7-wave {res["7-wave"][0]:.3f} ms, 2x4-wave {res["2x4-wave"][0]:.3f} ms, identical={same}')
PY
grep -v amdgpu $O/attn_time28.log
