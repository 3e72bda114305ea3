"""debug: first auxiliary mask logits (mask_pred_plus of the initial head call) of the configs[4] fixture, HIP vs oracle"""
import os, sys, warnings
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rsprompter_amd as ra
import rsprompter_amd.debug as dbg
dbg.KEEP_TRACES = True
from oracle import glue
from oracle.query import QueryOracle
from rsprompter_amd.default_configs import rsprompter_query_lora
from rsprompter_amd.structures import DetDataSample
from rsprompter_amd.synth import synth_images, synth_metas, synth_state_dict
dev = torch.device('cuda:0')
arch = sys.argv[1] if len(sys.argv) > 1 else 'base'
oracle = QueryOracle(arch, 1, 100, max_per_image=100, lora=dict(r=16, alpha=32))
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    model = ra.build_model(rsprompter_query_lora(arch, 1, (100, 5)))
sd = synth_state_dict(oracle, seed=2)
oracle.load_state_dict(sd); model.load_state_dict(sd, strict=True); model = model.to(dev)
imgs = synth_images(1, seed=77); metas = synth_metas(1, ori_shape=(512, 512), scale_factor=(2.0, 2.0))
x = glue.data_preprocess(imgs, [123.675, 116.28, 103.53], [58.395, 57.12, 57.375], True, 32)
ref, tr = oracle.predict(x, metas)
out = model.test_step(dict(inputs=[i.to(dev) for i in imgs], data_samples=[DetDataSample(metainfo=dict(m)) for m in metas]))
t = model.panoptic_head._last_trace
for n in range(len(t['mask_pred_plus_all'])):
    a = t['mask_pred_plus_all'][n].detach().cpu().reshape(tr['mask_pred_plus_all'][n].shape)
    b = tr['mask_pred_plus_all'][n]
    e = (a - b).abs()
    i = int(e.argmax())
    idx = [int(v) for v in torch.unravel_index(torch.tensor(i), e.shape)]
    print(f'head call {n}: shape {tuple(a.shape)} max err {float(e.max()):.3e} at {idx}: ours {float(a.flatten()[i]):.5f} ref {float(b.flatten()[i]):.5f}; '
          f'elements off by > 1e-3: {int((e > 1e-3).sum())}; ours finite {bool(torch.isfinite(a).all())}')
    if float(e.max()) > 1e-3:
        bad = (e > 1e-3).nonzero()
        print('   first bad', bad[:5].tolist(), ' last bad', bad[-3:].tolist())
        print('   bad per query (first 10 queries):', [(int(q), int((e[0, q] > 1e-3).sum())) for q in range(10)])
