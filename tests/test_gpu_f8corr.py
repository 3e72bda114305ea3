"""-m gpu: the opt-in fp8-corrected product mode (ops.F8_CORR / RSP_F8CORR=1 / bench.py --f8corr) end to end.

The default build runs the three-pass fp16x3 product everywhere and is held to the tight bounds of the other test files.
This file pins what the fast mode delivers instead: the same 1e-3 budget on embeddings and mask
logits with about 8x less margin, detections that move only where oracle scores tie within the (larger) score error."""
import os
import sys
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _match import match_detections  # noqa: E402

MEAN = [123.675, 116.28, 103.53]
STD = [58.395, 57.12, 57.375]


def _maxerr(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())


def test_encoder_f8corr_within_budget(dev, monkeypatch):
    from oracle import hf_sam
    from rsprompter_amd import ops
    from rsprompter_amd.sam_encoder import RSSamVisionEncoder
    from rsprompter_amd.synth import synth_state_dict
    monkeypatch.setattr(ops, 'F8_CORR', True)
    m = RSSamVisionEncoder('sam_vit_base', extra_config=dict(output_hidden_states=True))
    sd = synth_state_dict(m.vision_encoder, seed=0)
    m.vision_encoder.load_state_dict(sd)
    o = hf_sam.build_vision_encoder('base')
    o.load_state_dict(sd, strict=True)
    x = torch.randn(1, 3, 1024, 1024, generator=torch.Generator().manual_seed(11))
    emb_ref, hs_ref = hf_sam.run_vision_encoder(o, x)
    m = m.to(dev)
    prof = ops.Profiler()
    ops.set_profiler(prof)
    out = m(x.to(dev))
    torch.cuda.synchronize()
    ops.set_profiler(None)
    names = set(prof.summary())
    assert m.vision_encoder._packed['f8'] and any(n.startswith('gemm_f16f8_dma_kernel') for n in names), names
    e_hs = max(_maxerr(h, r) for h, r in zip(out[1], hs_ref))
    e_emb = _maxerr(out[0], emb_ref)
    print('f8corr ViT-B: hidden-state err %.2e, embedding err %.2e' % (e_hs, e_emb))
    assert e_hs < 1e-3 and e_emb < 5e-4          # measured 3.6e-4 / 1.0e-4 (fp16x3: 4e-5 / 1.5e-5)


def test_anchor_f8corr_end_to_end(dev, monkeypatch):
    import rsprompter_amd as ra
    from oracle import glue
    from oracle.anchor import AnchorOracle
    from rsprompter_amd import ops
    from rsprompter_amd.default_configs import rsprompter_anchor
    from rsprompter_amd.structures import DetDataSample
    from rsprompter_amd.synth import synth_images, synth_metas, synth_state_dict
    monkeypatch.setattr(ops, 'F8_CORR', True)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = ra.build_model(rsprompter_anchor('base', 10))
    from _oracle_cache import anchor_base_two_tiles              # the oracle run of tests/test_gpu_anchor.py's fixture
    oracle, sd, imgs, metas, x, ref, tr = anchor_base_two_tiles()
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    out = model.test_step(dict(inputs=[i.to(dev) for i in imgs], data_samples=[DetDataSample(metainfo=dict(m)) for m in metas]))
    assert model.backbone.vision_encoder._packed['f8']
    low = model.roi_head._last_mask_trace['mask_preds'].cpu()
    ours0 = ref0 = 0
    for b in range(2):
        pi, r = out[b].pred_instances, ref[b]
        k = r['labels'].shape[0]
        # scores carry the 1e-4 feature error: entries within 5e-4 of a neighbour / the cut may swap or straddle it
        pairs = match_detections(pi.bboxes, pi.scores, pi.labels, r['bboxes'], r['scores'], r['labels'], tie=5e-4,
                                 max_odd=12, score_tol=5e-4)
        ii = torch.tensor([i for i, _ in pairs]); jj = torch.tensor([j for _, j in pairs])
        e_low = _maxerr(low[ours0 + ii], tr['low_res_masks'][ref0 + jj])
        mism = float((pi.masks.cpu()[ii] != r['masks'][jj]).float().mean())
        print(f'f8corr anchor ViT-B img {b}: {k} dets, {len(pairs)} matched, low_res_masks err {e_low:.2e}, '
              f'mask pixel mismatch {mism:.2e}')
        assert len(pairs) >= k - 12 and e_low < 1e-3 and mism < 1e-3
        ours0 += pi.labels.shape[0]
        ref0 += k
    e_emb = _maxerr(model._last_embeddings, tr['image_embeddings'])
    print(f'f8corr anchor ViT-B: image embedding err {e_emb:.2e}')
    assert e_emb < 5e-4
