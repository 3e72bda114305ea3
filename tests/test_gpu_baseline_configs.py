"""-m gpu: end-to-end parity on the BASELINE.json configurations beyond configs[1] (which tests/test_gpu_anchor.py covers):

  configs[2]  rsprompter_query, SAM ViT-L, Nq = 100, WHU-shape metas (512 px tiles resized x2)
  configs[3]  rsprompter_anchor, SAM ViT-H (per-GPU slice, batch > 1)
  configs[4]  rsprompter_query, SAM ViT-H + LoRA(qkv, r16, alpha32), Nq = 100, WHU-shape
  configs/rsprompter/rsprompter_query-nwpu-peft-512.py (ViTSAM at 512 px + LoRA + PseudoFeatureAggregator)
  the encoder at batch 8 (window row maps with B > 1)

Each test runs the FREE-RUNNING HIP pipeline (`test_step`) against the CPU oracle on identical seeded weights / inputs
and asserts, besides the detection lists, the mask LOGITS of the free-running pass (north star: <= 1e-3):
query path -> `mask_pred` of every query; anchor path -> `low_res_masks` of the detections matched to the oracle's.
Reference: models.py:148-170 (anchor predict), :249-272 (query predict), :633-715 (head predict + fusion)."""
import os as _os
import sys as _sys
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu

_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _match import match_detections  # noqa: E402

MEAN = [123.675, 116.28, 103.53]
STD = [58.395, 57.12, 57.375]
LOGIT_TOL = 1e-3            # BASELINE.json north_star: mask logits within 1e-3 (fp32)


def _maxerr(a, b):
    return float((a.detach().float().cpu() - b.detach().float().cpu()).abs().max())


def _build(cfg, oracle, dev, seed=0):
    import rsprompter_amd as ra
    from rsprompter_amd.synth import synth_state_dict
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = ra.build_model(cfg)
    sd = synth_state_dict(oracle, seed=seed)
    oracle.load_state_dict(sd)
    res = model.load_state_dict(sd, strict=True)       # the oracle's (= the reference's) key layout loads unchanged
    assert not res.missing_keys and not res.unexpected_keys
    return model.to(dev)


def _samples(metas):
    from rsprompter_amd.structures import DetDataSample
    return [DetDataSample(metainfo=dict(m)) for m in metas]


def _check_query(model, oracle, imgs, metas, dev, tag):
    from oracle import glue
    x = glue.data_preprocess(imgs, MEAN, STD, True, 32)
    ref, tr = oracle.predict(x, metas)
    out = model.test_step(dict(inputs=[i.to(dev) for i in imgs], data_samples=_samples(metas)))
    # ---- free-running logits of EVERY query (no selection involved): SAM mask logits and class logits
    cls, lazy = model._last_head_out
    e_mask = _maxerr(lazy.low_res, tr['mask_pred'])
    e_cls = _maxerr(cls, tr['cls_pred'])
    rng = float(tr['mask_pred'].abs().max())
    print(f'{tag}: free-running SAM mask logits err {e_mask:.2e} (range {rng:.1f}), class logits err {e_cls:.2e}')
    # per-query view + the discrete decisions upstream of the logits (models.py:381-392: attn_mask = sigmoid < 0.5)
    per_q = (lazy.low_res.detach().float().cpu() - tr['mask_pred']).abs().flatten(2).amax(2)          # [B, Nq]
    flips, flipped_q = [], torch.zeros_like(per_q, dtype=torch.bool)
    for a, b in zip(model.panoptic_head._last_trace['attn_masks'], tr['attn_masks']):
        nq = a.shape[-2]
        ref_m = b.view(len(imgs), -1, nq, b.shape[-1])[:, 0]
        diff = a.cpu().bool().view_as(ref_m) != ref_m
        flips.append(int(diff.sum()))
        flipped_q |= diff.any(-1)
    top = per_q.flatten().topk(5).values.tolist()
    print(f'{tag}: attention-mask bits that differ from the oracle per decoder layer: {flips} '
          f'({int(flipped_q.sum())} queries touched); 5 largest per-query logit errors: {["%.2e" % v for v in top]}; '
          f'median {float(per_q.median()):.2e}')
    # The masked decoder thresholds its auxiliary masks (sigmoid < 0.5 <=> logit < 0, models.py:390): a logit within the
    # round-off of 0 lands on the other side -- a DISCRETE difference like a score tie in the anchor path.  How many:
    # Nq x keys x layers = 100 x ~4300 x 6 = 2.6 M decisions per image, logits spread over +-5 (density ~0.1 per unit near
    # 0), auxiliary-mask error ~2e-5 -> an expectation of ~5 differing bits (measured 4 and 5 on two runs of the ViT-H +
    # LoRA fixture; 0-1 on the others).  The bound is the 99.9 % quantile of that Poisson count, not the observed value.
    # A query whose attention mask differs in some layer attends one more / one fewer key, which moves its logits by more
    # than round-off: such queries are held to 1e-2, every other query to the 1e-3 budget.
    assert int(flipped_q.sum()) <= 12 and sum(flips) <= 14
    assert float(per_q[~flipped_q].max()) < LOGIT_TOL and e_cls < LOGIT_TOL
    assert float(per_q.max()) < 1e-2
    for b in range(len(imgs)):
        pi, r = out[b].pred_instances, ref[b]
        assert pi.masks.dtype == torch.bool and tuple(pi.masks.shape) == tuple(r['masks'].shape)
        assert tuple(pi.masks.shape[1:]) == tuple(metas[b]['ori_shape'][:2])
        same = pi.query_indices.cpu().long() == r['query_indices']
        sc = r['scores']
        if not bool(same.all()):
            # an entry may only differ where the oracle's own score ties with a neighbour / the cut-off to fp32 noise
            d = (sc[:, None] - sc[None, :]).abs() + torch.eye(len(sc)) * 1e9
            near_tie = (d.min(1).values < 5e-5) | ((sc - float(sc.min())).abs() < 5e-5)
            assert bool(near_tie[~same].all()) and int((~same).sum()) <= 4, f'{int((~same).sum())} query indices differ'
        assert torch.equal(pi.labels.cpu()[same], r['labels'][same])
        mism = float((pi.masks.cpu()[same] != r['masks'][same]).float().mean())
        e_sc = _maxerr(pi.scores[same.to(pi.scores.device)], sc[same])
        print(f'{tag} img {b}: {int(same.sum())}/{len(sc)} query indices equal, score err {e_sc:.2e}, '
              f'mask pixel mismatch {mism:.2e}')
        assert e_sc < 1e-4 and mism < 1e-3


def test_config2_query_vitl_nq100_whu(dev):
    """BASELINE.json configs[2] tree (rsprompter_query-whu.py with the large ids): 2 tiles, Nq = 100 -> 200 prompt sets
    through the two-way decoder; WHU metas: ori_shape 512, scale_factor 2 (second resize of the logits)."""
    from oracle.query import QueryOracle
    from rsprompter_amd.default_configs import rsprompter_query
    from rsprompter_amd.synth import synth_images, synth_metas
    oracle = QueryOracle('large', 1, 100, max_per_image=100)
    model = _build(rsprompter_query('large', 1, (100, 5)), oracle, dev)
    imgs = synth_images(2)
    metas = synth_metas(2, ori_shape=(512, 512), scale_factor=(2.0, 2.0))
    _check_query(model, oracle, imgs, metas, dev, 'configs[2] query ViT-L')


def test_config4_query_vith_lora_nq100_whu(dev):
    """BASELINE.json configs[4]: query path on ViT-H + LoRA adapters (non-zero A and B), Nq = 100, WHU-shape."""
    from oracle.query import QueryOracle
    from rsprompter_amd.default_configs import rsprompter_query_lora
    from rsprompter_amd.synth import synth_images, synth_metas
    oracle = QueryOracle('huge', 1, 100, max_per_image=100, lora=dict(r=16, alpha=32))
    model = _build(rsprompter_query_lora('huge', 1, (100, 5)), oracle, dev, seed=2)
    assert any('lora_B.default' in k for k in model.state_dict())
    imgs = synth_images(1, seed=77)
    metas = synth_metas(1, ori_shape=(512, 512), scale_factor=(2.0, 2.0))
    _check_query(model, oracle, imgs, metas, dev, 'configs[4] query ViT-H+LoRA')


def test_query_nwpu_peft512_config(dev):
    """configs/rsprompter/rsprompter_query-nwpu-peft-512.py: 512-px ViTSAM + LoRA + PseudoFeatureAggregator, 10 classes,
    70 queries, 2 tiles."""
    from oracle.query import QueryOracle
    from rsprompter_amd.default_configs import rsprompter_query_peft512
    from rsprompter_amd.synth import synth_images, synth_metas
    oracle = QueryOracle('base', 10, 70, max_per_image=70, peft512=True)
    model = _build(rsprompter_query_peft512('base', 10, (70, 5)), oracle, dev, seed=3)
    imgs, metas = synth_images(2, size=(512, 512)), synth_metas(2, size=(512, 512))
    _check_query(model, oracle, imgs, metas, dev, 'query peft-512')


def test_config3_anchor_vith_batch2(dev):
    """BASELINE.json configs[3] per-GPU slice (rsprompter_anchor, SAM ViT-H; `_base_/rsprompter_anchor.py` defaults are
    huge): 2 tiles free-running; detections matched to the oracle's, then the LOW-RES MASK LOGITS of the matched
    instances compared (the anchor path's logits depend on which boxes were detected, hence the matching)."""
    from oracle import glue
    from oracle.anchor import AnchorOracle
    from rsprompter_amd.default_configs import rsprompter_anchor
    from rsprompter_amd.synth import synth_images, synth_metas
    B = 2
    oracle = AnchorOracle('huge', 10)
    model = _build(rsprompter_anchor('huge', 10), oracle, dev)
    imgs, metas = synth_images(B), synth_metas(B)
    x = glue.data_preprocess(imgs, MEAN, STD, True, 32)
    ref, tr = oracle.predict(x, metas)
    out = model.test_step(dict(inputs=[i.to(dev) for i in imgs], data_samples=_samples(metas)))
    low = model.roi_head._last_mask_trace['mask_preds'].cpu()                # [sum k, 1, 256, 256], image-major
    assert low.shape[0] == sum(o.pred_instances.labels.shape[0] for o in out)
    ours0 = ref0 = 0
    worst = 0.0
    for b in range(B):
        pi, r = out[b].pred_instances, ref[b]
        k = r['labels'].shape[0]
        assert pi.labels.shape[0] == k and tuple(pi.masks.shape[1:]) == (1024, 1024)
        pairs = match_detections(pi.bboxes, pi.scores, pi.labels, r['bboxes'], r['scores'], r['labels'])
        ii = torch.tensor([i for i, _ in pairs]); jj = torch.tensor([j for _, j in pairs])
        e_low = _maxerr(low[ours0 + ii], tr['low_res_masks'][ref0 + jj])
        mism = float((pi.masks.cpu()[ii] != r['masks'][jj]).float().mean())
        print(f'configs[3] anchor ViT-H img {b}: {k} dets, {len(pairs)} matched, low_res_masks err {e_low:.2e} '
              f'(range {float(tr["low_res_masks"].abs().max()):.1f}), mask pixel mismatch {mism:.2e}')
        assert e_low < LOGIT_TOL and mism < 1e-3
        worst = max(worst, e_low)
        ours0 += pi.labels.shape[0]
        ref0 += k
    e_emb = _maxerr(model._last_embeddings, tr['image_embeddings'])
    print(f'configs[3]: image embedding err {e_emb:.2e}, worst matched mask-logit err {worst:.2e}')
    assert e_emb < LOGIT_TOL


def test_encoder_batch8_row_maps(dev):
    """window partition / unpartition row maps with B = 8 (the bench batch): every image of the batch must equal the
    oracle's single-image forward of that image (HF:900-952; images are independent)."""
    from oracle import hf_sam
    from rsprompter_amd.sam_encoder import RSSamVisionEncoder
    from rsprompter_amd.synth import synth_state_dict
    m = RSSamVisionEncoder('sam_vit_base', extra_config=dict(output_hidden_states=True))
    sd = synth_state_dict(m.vision_encoder, seed=0)
    m.vision_encoder.load_state_dict(sd)
    o = hf_sam.build_vision_encoder('base')
    o.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(8, 3, 1024, 1024, generator=g)
    out = m.to(dev)(x.to(dev))
    emb, hs = out[0].cpu(), [h.cpu() for h in out[1]]
    for b in (0, 3, 7):
        emb_ref, hs_ref = hf_sam.run_vision_encoder(o, x[b:b + 1])
        e = max(float((h[b:b + 1] - r).abs().max()) for h, r in zip(hs, hs_ref))
        e_emb = float((emb[b:b + 1] - emb_ref).abs().max())
        print(f'batch-8 encoder, image {b}: hidden-state err {e:.2e}, embedding err {e_emb:.2e}')
        assert e < LOGIT_TOL and e_emb < LOGIT_TOL
