"""-m gpu: the SAMSeg sibling models `SAMSegMaskRCNN` / `SAMSegMask2Former` (models.py:1219-1274) on the HIP
kernels against its CPU oracle (oracle/samseg.py, mask branch pinned on the real fcn_mask_head.py)."""
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


def test_paste_masks_kernel_matches_reference_vectors(dev):
    """rsp_paste_masks against the outputs of the real FCNMaskHead._predict_by_feat_single (golden)."""
    from rsprompter_amd import ops
    g = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_vectors_samseg.pt'),
                   weights_only=False)
    for c in g['predict_single']:
        meta = c['meta']
        h, w = c['masks'].shape[-2:]
        got = ops.paste_masks(c['logits'].permute(0, 2, 3, 1).contiguous().to(dev), c['labels'].to(dev),
                              c['boxes_out'].to(dev), (h, w), 0.5)
        mism = float((got.cpu() != c['masks']).float().mean())
        print(f'paste {meta}: mismatch {mism:.2e}')
        assert mism < 1e-4          # pixels whose pasted probability sits within fp32 noise of the threshold
        # mask_thr_binary < 0 (fcn_mask_head.py:390-394): the probabilities as uint8, against the real file's output
        soft = ops.paste_masks(c['logits'].permute(0, 2, 3, 1).contiguous().to(dev), c['labels'].to(dev),
                               c['boxes_out'].to(dev), (h, w), -1.0).cpu()
        assert soft.dtype == torch.uint8 and soft.shape == c['masks_soft'].shape
        d = (soft.int() - c['masks_soft'].int()).abs()
        print(f'soft paste {meta}: {float((d != 0).float().mean()):.2e} of the bytes differ, max {int(d.max())}')
        assert int(d.max()) <= 1 and float((d != 0).float().mean()) < 2e-3   # p * 255 within fp32 noise of an integer


def test_samseg_maskrcnn_end_to_end(dev):
    import rsprompter_amd as ra
    from oracle import glue
    from oracle.samseg import SAMSegMaskRCNNOracle
    from rsprompter_amd.default_configs import samseg_maskrcnn
    from rsprompter_amd.structures import DetDataSample
    from rsprompter_amd.synth import synth_images, synth_metas, synth_state_dict
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = ra.build_model(samseg_maskrcnn('base', 10))
    oracle = SAMSegMaskRCNNOracle('base', 10)
    sd = synth_state_dict(oracle, seed=0)
    oracle.load_state_dict(sd)
    res = model.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    model = model.to(dev)
    imgs = synth_images(2)
    metas = synth_metas(2, ori_shape=(512, 512), scale_factor=(2.0, 2.0))
    x = glue.data_preprocess(imgs, MEAN, STD, True, 32)
    ref, tr = oracle.predict(x, metas)
    # mask logits on the oracle's RoI features (stage test), then the free-running pipeline
    lg = model.roi_head.mask_head(tr['mask_feats'].to(dev).contiguous(memory_format=torch.channels_last))
    e = float((lg.cpu() - tr['mask_logits']).abs().max())
    print('FCN mask head logits err %.2e (range %.1f)' % (e, float(tr['mask_logits'].abs().max())))
    assert e < 1e-3
    out = model.test_step(dict(inputs=[i.to(dev) for i in imgs], data_samples=[DetDataSample(metainfo=dict(m)) for m in metas]))
    for b in range(2):
        pi, r = out[b].pred_instances, ref[b]
        assert tuple(pi.masks.shape[1:]) == (512, 512) and pi.labels.shape[0] == r['labels'].shape[0]
        pairs = match_detections(pi.bboxes, pi.scores, pi.labels, r['bboxes'], r['scores'], r['labels'])
        ii = torch.tensor([i for i, _ in pairs]); jj = torch.tensor([j for _, j in pairs])
        mism = float((pi.masks.cpu()[ii] != r['masks'][jj]).float().mean())
        print(f'SAMSegMaskRCNN img {b}: {pi.labels.shape[0]} dets, {len(pairs)} matched, mask mismatch {mism:.2e}')
        assert mism < 1e-3


def _maxerr(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())


def test_msdeform_attn_head_dim_32_and_groupnorm_256(dev):
    """the two kernels the 256-wide pixel decoder needs beyond the RSPrompter shapes: rsp_msdeform_attn_ex(head_dim=32)
    and GroupNorm(32) over 256 channels (8 per group)."""
    import torch.nn.functional as F
    from oracle.query import MSDeformAttn
    from rsprompter_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 256, 12, 20, generator=g) * 2 + 0.5
    w, b = torch.randn(256, generator=g), torch.randn(256, generator=g)
    ref = F.group_norm(x.double(), 32, w.double(), b.double(), 1e-5)
    got = ops.groupnorm(x.permute(0, 2, 3, 1).contiguous().to(dev), w.to(dev), b.to(dev), 32, relu=True)
    assert _maxerr(got.permute(0, 3, 1, 2), F.relu(ref)) < 2e-5
    torch.manual_seed(77)
    m = MSDeformAttn(256)
    shapes = [(4, 4), (8, 8), (16, 16)]
    ntok = sum(h * w_ for h, w_ in shapes)
    q, pos = torch.randn(2, ntok, 256, generator=g), torch.randn(2, ntok, 256, generator=g)
    refp = torch.rand(ntok, 2, generator=g)
    with torch.no_grad():
        m.sampling_offsets.weight.mul_(20)          # offsets large enough to leave the maps
        value = m.value_proj(q)
        ow = torch.cat([m.sampling_offsets(q + pos), m.attention_weights(q + pos)], -1)
        want = (m(q, pos, refp[None, :, None].repeat(2, 1, 3, 1), torch.tensor(shapes)) - q).double()
    got = ops.msdeform_attn(value.reshape(-1, 256).contiguous().to(dev), ow.reshape(-1, 288).contiguous().to(dev),
                            refp.to(dev), 2, ntok, shapes, head_dim=32)
    got_full = got.cpu().double().view(2, ntok, 256) @ m.output_proj.weight.double().t() + m.output_proj.bias.double()
    assert _maxerr(got_full, want) < 1e-4
    with pytest.raises(ValueError):
        ops.msdeform_attn(value.reshape(-1, 256).contiguous().to(dev), ow.reshape(-1, 288).contiguous().to(dev),
                          refp.to(dev), 2, ntok, shapes, head_dim=16)


def test_samseg_mask2former_end_to_end(dev):
    """SAMSegMask2Former (standard Mask2FormerHead, feat 256, 9 decoder layers) on the HIP kernels against
    oracle/samseg.py (head pinned on the real mask2former_head.py): head stage test on the oracle's FPN, then test_step."""
    import rsprompter_amd as ra
    from oracle import glue
    from oracle.samseg import SAMSegMask2FormerOracle
    from rsprompter_amd.default_configs import samseg_mask2former
    from rsprompter_amd.structures import DetDataSample
    from rsprompter_amd.synth import synth_images, synth_metas, synth_state_dict
    NQ = 70
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = ra.build_model(samseg_mask2former('base', 10, NQ))
    oracle = SAMSegMask2FormerOracle('base', 10, num_queries=NQ)
    sd = synth_state_dict(oracle, seed=0)
    oracle.load_state_dict(sd)
    res = model.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    model = model.to(dev)
    imgs, metas = synth_images(1), synth_metas(1)
    x = glue.data_preprocess(imgs, MEAN, STD, True, 32)
    ref, tr = oracle.predict(x, metas)
    feats = [f.to(dev).contiguous(memory_format=torch.channels_last) for f in tr['fpn']]
    cls, mask_pred, t = model.panoptic_head(feats, None)
    flips = []
    for i, (a, b) in enumerate(zip(t['attn_masks'], tr['attn_masks'])):
        flips.append(float((a.cpu().bool() != b.view(1, 8, NQ, -1)[:, 0]).float().mean()))
    e_q = max(_maxerr(a.view(1, NQ, -1), b) for a, b in zip(t['query_feats'], tr['query_feats'][1:]))
    e_cls, e_mask = _maxerr(cls, tr['cls_pred']), _maxerr(mask_pred, tr['mask_pred'])
    print('Mask2FormerHead on oracle FPN: attention-mask flips max %.2e, query_feat %.2e cls %.2e mask logits %.2e '
          '(range %.1f)' % (max(flips), e_q, e_cls, e_mask, float(tr['mask_pred'].abs().max())))
    assert max(flips) < 1e-3 and e_q < 1e-3 and e_cls < 1e-3 and e_mask < 2e-3
    out = model.test_step(dict(inputs=[i.to(dev) for i in imgs], data_samples=[DetDataSample(metainfo=dict(m)) for m in metas]))
    pi, r = out[0].pred_instances, ref[0]
    assert pi.masks.dtype == torch.bool and tuple(pi.masks.shape) == tuple(r['masks'].shape)
    same = pi.query_indices.cpu().long() == r['query_indices']
    mism = float((pi.masks.cpu()[same] != r['masks'][same]).float().mean())
    e_s = _maxerr(pi.scores[same.to(pi.scores.device)], r['scores'][same])
    print('SAMSegMask2Former e2e: query-index agreement %.3f, score err %.2e, mask mismatch %.2e' %
          (float(same.float().mean()), e_s, mism))
    assert float(same.float().mean()) > 0.9 and e_s < 1e-3 and mism < 1e-3
