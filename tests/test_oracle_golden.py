"""CPU: pin the oracle (and the product's host-side constant builders) against golden vectors produced
by executing the REAL reference source files (tests/golden/make_golden.py -> reference_vectors.pt) and
against the two known-answer tests the reference inherits."""
import os

import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_vectors.pt')


@pytest.fixture(scope='module')
def gold():
    return torch.load(GOLD, weights_only=False)


def test_delta2bbox_known_answer_from_reference_tests():
    """tests/test_models/test_task_modules/test_coder/test_delta_xywh_bbox_coder.py:9-24"""
    from oracle import glue
    rois = torch.Tensor([[0., 0., 1., 1.], [0., 0., 1., 1.], [0., 0., 1., 1.], [5., 5., 5., 5.]])
    deltas = torch.Tensor([[0., 0., 0., 0.], [1., 1., 1., 1.], [0., 0., 2., -1.], [0.7, -1.9, -0.5, 0.3]])
    expected = torch.Tensor([[0.0000, 0.0000, 1.0000, 1.0000], [0.1409, 0.1409, 2.8591, 2.8591],
                             [0.0000, 0.3161, 4.1945, 0.6839], [5.0000, 5.0000, 5.0000, 5.0000]])
    assert torch.allclose(glue.delta2bbox(rois, deltas, max_shape=(32, 32)), expected, atol=1e-4)


def test_anchor_generator_known_answer_from_reference_tests():
    """tests/test_models/test_task_modules/test_prior_generators/test_anchor_generator.py:290-309"""
    from oracle import glue
    from rsprompter_amd.anchor_heads import AnchorGenerator
    exp = [torch.Tensor([[-2., -2., 2., 2.], [2., -2., 6., 2.], [-2., 2., 2., 6.], [2., 2., 6., 6.]]),
           torch.Tensor([[-4., -4., 4., 4.]])]
    got = [glue.grid_priors([(2, 2)], [4], [1.], [1.])[0], glue.grid_priors([(1, 1)], [8], [1.], [1.])[0]]
    ours = AnchorGenerator(strides=[4, 8], ratios=[1.], scales=[1.], base_sizes=[4, 8]).grid_priors([(2, 2), (1, 1)])
    for e, g, o in zip(exp, got, ours):
        assert torch.equal(e, g) and torch.equal(e, o)


def test_delta2bbox_matches_reference_source(gold):
    from oracle import glue
    for key in ('delta2bbox_kat', 'delta2bbox_rand', 'delta2bbox_rpn'):
        d = gold[key]
        out = glue.delta2bbox(d['rois'], d['deltas'], stds=d.get('stds', (1., 1., 1., 1.)), max_shape=d['max_shape'])
        assert torch.equal(out, d['out']), key


def test_anchor_generator_matches_reference_source(gold):
    from oracle import glue
    from rsprompter_amd.anchor_heads import AnchorGenerator
    a = gold['anchors']
    pri = glue.grid_priors(a['sizes'], [4, 8, 16, 32, 64], [4, 8], [0.5, 1.0, 2.0])
    gen = AnchorGenerator(strides=[4, 8, 16, 32, 64], ratios=[0.5, 1.0, 2.0], scales=[4, 8])
    ours = gen.grid_priors(a['sizes'])
    for lvl in range(5):
        assert torch.equal(glue.gen_base_anchors([4, 8, 16, 32, 64][lvl], [4, 8], [0.5, 1.0, 2.0]), a['base'][lvl])
        assert torch.equal(gen.base_anchors[lvl], a['base'][lvl])
        assert pri[lvl].shape[0] == a['counts'][lvl] == ours[lvl].shape[0]
        assert torch.equal(pri[lvl][a['sample_idx'][lvl]], a['samples'][lvl])
        assert torch.equal(ours[lvl][a['sample_idx'][lvl]], a['samples'][lvl])
    for o, e in zip(glue.grid_priors([(2, 2), (1, 1)], [4, 8], [1.], [1.]), gold['anchors_kat']):
        assert torch.equal(o, e)


def test_positional_encodings_match_reference_source(gold):
    from oracle import glue
    from rsprompter_amd.anchor_heads import _sine_pe
    assert torch.equal(glue.sine_positional_encoding(1, 24, 40, 128), gold['sine_pe'])
    assert torch.equal(_sine_pe(24, 40, 128), gold['sine_pe'])          # product-side constant table
    ip = gold['image_pe']
    assert torch.equal(glue.image_wide_positional_embeddings(ip['G'], 16), ip['out'])
    from rsprompter_amd.sam_decoder import RSSamPositionalEmbedding
    m = RSSamPositionalEmbedding('sam_vit_base')
    m.shared_image_embedding.positional_embedding.data.copy_(ip['G'])
    assert torch.equal(m.image_wide(16), ip['out'])


def test_window_and_relpos_semantics_match_vit_sam_source(gold):
    """window_partition / unpartition row maps and rel-pos tables (vit_sam.py:17-157) versus the host
    logic of the HIP encoder (row maps) and the torch restatement used by the kernel tests."""
    from rsprompter_amd.sam_encoder import SamVisionEncoderHIP, resize_rel_pos
    from rsprompter_amd.nnutil import SAM_ARCH
    w = gold['window']
    SAM_ARCH['t'] = dict(hidden=8, depth=1, heads=1, global_idx=(), mlp=8)
    try:
        enc = SamVisionEncoderHIP('t', image_size=320)   # 20x20 grid, window 14 -> 2x2 windows with padding
    finally:
        SAM_ARCH.pop('t')
    m, nw = enc._window_map(2, torch.device('cpu'))[:2]
    x = w['x'].reshape(-1, 8)
    rows = torch.where((m >= 0)[:, None], x[m.clamp(min=0).long()], torch.zeros(1))
    assert nw == 2 and torch.equal(rows.view(-1, 14, 14, 8), w['windows'])
    back = torch.zeros_like(x)
    back[m[m >= 0].long()] = w['windows'].reshape(-1, 8)[m >= 0]
    assert torch.equal(back.view(2, 20, 20, 8), w['back'])
    # the inverse map the GEMMs use (qkv scatters token rows to window order, proj gathers them back) and the pad rows
    _, _, tok2win, pad_rows = enc._window_map(2, torch.device('cpu'))
    assert torch.equal(m[tok2win.long()].long(), torch.arange(2 * 400))
    assert torch.equal(pad_rows.long(), (m < 0).nonzero()[:, 0]) and tok2win.shape[0] + pad_rows.shape[0] == m.shape[0]
    assert torch.equal(w['windows'].reshape(-1, 8)[tok2win.long()], x)
    rp = gold['rel_pos']
    idx = torch.arange(14)[:, None] - torch.arange(14)[None, :] + 13
    assert torch.equal(resize_rel_pos(rp['rel_pos'], 14)[idx], rp['same'])
    idx20 = torch.arange(20)[:, None] - torch.arange(20)[None, :] + 19
    assert torch.allclose(resize_rel_pos(rp['rel_pos'], 20)[idx20], rp['resized'], atol=1e-6)
    d = gold['decomposed_rel_pos']
    Rh, Rw = d['rph'][idx], d['rpw'][idx]
    rq = d['q'].view(3, 14, 14, 16)
    rel_h = torch.einsum('bhwc,hkc->bhwk', rq, Rh)
    rel_w = torch.einsum('bhwc,wkc->bhwk', rq, Rw)
    ours = (d['attn'].view(3, 14, 14, 14, 14) + rel_h[..., None] + rel_w[..., None, :]).view(3, 196, 196)
    assert torch.allclose(ours, d['out'], atol=1e-5)


def test_oracle_vitsam_helpers_match_reference_source(gold):
    from oracle import vitsam
    w = gold['window']
    win, pad = vitsam.window_partition(w['x'], 14)
    assert torch.equal(win, w['windows']) and tuple(pad) == tuple(w['pad_hw'])
    assert torch.equal(vitsam.window_unpartition(win, 14, pad, (20, 20)), w['back'])
    rp = gold['rel_pos']
    assert torch.equal(vitsam.get_rel_pos(14, 14, rp['rel_pos']), rp['same'])
    assert torch.allclose(vitsam.get_rel_pos(20, 20, rp['rel_pos']), rp['resized'], atol=1e-6)


def test_ln2d_and_aggregator_match_reference_source(gold):
    from oracle.anchor import LN2d, FeatureAggregator
    l = gold['ln2d']
    m = LN2d(8)
    m.weight.data.copy_(l['w']); m.bias.data.copy_(l['b'])
    assert torch.allclose(m(l['x']), l['out'], atol=1e-6)
    a = gold['aggregator']
    agg = FeatureAggregator('base', 16, 64, range(1, 13, 2)).eval()
    agg.load_state_dict(a['state'], strict=True)          # same key layout as the reference class
    with torch.no_grad():
        assert torch.allclose(agg(a['inputs']), a['out'], atol=1e-5)


def test_mask_postprocess_matches_reference_source(gold):
    from oracle import glue
    for tag in ('ident', 'rescale', 'odd'):
        d = gold[f'mask_post_{tag}']
        masks, boxes, _ = glue.mask_postprocess_single(d['low'], d['boxes'].clone(), d['meta'], 0.5, True)
        assert torch.equal(masks, d['masks']), tag
        assert torch.equal(boxes, d['boxes_out']), tag


def test_coco_rle_restatement_reproduces_reference_strings():
    """oracle/rle.py against the compressed RLE strings of the reference's tests/data/vis_sample.json."""
    import json
    import os
    import numpy as np
    from oracle import rle
    d = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'coco_rle_strings.json')))
    assert len(d['items']) >= 3
    for it in d['items']:
        h, w = it['size']
        m = rle.rle_decode(rle.rle_from_string(it['counts']), h, w)
        assert int(m.sum()) == int(it['area'])
        ys, xs = np.nonzero(m)
        assert [int(xs.min()), int(ys.min()), int(xs.max() - xs.min() + 1), int(ys.max() - ys.min() + 1)] == it['bbox']
        assert rle.encode(m)['counts'].decode() == it['counts']
    g = np.random.default_rng(0)
    for shape in ((7, 5), (64, 33), (1, 9)):
        m = g.random(shape) > 0.5
        c = rle.rle_counts(m)
        assert int(c.sum()) == m.size
        assert np.array_equal(rle.rle_decode(rle.rle_from_string(rle.rle_to_string(c)), *shape), m)


def test_rle_host_string_compression_matches_restatement():
    """rsprompter_amd/rle.py::_counts_to_string (the host half of encode_mask_results) vs oracle/rle.py, incl. the
    reference's own strings."""
    import json
    import os
    import numpy as np
    from oracle import rle
    from rsprompter_amd.rle import _counts_to_string
    d = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'coco_rle_strings.json')))
    for it in d['items']:
        cnts = [int(c) for c in rle.rle_from_string(it['counts'])]
        assert _counts_to_string(cnts).decode() == it['counts']
    g = np.random.default_rng(1)
    for _ in range(20):
        cnts = [int(v) for v in g.integers(0, 5000, size=int(g.integers(1, 60)))]
        assert _counts_to_string(cnts) == rle.rle_to_string(cnts)


def test_query_postprocess_restatements_match_reference_vectors():
    """oracle/query.py::mask2bbox / instance_postprocess against vectors produced by the REAL reference sources
    (tests/golden/make_golden_query.py: structures/mask/utils.py:56-77, maskformer_fusion_head.py:126-182).
    The reference keeps its top-k with sorted=False (order unspecified): compared order-free by (query, label)."""
    import os
    from oracle import query as oq
    d = torch.load(os.path.join(os.path.dirname(__file__), 'golden', 'reference_vectors_query.pt'))
    m = d['mask2bbox']
    assert torch.equal(oq.mask2bbox(m['masks']), m['out'])
    for c in d['instance_postprocess']:
        r = oq.instance_postprocess(c['mask_cls'], c['mask_pred'], c['num_classes'], c['max_per_image'])
        assert r['labels'].shape[0] == c['labels'].shape[0]

        def key(scores, labels, boxes):
            # det score, label and box identify an entry; sort for an order-free comparison
            k = torch.stack([scores.double(), labels.double(), boxes[:, 0].double(), boxes[:, 1].double()], 1)
            idx = sorted(range(k.shape[0]), key=lambda i: tuple(k[i].tolist()))
            return torch.tensor(idx)
        ia, ib = key(r['scores'], r['labels'], r['bboxes']), key(c['scores'], c['labels'], c['bboxes'])
        assert torch.equal(r['labels'][ia], c['labels'][ib])
        assert torch.equal(r['bboxes'][ia], c['bboxes'][ib])
        assert float((r['scores'][ia] - c['scores'][ib]).abs().max()) < 1e-6
        assert torch.equal(r['masks'][ia], c['masks'][ib])
    # RSMaskFormerFusionHead.predict (models.py:663-715): padding crop, logit resize, then the above
    for c in d['fusion_predict']:
        r = oq.fusion_predict(c['mask_cls'], c['mask_pred'], [c['meta']], c['num_classes'], c['max_per_image'], True)[0]
        ia, ib = key(r['scores'], r['labels'], r['bboxes']), key(c['scores'], c['labels'], c['bboxes'])
        assert tuple(r['masks'].shape) == tuple(c['masks'].shape)
        assert torch.equal(r['labels'][ia], c['labels'][ib]) and torch.equal(r['bboxes'][ia], c['bboxes'][ib])
        assert float((r['scores'][ia] - c['scores'][ib]).abs().max()) < 1e-6
        assert torch.equal(r['masks'][ia], c['masks'][ib])


def test_detection_glue_restatements_match_reference_vectors():
    """oracle/glue.py::rpn_predict_single / multiclass_nms / map_roi_levels against vectors produced by the REAL
    reference sources around an injected batched_nms (tests/golden/make_golden_heads.py): pins the index-deciding glue
    of the anchor path (rpn_head.py:134-304, bbox_nms.py:12-105, single_level_roi_extractor.py:44-63)."""
    import os
    from oracle import glue
    d = torch.load(os.path.join(os.path.dirname(__file__), 'golden', 'reference_vectors_heads.pt'))
    # the second group: softmax objectness ([fg, bg] per anchor, rpn_head.py:193-200) on the real class with use_sigmoid_cls=False
    for c, sig in [(c, True) for c in d['rpn_predict_single']] + [(c, False) for c in d['rpn_predict_single_softmax']]:
        priors = glue.grid_priors(c['sizes'], [4, 8, 16, 32, 64], [4, 8], [0.5, 1.0, 2.0])
        r = glue.rpn_predict_single(c['cls'], c['reg'], priors, c['img_shape'], nms_pre=c['nms_pre'],
                                    max_per_img=c['max_per_img'], iou_thr=c['iou_thr'], min_bbox_size=c['min_bbox_size'],
                                    use_sigmoid_cls=sig)
        assert r['bboxes'].shape == c['bboxes'].shape, (r['bboxes'].shape, c['bboxes'].shape)
        assert torch.equal(r['scores'], c['scores'])

        def canon(boxes, scores):
            # the reference sorts with torch.sort(descending=True) (NOT stable, rpn_head.py:208): entries whose fp32
            # sigmoid scores are exactly equal may come in either order -> canonical order inside equal-score runs
            k = torch.cat([-scores[:, None].double(), boxes.double()], 1)
            idx = sorted(range(k.shape[0]), key=lambda i: tuple(k[i].tolist()))
            return boxes[torch.tensor(idx)]
        assert torch.equal(canon(r['bboxes'], r['scores']), canon(c['bboxes'], c['scores']))
        n_swapped = int((r['bboxes'] != c['bboxes']).any(1).sum())
        assert n_swapped <= 4                                # only exact ties may differ in position
    for c in d['multiclass_nms']:
        dets, labels, inds = glue.multiclass_nms(c['boxes'], c['scores'], c['score_thr'], c['iou_thr'], c['max_num'])
        assert torch.equal(dets, c['dets']) and torch.equal(labels, c['labels']) and torch.equal(inds, c['inds'])
    for c in d['bbox_head_predict_single']:
        dets, labels, _ = glue.bbox_head_predict_single(c['roi'], c['cls_score'], c['bbox_pred'], c['img_shape'],
                                                        c['num_classes'], c['score_thr'], c['iou_thr'], c['max_per_img'])
        assert torch.equal(dets[:, :4], c['bboxes']) and torch.equal(dets[:, 4], c['scores'])
        assert torch.equal(labels, c['labels'])
    # rescale=True (mask-less RoI heads: SAMDet's detector): bbox_head.py:549-552 through the REAL scale_boxes -- bit-exact,
    # which a division by the scale factor instead of the product with fp32(1 / s) would not be
    for c in d['bbox_head_predict_single_rescale']:
        dets, labels, _ = glue.bbox_head_predict_single(c['roi'], c['cls_score'], c['bbox_pred'], c['img_shape'],
                                                        c['num_classes'], c['score_thr'], c['iou_thr'], c['max_per_img'],
                                                        scale_factor=c['scale_factor'])
        assert torch.equal(dets[:, :4], c['bboxes']) and torch.equal(dets[:, 4], c['scores'])
        assert torch.equal(labels, c['labels'])
    m = d['map_roi_levels']
    assert torch.equal(glue.map_roi_levels(m['rois'], m['num_levels'], m['finest_scale']), m['out'])


def _coder_kw(kw):
    """DeltaXYWHBBoxCoder constructor keywords -> oracle.glue.delta2bbox keywords"""
    m = dict(target_means='means', target_stds='stds')
    return {m.get(k, k): v for k, v in kw.items()}


def test_box_coder_branches_match_reference_vectors():
    """The DeltaXYWHBBoxCoder branches no RSPrompter config uses (target_means != 0, clip_border=False, add_ctr_clamp:
    delta_xywh_bbox_coder.py:264-361) against the REAL coder, alone and inside the REAL RPNHead / BBoxHead
    `_predict_by_feat_single` (tests/golden/make_golden_coder.py), incl. the reference's own known-answer test of the
    centre clamp (test_delta_xywh_bbox_coder.py:44-57)."""
    import os
    from oracle import glue
    d = torch.load(os.path.join(os.path.dirname(__file__), 'golden', 'reference_vectors_coder.pt'))
    k = d['kat_ctr_clamp']
    out = glue.delta2bbox(k['rois'], k['deltas'], max_shape=k['max_shape'], **_coder_kw(k['coder']))
    assert torch.allclose(out, k['expected'], atol=1e-4) and torch.equal(out, k['out'])
    differs = 0
    for name, kw in d['coders'].items():
        c = d['decode'][name]
        out = glue.delta2bbox(c['rois'], c['deltas'], max_shape=c['max_shape'], **_coder_kw(kw))
        assert torch.equal(out, c['out']), name
        differs += int(not torch.equal(out, d['decode']['plain']['out']))
        c = d['rpn_predict_single'][name]
        priors = glue.grid_priors(c['sizes'], [4, 8, 16, 32, 64], [4, 8], [0.5, 1.0, 2.0])
        r = glue.rpn_predict_single(c['cls'], c['reg'], priors, c['img_shape'], nms_pre=c['nms_pre'],
                                    max_per_img=c['max_per_img'], iou_thr=c['iou_thr'], min_bbox_size=c['min_bbox_size'],
                                    coder=_coder_kw(kw))
        assert torch.equal(r['scores'], c['scores']), name
        if not torch.equal(r['bboxes'], c['bboxes']):      # the reference's sort is not stable (rpn_head.py:208): rows may
            bad = (r['bboxes'] != c['bboxes']).any(1)      # only trade places inside runs of exactly equal scores
            assert int(bad.sum()) <= 4, name
            for sc in r['scores'][bad].unique():
                run = r['scores'] == sc
                assert sorted(map(tuple, r['bboxes'][run].tolist())) == sorted(map(tuple, c['bboxes'][run].tolist())), name
        c = d['bbox_head_predict_single'][name]
        dets, labels, _ = glue.bbox_head_predict_single(c['roi'], c['cls_score'], c['bbox_pred'], c['img_shape'],
                                                        c['num_classes'], c['score_thr'], c['iou_thr'], c['max_per_img'],
                                                        coder=_coder_kw(kw))
        assert torch.equal(dets[:, :4], c['bboxes']) and torch.equal(dets[:, 4], c['scores']), name
        assert torch.equal(labels, c['labels']), name
    assert differs == len(d['coders']) - 1          # every variant decodes different boxes than the shipped coder


def test_mmcv_leaf_known_answers():
    """Hand-derivable known-answer vectors for the mmcv LEAF ops whose source is not under /root/reference
    (SURVEY.md App. B): the C / torch restatements (oracle/mmcv_ops.c, oracle/query.py) are checked by something other
    than themselves.  Every expected value below follows from the operator's published definition by hand."""
    import torch.nn as nn
    from oracle import cops, glue
    from oracle.query import FFN, MHA, MSDeformAttn

    # ---- RoIAlign(aligned=True, sampling_ratio=0, avg) -------------------------------------------------------------
    H = W = 16
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')
    const = torch.full((1, 1, H, W), 3.25)
    ramp = (2.0 * xs + 0.5 * ys)[None, None]                       # bilinear interpolation of a linear map is exact
    feat = torch.cat([const, ramp], 1)
    roi = torch.tensor([[0., 4., 8., 20., 24.]])                   # scale 0.5 -> x in [1.5, 9.5], y in [3.5, 11.5]
    out = cops.roi_align(feat, roi, 2, 0.5, 0, True)
    assert torch.allclose(out[0, 0], torch.full((2, 2), 3.25))
    # the samples of a bin are symmetric about its centre, so the bin average of a linear map is its centre value
    xc, yc = torch.tensor([3.5, 7.5]), torch.tensor([5.5, 9.5])
    assert torch.allclose(out[0, 1], 2.0 * xc[None, :] + 0.5 * yc[:, None], atol=1e-5)
    # a RoI hanging over the left border: samples with x < -1 contribute 0, samples in [-1, 0] are clamped to x = 0
    roi2 = torch.tensor([[0., -8., 0., 0., 4.]])                   # scale 1: x in [-8.5, -0.5], one 1x1 bin, 8x4 samples
    o2 = cops.roi_align(torch.ones(1, 1, H, W), roi2, 1, 1.0, 0, True)
    assert abs(float(o2) - 1.0 / 8.0) < 1e-6                       # only the sample column x = -1.0 survives: 4 of 32
    # ---- nms (offset 0, IoU > thr suppresses, strict) and batched_nms ----------------------------------------------
    boxes = torch.tensor([[0., 0., 10., 10.], [1., 1., 11., 11.], [20., 20., 30., 30.], [0., 0., 10., 5.]])
    scores = torch.tensor([0.9, 0.8, 0.7, 0.6])
    # IoU(0,1) = 81 / 119 = 0.6807;  IoU(0,3) = 50 / 100 = 0.5 EXACTLY;  IoU(1,3) = 36 / 114
    assert cops.nms(boxes, scores, 0.7)[1].tolist() == [0, 1, 2, 3]
    assert cops.nms(boxes, scores, 0.6)[1].tolist() == [0, 2, 3]                 # box 1 goes (0.6807 > 0.6)
    assert cops.nms(boxes, scores, 0.5)[1].tolist() == [0, 2, 3]                 # 0.5 is NOT > 0.5: box 3 stays
    assert cops.nms(boxes, scores, 0.49)[1].tolist() == [0, 2]
    tie = torch.tensor([0.5, 0.5, 0.5, 0.5])
    assert cops.nms(boxes, tie, 0.6)[1].tolist() == [0, 2, 3]                    # stable order on exact ties
    _, keep = glue.batched_nms(boxes, scores, torch.tensor([0, 1, 0, 1]), 0.35)  # same-id pair (1, 3): IoU 36/114 = 0.316
    assert sorted(keep.tolist()) == [0, 1, 2, 3]                                 # different ids never suppress each other
    _, keep = glue.batched_nms(boxes, scores, torch.tensor([0, 0, 0, 0]), 0.35)
    assert sorted(keep.tolist()) == [0, 2]
    # ---- MultiScaleDeformableAttention: zero offsets + uniform weights + identity projections ----------------------
    torch.manual_seed(0)
    m = MSDeformAttn(dim=8, heads=2, levels=2, points=2)
    with torch.no_grad():
        for lin in (m.sampling_offsets, m.attention_weights):
            lin.weight.zero_(); lin.bias.zero_()                  # offsets 0, logits 0 -> softmax = 1 / (L * P)
        for lin in (m.value_proj, m.output_proj):
            lin.weight.copy_(torch.eye(8)); lin.bias.zero_()
    shapes = torch.tensor([[2, 2], [1, 1]])
    q = torch.arange(5 * 8, dtype=torch.float32).view(1, 5, 8) / 10.0             # tokens: 4 of level 0, 1 of level 1
    # reference point = centre of pixel (y=1, x=0) of the 2x2 level == centre region of the 1x1 level
    ref = torch.tensor([0.25, 0.75]).view(1, 1, 1, 2).repeat(1, 5, 2, 1)
    out = m(q, torch.zeros_like(q), ref, shapes)
    # sampling AT a pixel centre returns that pixel: level 0 -> token 2 (row 1, col 0); level 1: (0.25, 0.75) of a 1x1 map
    # is bilinear between the pixel (weight 0.75 * 0.75) and the zero padding -> 0.5625 * token 4
    want = q + 0.5 * (q[:, 2:3] + 0.5625 * q[:, 4:5])
    assert torch.allclose(out, want, atol=1e-6)
    # ---- MultiheadAttention wrapper: q = query + query_pos, k = key + key_pos, v = value, + identity ----------------
    a = MHA(dim=8, heads=2)
    with torch.no_grad():
        a.attn.in_proj_weight.copy_(torch.cat([torch.eye(8)] * 3)); a.attn.in_proj_bias.zero_()
        a.attn.out_proj.weight.copy_(torch.eye(8)); a.attn.out_proj.bias.zero_()
    qq, kk = torch.randn(1, 3, 8), torch.randn(1, 4, 8)
    mask = torch.ones(2, 3, 4, dtype=torch.bool)
    mask[:, :, 1] = False                                          # every query may only see key 1
    got = a(qq, kk, kk * 2.0, torch.randn(1, 3, 8), torch.randn(1, 4, 8), mask)
    assert torch.allclose(got, qq + (kk * 2.0)[:, 1:2].expand(1, 3, 8), atol=1e-6)   # the VALUE has no positional term
    # ---- FFN: x + Linear(ReLU(Linear(x))) --------------------------------------------------------------------------
    f = FFN(4, 6)
    x = torch.randn(2, 4)
    with torch.no_grad():
        w0, b0 = f.layers[0][0].weight, f.layers[0][0].bias
        w1, b1 = f.layers[1].weight, f.layers[1].bias
        assert torch.allclose(f(x), x + torch.relu(x @ w0.t() + b0) @ w1.t() + b1, atol=1e-6)


def test_mmcv_leaf_second_formulations():
    """The mmcv leaf ops whose source is absent from /root/reference (RoIAlign, MultiScaleDeformableAttention sampling) are
    restated in oracle/ from their published definitions and pinned by hand-derived known answers only (the
    oracle is its own only witness there).  Here each gets an INDEPENDENT second formulation, written from the definition
    along another route, and both must agree on random inputs:
      * RoIAlign(avg, aligned, adaptive grid): bilinear interpolation is separable, so a RoI's output is Ay . f . Ax^T with
        1-D interpolation-and-averaging matrices built from the border rules (sample outside [-1, size] -> no weight,
        clamp to 0, last pixel takes all weight beyond size - 1) -- no per-sample loop, no 2-D weights;
      * MSDeformAttn sampling: explicit four-corner gathers with validity masks at pixel coordinates loc * size - 0.5
        instead of F.grid_sample."""
    import math
    from oracle import cops
    from oracle.query import MSDeformAttn
    g = torch.Generator().manual_seed(7)

    # ---- RoIAlign -----------------------------------------------------------------------------------------------------
    def interp_matrix(lo, length, nbin, size):
        """[nbin, size]: row p = average over the bin's samples of the 1-D tent weights (mmcv bilinear_interpolate rules)"""
        grid = max(int(math.ceil(length / nbin)), 1) if length > 0 else 1
        n_samp = int(math.ceil(length / nbin))
        A = torch.zeros(nbin, size, dtype=torch.float64)
        if n_samp <= 0:
            return A, 1
        bin_sz = length / nbin
        for p in range(nbin):
            for i in range(n_samp):
                y = lo + p * bin_sz + (i + 0.5) * bin_sz / n_samp
                if y < -1.0 or y > size:
                    continue
                y = max(y, 0.0)
                yl = int(y)
                if yl >= size - 1:
                    A[p, size - 1] += 1.0
                else:
                    A[p, yl] += 1.0 - (y - yl)
                    A[p, yl + 1] += y - yl
        return A, n_samp

    feat = torch.randn(2, 5, 24, 30, generator=g)
    rois = torch.tensor([[0, 3.2, 4.1, 40.7, 33.3], [1, -9.0, -6.0, 25.0, 18.5], [0, 50.0, 30.0, 75.0, 60.0],
                         [1, 10.0, 10.0, 10.4, 10.2], [0, 0.0, 0.0, 59.9, 47.9]])
    for P, scale in ((7, 0.5), (14, 0.25), (3, 1.0)):
        got = cops.roi_align(feat, rois, P, scale, 0, True).double()
        for k, r in enumerate(rois.tolist()):
            b = int(r[0])
            x1, y1, x2, y2 = (float(torch.tensor(v * scale, dtype=torch.float32) - 0.5) for v in r[1:])
            Ay, gh = interp_matrix(y1, y2 - y1, P, feat.shape[2])
            Ax, gw = interp_matrix(x1, x2 - x1, P, feat.shape[3])
            want = torch.einsum('ph,chw,qw->cpq', Ay, feat[b].double(), Ax) / max(gh * gw, 1)
            assert float((got[k] - want).abs().max()) < 1e-4, (P, scale, k, float((got[k] - want).abs().max()))

    # ---- MSDeformAttn sampling ----------------------------------------------------------------------------------------
    bs, nq, H_, D, L, Pn = 2, 11, 4, 8, 3, 4
    shapes = torch.tensor([[6, 9], [3, 5], [2, 2]])
    ntok = int((shapes[:, 0] * shapes[:, 1]).sum())
    value = torch.randn(bs, ntok, H_, D, generator=g)
    loc = torch.rand(bs, nq, H_, L, Pn, 2, generator=g) * 1.3 - 0.15          # some points outside the map: zero padding
    w = torch.rand(bs, nq, H_, L, Pn, generator=g)
    got = MSDeformAttn._sample(value, shapes, loc, w)                          # [bs, nq, H*D]
    want = torch.zeros(bs, nq, H_, D, dtype=torch.float64)
    start = 0
    for lvl, (h, w_) in enumerate(shapes.tolist()):
        v = value[:, start:start + h * w_].double().view(bs, h, w_, H_, D)
        start += h * w_
        x = loc[:, :, :, lvl, :, 0].double() * w_ - 0.5                         # [bs, nq, H, P]
        y = loc[:, :, :, lvl, :, 1].double() * h - 0.5
        x0, y0 = torch.floor(x), torch.floor(y)
        for dy in (0, 1):
            for dx in (0, 1):
                xi, yi = x0 + dx, y0 + dy
                wt = (1 - (x - xi).abs()) * (1 - (y - yi).abs())
                ok = (xi >= 0) & (xi < w_) & (yi >= 0) & (yi < h)
                xi_c, yi_c = xi.clamp(0, w_ - 1).long(), yi.clamp(0, h - 1).long()
                bi = torch.arange(bs).view(bs, 1, 1, 1).expand_as(xi_c)
                hi = torch.arange(H_).view(1, 1, H_, 1).expand_as(xi_c)
                samp = v[bi, yi_c, xi_c, hi]                                    # [bs, nq, H, P, D]
                want += (samp * (wt * ok * w[:, :, :, lvl].double())[..., None]).sum(3)
    assert float((got.double().view(bs, nq, H_, D) - want).abs().max()) < 1e-5
