"""CPU: the oracle's module forwards against the REAL reference classes executed in the build container
(tests/golden/make_golden_forwards.py -> golden/reference_vectors_forwards.pt): RSSimpleFPN, PseudoFeatureAggregator
(+ RSFPN), RSPrompterAnchorMaskHead, RSMask2FormerHead (incl. MSDeformAttnPixelDecoder and the Mask2Former decoder
layers) and ViTSAM.  Weights and inputs are pure functions of (seed, key, shape) on both sides, so each test also pins
the oracle's `state_dict` key layout (names AND shapes AND order) to the real class's."""
import os

import pytest
import torch
from torch import nn

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_vectors_forwards.pt')
TOL = 2e-5          # fp32 re-association noise between two CPU evaluations of the same network


@pytest.fixture(scope='module')
def gold():
    return torch.load(GOLD, weights_only=False)


def rnd(spec):
    seed, shape = spec
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def load(m, g):
    from rsprompter_amd.synth import synth_state_dict
    got = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    assert sorted(got) == sorted(g['keys']), (set(got) ^ set(g['keys']))
    m.load_state_dict(synth_state_dict(m, g['seed']), strict=True)
    return m.eval()


def err(a, b):
    return float((a - b).abs().max())


@torch.no_grad()
def test_simple_fpn_matches_real_class(gold):
    from oracle.anchor import SimpleFPN
    g = gold['simple_fpn']
    assert g['eps'] == [1e-5, 1e-5, 1e-5]      # mmcv build_norm_layer's eps default reaches LN2d (not LN2d's own 1e-6)
    m = load(SimpleFPN(), g)
    outs = m(rnd(g['x']))
    assert len(outs) == len(g['outs']) == 5
    for a, b in zip(outs, g['outs']):
        assert a[:, ::8].shape == b.shape and err(a[:, ::8], b) < TOL


@torch.no_grad()
def test_pseudo_aggregator_neck_matches_real_class(gold):
    from oracle.anchor import PseudoAggregator, SimpleFPN
    g = gold['pseudo_neck']
    neck = nn.Module()
    neck.feature_aggregator, neck.feature_spliter = PseudoAggregator(256, 512, 256), SimpleFPN()
    load(neck, g)
    x = rnd(g['x'])
    agg = neck.feature_aggregator((x,))
    assert err(agg, g['agg']) < TOL
    for a, b in zip(neck.feature_spliter(agg), g['outs']):
        assert err(a[:, ::8], b) < TOL


@torch.no_grad()
def test_anchor_mask_head_matches_real_class(gold):
    from oracle.anchor import MaskHead
    g = gold['anchor_mask_head']
    m = load(MaskHead(), g)
    emb = rnd(g['emb'])
    pe = rnd(g['pe']).repeat(emb.shape[0], 1, 1, 1)
    low, iou, _ = m(rnd(g['feats']), emb, pe, g['roi_img'])
    assert low.shape == g['low_res_masks'].shape
    assert err(low, g['low_res_masks']) < 1e-4 and err(iou, g['iou']) < 1e-4


@torch.no_grad()
def test_query_head_matches_real_class(gold):
    from oracle.query import QueryHead
    g = gold['query_head']
    m = load(QueryHead(g['num_classes'], g['num_queries']), g)
    xs = [rnd(s) for s in g['xs']]
    emb = rnd(g['emb'])
    pe = rnd(g['pe']).repeat(g['batch'], 1, 1, 1)
    # pixel decoder (msdeformattn_pixel_decoder.py:144-246)
    mf, mem = m.pixel_decoder(xs)
    assert err(mf[:, ::4, ::4, ::4], g['mask_features']) < 1e-4
    for a, b in zip(mem, g['memories']):
        assert err(a[:, ::2], b) < 1e-4
    # the whole head: all 7 class predictions and auxiliary masks, the last SAM-decoder result
    cls, mask, tr = m(xs, emb, pe)
    assert len(tr['cls_pred_all']) == len(g['cls_pred_all']) == 7
    for a, b in zip(tr['cls_pred_all'], g['cls_pred_all']):
        assert err(a, b) < 1e-4
    for a, b in zip(tr['mask_pred_plus_all'], g['mask_pred_plus_all']):
        assert err(a[:, :, ::4, ::4], b) < 2e-4
    assert err(cls, g['cls_pred']) < 1e-4
    assert err(mask[:, :, ::2, ::2], g['mask_pred']) < 2e-4
    assert err(tr['mask_pred_plus'][:, :, ::2, ::2], g['mask_pred_plus']) < 2e-4
    # the FIRST `_forward_head` call's SAM result (the oracle skips it at inference: models.py:644-646 only reads the last)
    qf = m.query_feat.weight.unsqueeze(0).repeat((g['batch'], 1, 1))
    _, first, _, _, _ = m._forward_head(qf, mf, mem[0].shape[-2:], emb, pe, True)
    assert err(first[:, :, ::4, ::4], g['mask_pred_first']) < 2e-4
    # one decoder layer on its own (mask2former_layers.py:73-135)
    d = g['dec_layer']
    lay = m.transformer_decoder.layers[d['index']]
    kv = rnd(d['kv'])
    mask_b = rnd(d['mask']) < 0.0
    mask_b[:, :, 0] = False
    out = lay(rnd(d['q']), kv, kv, rnd(d['qpos']), rnd(d['kpos']), mask_b)
    assert err(out, d['out']) < 1e-4


@torch.no_grad()
@pytest.mark.parametrize('name', ['levels2', 'levels4_proj', 'no_sincos', 'no_decoder_plus', 'multimask'])
def test_query_head_option_branches_match_real_class(name):
    """The RSMask2FormerHead branches no shipped config selects (num_transformer_feat_level != 3,
    enforce_decoder_input_project, with_sincos=False, decoder_plus=False, multimask_output=True: [B, 3 Nq, h, w] masks)
    against the REAL class run with those arguments (tests/golden/make_golden_query_options.py): state_dict layout and
    every stage's outputs."""
    from oracle.query import QueryHead
    allg = torch.load(os.path.join(os.path.dirname(GOLD), 'reference_vectors_query_options.pt'), weights_only=False)
    g = allg[name]
    if name == 'multimask':
        # with decoder_plus=False the real class fails in its first decoder layer (the folded masks as attention mask)
        assert allg['multimask_no_decoder_plus']['raised']['type'] == 'RuntimeError'
        with pytest.raises(ValueError):
            QueryHead(g['num_classes'], g['num_queries'], multimask_output=True, decoder_plus=False)
    m = load(QueryHead(g['num_classes'], g['num_queries'], **g['head_kwargs']), g)
    xs = [rnd(s) for s in g['xs']]
    emb = rnd(g['emb'])
    pe = rnd(g['pe']).repeat(g['batch'], 1, 1, 1)
    mf, mem = m.pixel_decoder(xs)
    assert len(mem) == g['n_memories']
    assert err(mf[:, ::4, ::4, ::4], g['mask_features']) < 1e-4
    for a, b in zip(mem, g['memories']):
        assert err(a[:, ::2], b) < 1e-4
    cls, mask, tr = m(xs, emb, pe)
    assert len(tr['cls_pred_all']) == len(g['cls_pred_all']) == 7
    for a, b in zip(tr['cls_pred_all'], g['cls_pred_all']):
        assert err(a, b) < 1e-4
    stage = g.get('mask_pred_plus_all', g.get('mask_pred_all'))     # what each stage cuts its attention mask from
    for a, b in zip(tr['mask_pred_plus_all'], stage):
        assert err(a[:, :, ::4, ::4], b) < 2e-4
    assert mask.shape[:2] == g['mask_pred'].shape[:2] and err(mask[:, :, ::2, ::2], g['mask_pred']) < 2e-4


@torch.no_grad()
def test_vitsam_forward_matches_real_class(gold):
    from oracle.vitsam import ViTSAM
    g = gold['vitsam']
    m = load(ViTSAM('base', g['img_size'], lora=False), g)
    y = m(rnd(g['x']))
    assert isinstance(y, tuple) and len(y) == 1
    assert err(y[0][:, ::2], g['out']) < 1e-4


# ----------------------------------------------------------------------------- SAMSeg sibling model (mask branch)
@torch.no_grad()
def test_fcn_mask_head_and_paste_match_real_file():
    from oracle import samseg
    g = torch.load(os.path.join(os.path.dirname(GOLD), 'reference_vectors_samseg.pt'), weights_only=False)
    h = g['fcn_head']
    m = load(samseg.FCNMaskHead(num_classes=10), h)
    assert err(m(rnd(h['x'])), h['out']) < 1e-4
    p = g['paste']
    got = samseg.paste_masks(p['probs'], p['boxes'], *p['img_hw'])
    assert err(got, p['out']) < 1e-6
    for c in g['predict_single']:
        masks, bb, _ = samseg.fcn_predict_single(c['logits'], c['boxes'].clone(), c['labels'], c['meta'], 0.5, c['rescale'])
        assert masks.shape == c['masks'].shape and torch.equal(masks, c['masks'])
        assert torch.allclose(bb, c['boxes_out'])


@torch.no_grad()
def test_mask2former_head_matches_real_class():
    """SAMSegMask2Former's panoptic head: the oracle against the REAL mmdet Mask2FormerHead (mask2former_head.py) run on
    the reference's samseg-mask2former config (feat 256, 9 decoder layers, FFN 2048 / 1024, PE num_feats 128)."""
    from oracle import samseg
    g = torch.load(os.path.join(os.path.dirname(GOLD), 'reference_vectors_samseg.pt'), weights_only=False)['m2f_head']
    m = load(samseg.Mask2FormerHead(g['num_classes'], g['num_queries']), g)
    xs = [rnd(s) for s in g['xs']]
    mf, mem = m.pixel_decoder(xs)
    assert err(mf[:, ::4, ::2, ::2], g['mask_features']) < 1e-4
    for a, b in zip(mem, g['memories']):
        assert err(a[:, ::4], b) < 1e-4
    cls, mask, tr = m(xs)
    assert len(tr['cls_pred_all']) == len(g['cls_pred_all']) == 10
    for a, b in zip(tr['cls_pred_all'], g['cls_pred_all']):
        assert err(a, b) < 1e-4
    for a, b in zip(tr['mask_pred_all'], g['mask_pred_all']):
        assert err(a[:, :, ::2, ::2], b) < 2e-4
    assert err(mask, g['mask_pred']) < 2e-4


@torch.no_grad()
def test_samdet_resnet_fpn_and_predict_glue_match_real_files():
    """oracle/samdet.py against the REAL mmdet ResNet-50 / FPN classes and the REAL SAMDet.predict (models.py:1155-1213)
    run in the build container (tests/golden/make_golden_samdet.py)."""
    import numpy as np
    from oracle import samdet
    from rsprompter_amd.synth import synth_state_dict
    g = torch.load(os.path.join(os.path.dirname(GOLD), 'reference_vectors_samdet.pt'), weights_only=False)
    r = g['resnet_fpn']
    net = load(samdet.ResNet(50), dict(keys=r['backbone_keys'], seed=r['seed'][0]))
    neck = load(samdet.FPN(), dict(keys=r['neck_keys'], seed=r['seed'][1]))
    c = net(rnd(r['x']))
    p = neck(c)
    cs, fs = r['strides']
    assert [tuple(t.shape) for t in c] == r['c_shapes'] and [tuple(t.shape) for t in p] == r['p_shapes']
    for got, want in zip(c, r['c']):
        assert err(got[:, ::cs], want) < 1e-4 * max(1.0, float(want.abs().max()))
    for got, want in zip(p, r['p']):
        assert err(got[:, ::fs], want) < 1e-4 * max(1.0, float(want.abs().max()))
    s = g['samdet_predict']
    sam = samdet.build_sam_model('base')
    sam.load_state_dict(synth_state_dict(sam, s['seed']))
    imgs = rnd(s['imgs'])
    for img, meta, boxes, packed, shape in zip(imgs, s['metas'], s['boxes'], s['masks'], s['shapes']):
        if boxes.shape[0] == 0:
            assert shape[0] == 0
            continue
        sf = boxes.new_tensor(meta['scale_factor']).repeat((1, 2))
        masks, _ = samdet.sam_box_masks(sam, img, boxes * sf, meta)
        want = torch.from_numpy(np.unpackbits(packed.numpy())[:int(np.prod(shape))].reshape(shape)).bool()
        assert tuple(masks.shape) == shape
        assert float((masks != want).float().mean()) < 1e-4
