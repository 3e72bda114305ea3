"""-m gpu: RSPrompterQuery (HIP) against the CPU oracle on identical seeded weights and inputs
(ViT-B, 1 x 1024^2 synthetic tile, 30 queries / 1 class = the rsprompter_query-ssdd tree)."""
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.quick]      # quick: the kernel-level tier (`-m "gpu and quick"`, < 2 min)

MEAN = [123.675, 116.28, 103.53]
STD = [58.395, 57.12, 57.375]
NQ, NC = 30, 1


def _cl(x, dev):
    return x.to(dev).contiguous(memory_format=torch.channels_last)


def _maxerr(a, b):
    return float((a.detach().float().cpu() - b.detach().float().cpu()).abs().max())


@pytest.fixture(scope='module')
def setup(dev):
    import warnings
    import rsprompter_amd as ra
    from oracle import glue
    from oracle.query import QueryOracle
    from rsprompter_amd.default_configs import rsprompter_query
    from rsprompter_amd.synth import synth_images, synth_metas, synth_state_dict
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = ra.build_model(rsprompter_query('base', NC, (NQ, 5)))
    oracle = QueryOracle('base', NC, NQ, max_per_image=NQ)
    sd = synth_state_dict(oracle, seed=0)
    oracle.load_state_dict(sd)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    imgs, metas = synth_images(1), synth_metas(1)
    x = glue.data_preprocess(imgs, MEAN, STD, True, 32)
    results, trace = oracle.predict(x, metas)
    return dict(model=model, oracle=oracle, imgs=imgs, metas=metas, x=x, results=results, trace=trace)


def test_query_kernels_unit(dev):
    """GroupNorm / bilinear resize / MSDeformAttn sampling / masked attention against torch."""
    import torch.nn.functional as F
    from oracle.query import MSDeformAttn
    from rsprompter_amd import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 128, 20, 24, generator=g) * 2 + 0.5
    w, b = torch.randn(128, generator=g), torch.randn(128, generator=g)
    ref = F.group_norm(x.double(), 32, w.double(), b.double(), 1e-5)
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev)
    got = ops.groupnorm(xh, w.to(dev), b.to(dev), 32, relu=True)
    assert _maxerr(got.permute(0, 3, 1, 2), F.relu(ref)) < 2e-5
    add = torch.randn(2, 20, 24, 128, generator=g)
    got = ops.groupnorm(xh, w.to(dev), b.to(dev), 32, add=add.to(dev))
    assert _maxerr(got.permute(0, 3, 1, 2), ref + add.permute(0, 3, 1, 2).double()) < 2e-5
    up = ops.resize_bilinear(xh, (40, 48))
    assert _maxerr(up.permute(0, 3, 1, 2), F.interpolate(x, size=(40, 48), mode='bilinear', align_corners=False)) < 1e-5
    # MSDeformAttn core (module init draws from the global RNG: pin it)
    torch.manual_seed(1234)
    m = MSDeformAttn()
    shapes = [(4, 4), (8, 8), (16, 16)]
    ntok = sum(h * w_ for h, w_ in shapes)
    q, pos = torch.randn(2, ntok, 128, generator=g), torch.randn(2, ntok, 128, generator=g)
    refp = torch.rand(ntok, 2, generator=g)
    with torch.no_grad():
        m.sampling_offsets.weight.mul_(20)          # offsets large enough to leave the maps
        qq = q + pos
        value = m.value_proj(q)
        ow = torch.cat([m.sampling_offsets(qq), m.attention_weights(qq)], -1)
        want = (m(q, pos, refp[None, :, None].repeat(2, 1, 3, 1), torch.tensor(shapes)) - q).double()
    got = ops.msdeform_attn(value.reshape(-1, 128).contiguous().to(dev), ow.reshape(-1, 288).contiguous().to(dev),
                            refp.to(dev), 2, ntok, shapes)
    # the kernel stops before output_proj: apply it here (fp64) instead of inverting it out of the oracle's result
    got_full = got.cpu().double().view(2, ntok, 128) @ m.output_proj.weight.double().t() + m.output_proj.bias.double()
    assert _maxerr(got_full, want) < 1e-4
    # masked attention incl. rows whose first key tiles are fully masked
    Bq, Tq, Tk, nh, dh = 2, 30, 200, 8, 16
    D = nh * dh
    qx, kx, vx = (torch.randn(Bq, t, D, generator=g) for t in (Tq, Tk, Tk))
    mask = torch.rand(Bq, Tq, Tk, generator=g) < 0.6
    mask[:, :, :130] |= torch.rand(Bq, Tq, 1, generator=g) < 0.5      # whole leading tiles blocked for half the rows
    mask[:, :, -1] = False                                            # never fully blocked
    qh = qx.double().view(Bq, Tq, nh, dh).transpose(1, 2)
    kh = kx.double().view(Bq, Tk, nh, dh).transpose(1, 2)
    vh = vx.double().view(Bq, Tk, nh, dh).transpose(1, 2)
    s = (qh * dh ** -0.5) @ kh.transpose(-1, -2)
    s = s.masked_fill(mask[:, None], float('-inf'))
    want = (s.softmax(-1) @ vh).transpose(1, 2).reshape(Bq, Tq, D)
    out = torch.empty(Bq, Tq, D, device=dev)
    ops.attention(qx.to(dev), kx.to(dev), vx.to(dev), out, B=Bq, nh=nh, dh=dh, Tq=Tq, Tk=Tk, scale=dh ** -0.5,
                  q_strides=(Tq * D, D, dh), k_strides=(Tk * D, D, dh), v_strides=(Tk * D, D, dh),
                  o_strides=(Tq * D, D, dh), mask=mask.to(torch.uint8).to(dev))
    assert _maxerr(out, want) < 2e-5


@pytest.mark.parametrize('shapes,dim', [([(6, 5)], 128), ([(4, 4), (8, 6)], 128), ([(3, 3), (4, 6), (8, 8), (16, 12)], 128),
                                        ([(2, 2), (3, 3), (4, 4), (6, 6), (8, 8)], 256)])
def test_msdeform_attn_level_counts(dev, shapes, dim):
    """MultiScaleDeformableAttention with 1, 2, 4 and 5 levels (num_transformer_feat_level = the pixel decoder's num_levels
    != 3, mask2former_head.py:103-135; msda_kernel is templated on the level count) against the oracle's restatement,
    offsets large enough to leave the maps; 8 heads x 16 and x 32."""
    from oracle.query import MSDeformAttn
    from rsprompter_amd import ops
    L = len(shapes)
    g = torch.Generator().manual_seed(10 + L)
    torch.manual_seed(77 + L)
    m = MSDeformAttn(dim=dim, levels=L)
    ntok = sum(h * w_ for h, w_ in shapes)
    q, pos = torch.randn(2, ntok, dim, generator=g), torch.randn(2, ntok, dim, generator=g)
    refp = torch.rand(ntok, 2, generator=g)
    with torch.no_grad():
        m.sampling_offsets.weight.mul_(20)
        qq = q + pos
        value = m.value_proj(q)
        ow = torch.cat([m.sampling_offsets(qq), m.attention_weights(qq)], -1)              # [2, ntok, 8 * L * 4 * 3]
        want = (m(q, pos, refp[None, :, None].repeat(2, 1, L, 1), torch.tensor(shapes)) - q).double()
    got = ops.msdeform_attn(value.reshape(-1, dim).contiguous().to(dev), ow.reshape(-1, 96 * L).contiguous().to(dev),
                            refp.to(dev), 2, ntok, shapes, head_dim=dim // 8)
    got_full = got.cpu().double().view(2, ntok, dim) @ m.output_proj.weight.double().t() + m.output_proj.bias.double()
    assert _maxerr(got_full, want) < 1e-4


def test_pixel_decoder_given_oracle_fpn(setup, dev):
    m, tr = setup['model'], setup['trace']
    feats = [_cl(f, dev) for f in tr['fpn']]
    mf, mem = m.panoptic_head.pixel_decoder(feats)
    e_mf = _maxerr(mf, tr['mask_features'])
    e_mem = [_maxerr(a, b) for a, b in zip(mem, tr['memory'])]
    print('pixel decoder: mask_feature err %.2e (range %.1f), memories %s' %
          (e_mf, float(tr['mask_features'].abs().max()), ['%.2e' % e for e in e_mem]))
    assert e_mf < 1e-3 and max(e_mem) < 1e-3


def test_query_head_given_oracle_features(setup, dev):
    m, tr = setup['model'], setup['trace']
    feats = [_cl(f, dev) for f in tr['fpn']]
    emb, ipe = _cl(tr['image_embeddings'], dev), _cl(tr['image_pe'], dev)
    cls, mask_pred, t = m.panoptic_head(feats, None, emb, ipe)
    for i, (a, b) in enumerate(zip(t['attn_masks'], tr['attn_masks'])):
        ref = b.view(1, 8, NQ, -1)[:, 0]
        mism = float((a.cpu().bool() != ref).float().mean())
        print(f'layer {i}: attention-mask mismatch fraction {mism:.2e}')
        assert mism < 1e-3
    e_q = max(_maxerr(a.view(1, NQ, -1), b) for a, b in zip(t['query_feats'], tr['query_feats'][1:]))
    e_cls = _maxerr(cls, tr['cls_pred'])
    e_mpp = _maxerr(t['mask_pred_plus'], tr['mask_pred_plus'])
    e_sp = _maxerr(t['sparse_embeddings'], tr['sparse_embeddings'][:, 0])
    e_mask = _maxerr(mask_pred, tr['mask_pred'])
    print('query head: query_feat %.2e cls %.2e mask_pred_plus %.2e sparse %.2e SAM mask logits %.2e (range %.2f)' %
          (e_q, e_cls, e_mpp, e_sp, e_mask, float(tr['mask_pred'].abs().max())))
    assert e_q < 1e-3 and e_cls < 1e-3 and e_mpp < 2e-3 and e_sp < 1e-3 and e_mask < 1e-3
    # the SAM decoder in chunks of prompt sets (BASELINE configs[2]: 1600 per step, above the folded attention's 1023): every
    # query has its own dense-prompted source here (src_rows + identity map), forced at 7 per chunk -- bit-identical
    dec = m.panoptic_head.mask_decoder.mask_decoder
    dec.max_prompt_sets = 7
    try:
        cls2, mask2, _ = m.panoptic_head(feats, None, emb, ipe)
    finally:
        dec.max_prompt_sets = None
    assert torch.equal(mask2, mask_pred) and torch.equal(cls2, cls)


def test_fusion_head_given_oracle_logits(setup, dev):
    from rsprompter_amd.query_heads import LazyUpsampledMasks
    from rsprompter_amd.structures import DetDataSample
    m, tr = setup['model'], setup['trace']
    samples = [DetDataSample(metainfo=dict(mm)) for mm in setup['metas']]
    res = m.panoptic_fusion_head.predict(tr['cls_pred'].to(dev), LazyUpsampledMasks(tr['mask_pred'].to(dev), (1024, 1024)),
                                         samples, rescale=True)
    r, ref = res[0]['ins_results'], setup['results'][0]
    assert torch.equal(r.query_indices.cpu().long(), ref['query_indices'])       # bit-exact indices
    assert torch.equal(r.labels.cpu(), ref['labels'])
    assert _maxerr(r.scores, ref['scores']) < 1e-5
    assert torch.equal(r.bboxes.cpu(), ref['bboxes'])
    mism = float((r.masks.cpu() != ref['masks']).float().mean())
    print('fusion head: mask mismatch %.2e' % mism)
    assert mism < 1e-5


def test_query_end_to_end(setup, dev):
    from rsprompter_amd.structures import DetDataSample
    m = setup['model']
    samples = [DetDataSample(metainfo=dict(mm)) for mm in setup['metas']]
    out = m.test_step(dict(inputs=[i.to(dev) for i in setup['imgs']], data_samples=samples))
    pi, ref = out[0].pred_instances, setup['results'][0]
    assert pi.masks.dtype == torch.bool and tuple(pi.masks.shape) == tuple(ref['masks'].shape)
    same = pi.query_indices.cpu().long() == ref['query_indices']
    same_q = float(same.float().mean())
    mism = float((pi.masks.cpu()[same] != ref['masks'][same]).float().mean())
    print('query e2e: query-index agreement %.3f, score err %.2e, mask mismatch %.2e' %
          (same_q, _maxerr(pi.scores, ref['scores']), mism))
    # free-running pipeline: an entry may only differ from the oracle's where the oracle's own score sits on the
    # top-k cut-off or ties with a neighbour up to fp32 noise; the stage-wise tests are the index-exactness gates
    sc = ref['scores']
    cut = float(sc.min())
    if not bool(same.all()):
        near_tie = torch.zeros_like(same)
        d = (sc[:, None] - sc[None, :]).abs() + torch.eye(len(sc)) * 1e9
        near_tie = (d.min(1).values < 5e-5) | ((sc - cut).abs() < 5e-5)
        assert bool(near_tie[~same].all()) and int((~same).sum()) <= 4
    assert _maxerr(pi.scores, ref['scores']) < 1e-4 and mism < 1e-3


def test_fusion_head_rescale_paths(dev):
    """RSMaskFormerFusionHead.predict with padded / rescaled images (BASELINE.json configs[4] "WHU-shape": 512 px tiles
    resized x2; and a non-square 1.5x case): crop of the padding, second bilinear resize of the LOGITS to ori_shape,
    then instance_postprocess -- against oracle.query.fusion_predict (pinned on the reference's own code)."""
    import torch.nn.functional as F
    from oracle import query as oq
    from rsprompter_amd.query_heads import LazyUpsampledMasks, RSMaskFormerFusionHead
    from rsprompter_amd.structures import DetDataSample
    g = torch.Generator().manual_seed(21)
    nq, nc, k = 24, 2, 10
    head = RSMaskFormerFusionHead(num_things_classes=nc, num_stuff_classes=0,
                                  test_cfg=dict(panoptic_on=False, semantic_on=False, instance_on=True, max_per_image=k))
    for meta in (dict(img_shape=(1024, 1024), ori_shape=(512, 512), scale_factor=(2.0, 2.0), batch_input_shape=(1024, 1024)),
                 dict(img_shape=(900, 600), ori_shape=(600, 400), scale_factor=(1.5, 1.5), batch_input_shape=(1024, 1024))):
        cls = torch.randn(1, nq, nc + 1, generator=g) * 2
        blob = F.avg_pool2d(torch.randn(1, nq, 256, 256, generator=g), 9, 1, 4) * 12      # smooth logits, both signs
        up = F.interpolate(blob, size=(1024, 1024), mode='bilinear', align_corners=False)
        ref = oq.fusion_predict(cls, up, [meta], nc, k, rescale=True)[0]
        res = head.predict(cls.to(dev), LazyUpsampledMasks(blob.to(dev), (1024, 1024)), [DetDataSample(metainfo=dict(meta))],
                           rescale=True)[0]['ins_results']
        assert torch.equal(res.query_indices.cpu().long(), ref['query_indices'])
        assert torch.equal(res.labels.cpu(), ref['labels'])
        assert tuple(res.masks.shape) == tuple(ref['masks'].shape)
        assert _maxerr(res.scores, ref['scores']) < 1e-5
        mism = float((res.masks.cpu() != ref['masks']).float().mean())
        assert mism < 1e-5, mism
        assert float((res.bboxes.cpu() - ref['bboxes']).abs().max()) <= 1.0       # a flipped boundary pixel moves a box edge by 1


def test_tensor_mode_returns_raw_head_outputs(setup, dev):
    """`forward(mode='tensor')` (maskformer.py:153-170): raw head outputs of the last decoder stage."""
    from rsprompter_amd.structures import DetDataSample
    m, tr = setup['model'], setup['trace']
    samples = [DetDataSample(metainfo=dict(mm)) for mm in setup['metas']]
    cls_list, mask_list, mpp_list = m(setup['x'].to(dev), samples, mode='tensor')
    assert len(cls_list) == len(mask_list) == 1
    assert _maxerr(cls_list[0], tr['cls_pred']) < 1e-3 and _maxerr(mask_list[0], tr['mask_pred']) < 1e-3
    assert _maxerr(mpp_list[0], tr['mask_pred_plus']) < 2e-3
