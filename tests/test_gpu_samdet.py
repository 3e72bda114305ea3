"""-m gpu: `SAMDet` (models.py:1061-1215) on the HIP kernels against its CPU oracle (oracle/samdet.py; ResNet /
FPN / the predict glue pinned on the real resnet.py / fpn.py / models.py by tests/golden/make_golden_samdet.py)."""
import os
import sys
import warnings

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import torch_ops_mock as mock  # noqa: E402   (plain-torch fp32 statements of the ops, the checker here)
from _match import match_detections  # noqa: E402

MEAN = [123.675, 116.28, 103.53]
STD = [58.395, 57.12, 57.375]


def _err(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())


def test_resnet_leaf_kernels(dev):
    """stem conv 7x7 s2 (+ folded BN + ReLU), MaxPool2d(3, 2, 1), FPN nearest top-down add -- odd sizes included."""
    from rsprompter_amd import ops
    g = torch.Generator().manual_seed(5)
    for (B, H, W) in ((2, 75, 106), (1, 64, 64), (1, 33, 257)):
        x = torch.randn(B, 3, H, W, generator=g)
        w = torch.randn(147, 64, generator=g) / 12
        b = torch.randn(64, generator=g) * 0.1
        got = ops.resnet_stem(x.to(dev), w.to(dev), b.to(dev))
        want = mock.resnet_stem(x, w, b)
        assert got.shape == want.shape
        assert _err(got, want) < 2e-5 * max(1.0, float(want.abs().max()))
        p = ops.maxpool_nhwc(got, 3, 2, 1)
        wp = mock.maxpool_nhwc(want, 3, 2, 1)
        assert p.shape == wp.shape and _err(p, wp) < 2e-5 * max(1.0, float(want.abs().max()))
    for (h, w, H, W) in ((5, 7, 10, 14), (3, 4, 5, 7), (16, 16, 32, 32), (3, 5, 7, 11)):
        src = torch.randn(2, h, w, 256, generator=g)
        dst = torch.randn(2, H, W, 256, generator=g)
        got = ops.upsample_nearest_add_(dst.clone().to(dev), src.to(dev))
        want = mock.upsample_nearest_add_(dst.clone(), src)
        assert torch.equal(got.cpu(), want)


def test_gemm_relu_after_residual(dev):
    """RSP_ACT_RELU_POST = relu(A W^T + bias + res) on every GEMM path a Bottleneck's conv3 can take."""
    from rsprompter_amd import ops
    g = torch.Generator().manual_seed(6)
    for (M, N, K) in ((300, 256, 64), (5000, 256, 64), (70000, 256, 64), (70000, 512, 128), (3000, 1024, 256),
                      (40, 2048, 512)):
        a = torch.randn(M, K, generator=g)
        w = torch.randn(N, K, generator=g) / K ** 0.5
        b = torch.randn(N, generator=g) * 0.1
        r = torch.randn(M, N, generator=g)
        got = ops.gemm(a.to(dev), ops.PackedWeight(w.to(dev), b.to(dev)), res=r.to(dev), act=ops.ACT_RELU_POST)
        want = F.relu(a.double() @ w.double().t() + b.double() + r.double())
        assert float((got.cpu() == 0).float().mean()) > 0.2          # the ReLU really acted
        assert _err(got, want) < 2e-5, (M, N, K)
    # strided 1x1 convolution (the downsample branch) and the 3x3 stride-2 convolution
    x = torch.randn(2, 19, 27, 256, generator=g)
    w1 = torch.randn(512, 256, generator=g) / 16
    got = ops.gemm(x.to(dev), ops.PackedWeight(w1.to(dev), None), conv=(1, 2, 0))
    want = F.conv2d(x.permute(0, 3, 1, 2), w1.view(512, 256, 1, 1), stride=2).permute(0, 2, 3, 1).reshape(-1, 512)
    assert got.shape == want.shape and _err(got, want) < 2e-5
    w3 = torch.randn(128, 3, 3, 128, generator=g) / 34
    x3 = torch.randn(2, 19, 27, 128, generator=g)
    got = ops.gemm(x3.to(dev), ops.PackedWeight(w3.reshape(128, -1).to(dev), None), act=ops.ACT_RELU, conv=(3, 2, 1))
    want = F.relu(F.conv2d(x3.permute(0, 3, 1, 2), w3.permute(0, 3, 1, 2), stride=2, padding=1)).permute(0, 2, 3, 1)
    assert _err(got, want.reshape(-1, 128)) < 2e-5


def test_box_prompt_mask_post_and_scale_boxes(dev):
    from oracle import hf_sam
    from rsprompter_amd import ops
    g = torch.Generator().manual_seed(7)
    shared = hf_sam.build_positional_embedding('base')
    boxes = torch.rand(37, 4, generator=g) * 1000
    boxes[:, 2:] = boxes[:, :2] + torch.rand(37, 2, generator=g) * 300
    tl, br = torch.randn(1, 256, generator=g), torch.randn(1, 256, generator=g)
    G = shared.positional_embedding.detach()
    got = ops.sam_embed_boxes(boxes.to(dev), G.to(dev), tl.to(dev), br.to(dev), (1024, 1024))
    # HF's own expression (modeling_sam.py:647-656 over :552-566)
    with torch.no_grad():
        ce = shared((boxes + 0.5).reshape(1, -1, 2, 2), (1024, 1024))[0]
    ce[:, 0, :] += tl[0]
    ce[:, 1, :] += br[0]
    # HF initialises the Gaussian matrix with scale = hidden_size // 2, so the sin / cos arguments reach thousands of
    # radians here: a few ulp of the fp32 argument (fused or unfused 2-term dot product) is all two evaluations can agree to
    arg = float(((2 * (boxes + 0.5).reshape(-1, 2) / 1024 - 1) @ G).abs().max()) * 6.2832
    tol = 8 * 2.0 ** -24 * arg + 1e-5
    print(f'box prompt: max |argument| {arg:.0f} rad, tolerance {tol:.1e}, err {_err(got, ce):.1e}')
    assert got.shape == ce.shape and _err(got, ce) < tol
    assert _err(got, mock.sam_embed_boxes(boxes, G, tl, br, (1024, 1024))) < tol
    G1 = torch.randn(2, 128, generator=g)                                 # checkpoint-like magnitude (std 1)
    got = ops.sam_embed_boxes(boxes.to(dev), G1.to(dev), tl.to(dev), br.to(dev), (1024, 1024))
    assert _err(got, mock.sam_embed_boxes(boxes, G1, tl, br, (1024, 1024))) < 2e-5
    for (k, img, crop, out) in ((3, (768, 1024), (768, 1024), (300, 400)), (2, (1024, 1024), (1024, 1024), (512, 512)),
                                (1, (1024, 1024), (1000, 900), (333, 301)), (4, (1024, 1024), (1024, 1024), (1024, 1024))):
        low = F.avg_pool2d(torch.randn(k, 1, 256, 256, generator=g), 9, 1, 4)[:, 0] * 20
        m, v = ops.mask_post_logits(low.to(dev), img, crop, out, 0.0, want_val=True)
        wm, wv = mock.mask_post_logits(low, img, crop, out, 0.0, want_val=True)
        assert _err(v, wv) < 1e-4
        mism = (m.cpu() != wm)
        assert float(mism.float().mean()) < 1e-4 and bool((wv[mism].abs() < 1e-4).all())
    b = torch.rand(11, 4, generator=g) * 500
    f = (1 / 2.56, 1 / 1.7, 1 / 2.56, 1 / 1.7)
    assert torch.equal(ops.scale_boxes(b.to(dev), f).cpu(), b * torch.tensor(f, dtype=torch.float32))


def test_box_coder_branches_on_the_real_heads_vectors(dev):
    """rsp_rpn_decode / rsp_bbox_post with the DeltaXYWHBBoxCoder branches no RSPrompter config uses (target_means != 0,
    target_stds != 1 in the RPN, clip_border=False, add_ctr_clamp: delta_xywh_bbox_coder.py:264-361) on vectors of the REAL
    RPNHead / BBoxHead `_predict_by_feat_single` (tests/golden/make_golden_coder.py): every decode step rounds as the
    reference's eager fp32 expression, so the kept sets are the reference's and boxes agree to the decode's rounding."""
    from rsprompter_amd import ops
    from rsprompter_amd.anchor_heads import AnchorGenerator, DeltaXYWHBBoxCoder
    g = torch.load(os.path.join(HERE, 'golden', 'reference_vectors_coder.pt'), weights_only=False)
    gen = AnchorGenerator(strides=[4, 8, 16, 32, 64], ratios=[0.5, 1.0, 2.0], scales=[4, 8])
    base = torch.stack(gen.base_anchors, 0)
    for name, kw in g['coders'].items():
        coder = DeltaXYWHBBoxCoder(**kw)
        # ---- RPN
        c = g['rpn_predict_single'][name]
        A, LD = 6, 32
        heads = []
        for cl, rg in zip(c['cls'], c['reg']):
            _, H, W = cl.shape
            h = torch.zeros((H, W, LD))
            h[..., :A] = cl.permute(1, 2, 0)
            h[..., A:5 * A] = rg.permute(1, 2, 0)
            heads.append(h.reshape(H * W, LD).to(dev).contiguous())
        sel = ops.RpnSelector(base, [4, 8, 16, 32, 64], c['nms_pre'], c['max_per_img'], c['iou_thr'], c['min_bbox_size'],
                              coder, dev)
        out = sel(heads, c['sizes'], LD, torch.tensor([c['img_shape']], dtype=torch.float32, device=dev))
        k = int(out['count'][0])
        assert k == c['scores'].shape[0], (name, k, c['scores'].shape)
        gb, gs = out['boxes'][0, :k].cpu(), out['scores'][0, :k].cpu()
        assert _err(gs, c['scores']) < 1e-6, name
        bad = ((gb - c['bboxes']).abs().amax(1) > 2e-3).nonzero()[:, 0].tolist()
        assert len(bad) <= 4, (name, len(bad))                 # rows may trade places at (near-)equal scores only
        for i in bad:
            d = (c['bboxes'] - gb[i]).abs().amax(1)
            j = int(d.argmin())
            assert float(d[j]) < 2e-3 and abs(float(c['scores'][j]) - float(gs[i])) < 1e-6, (name, i, j)
        # ---- R-CNN box head
        c = g['bbox_head_predict_single'][name]
        n, nc = c['roi'].shape[0], c['num_classes']
        LDb = (5 * nc + 1 + 3) // 4 * 4
        head = torch.zeros((n, LDb))
        head[:, :nc + 1] = c['cls_score']
        head[:, nc + 1:5 * nc + 1] = c['bbox_pred']
        out = ops.bbox_post(head.to(dev), LDb, c['roi'].to(dev), torch.tensor([0, n]),
                            torch.tensor([c['img_shape']], dtype=torch.float32, device=dev), nc, c['score_thr'], coder,
                            c['iou_thr'], c['max_per_img'])
        k = int(out['count'][0])
        assert k == c['labels'].shape[0], name
        pairs = match_detections(out['boxes'][0, :k].cpu(), out['scores'][0, :k].cpu(), out['ids'][0, :k].cpu().long(),
                                 c['bboxes'], c['scores'], c['labels'])
        assert len(pairs) == k, name
        ii = torch.tensor([i for i, _ in pairs]); jj = torch.tensor([j for _, j in pairs])
        assert _err(out['boxes'][0, :k].cpu()[ii], c['bboxes'][jj]) < 2e-3, name
        assert _err(out['scores'][0, :k].cpu()[ii], c['scores'][jj]) < 1e-6, name
        assert int((ii != jj).sum()) <= 4, name                # rank swaps only at score ties
        print(f'coder {name}: RPN {c["labels"].shape[0]} / head detections match the real heads')


def test_rpn_softmax_objectness_on_the_real_heads_vectors(dev):
    """RPNHead(loss_cls.use_sigmoid=False): two scores [fg, bg] per anchor, `softmax(-1)[:, :-1]` (anchor_head.py:73-77,
    rpn_head.py:193-200) = sigmoid(fg - bg).  The head folds the difference into its packed 1x1 convolution (column a of
    the head GEMM = W[2a] - W[2a + 1]) and keeps the raw scores for forward(); the selection on the real RPNHead's vectors
    (tests/golden/make_golden_heads.py, use_sigmoid_cls=False) then runs the sigmoid kernels unchanged."""
    from rsprompter_amd import ops
    from rsprompter_amd.anchor_heads import AnchorGenerator, DeltaXYWHBBoxCoder, RPNHead
    from rsprompter_amd.synth import synth_state_dict
    ag = dict(type='AnchorGenerator', strides=[4, 8, 16, 32, 64], ratios=[0.5, 1.0, 2.0], scales=[4, 8])
    m = RPNHead(in_channels=32, feat_channels=32, anchor_generator=ag, loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False),
                test_cfg=dict(nms_pre=300, max_per_img=200, nms=dict(type='nms', iou_threshold=0.7), min_bbox_size=0))
    A = m.num_base_priors
    assert not m.use_sigmoid_cls and m.cls_out_channels == 2 and tuple(m.rpn_cls.weight.shape) == (2 * A, 32, 1, 1) and m.LD == 64
    assert RPNHead(in_channels=32, feat_channels=32, anchor_generator=ag).use_sigmoid_cls          # the default loss_cls
    assert not RPNHead(in_channels=32, feat_channels=32, anchor_generator=ag, loss_cls=dict(type='CrossEntropyLoss')).use_sigmoid_cls
    m.load_state_dict(synth_state_dict(m, seed=5))
    m = m.to(dev)
    gq = torch.Generator().manual_seed(91)
    feats = [torch.randn(2, 32, s, s + 4, generator=gq) for s in (16, 8)]
    cls, reg = m([f.to(dev).contiguous(memory_format=torch.channels_last) for f in feats])
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    for f, c, r in zip(feats, cls, reg):
        t = F.relu(F.conv2d(f, sd['rpn_conv.weight'], sd['rpn_conv.bias'], padding=1))
        want_c = F.conv2d(t, sd['rpn_cls.weight'], sd['rpn_cls.bias'])
        assert tuple(c.shape) == tuple(want_c.shape) == (2, 2 * A, f.shape[2], f.shape[3])
        assert _err(c, want_c) < 1e-4 and _err(r, F.conv2d(t, sd['rpn_reg.weight'], sd['rpn_reg.bias'])) < 1e-4
    heads, _ = m._heads([f.to(dev).contiguous(memory_format=torch.channels_last) for f in feats])
    for h, c in zip(heads, cls):
        obj = h.view(2, c.shape[2], c.shape[3], m.LD)[..., :A].permute(0, 3, 1, 2)
        assert _err(obj, c[:, 0::2].cpu() - c[:, 1::2].cpu()) < 1e-4       # the folded column is fg - bg
    # ---- selection against the real class
    g = torch.load(os.path.join(HERE, 'golden', 'reference_vectors_heads.pt'), weights_only=False)
    base = torch.stack(AnchorGenerator(strides=ag['strides'], ratios=ag['ratios'], scales=ag['scales']).base_anchors, 0)
    for c in g['rpn_predict_single_softmax']:
        LD, hs = 32, []
        for cl, rg in zip(c['cls'], c['reg']):
            _, H, W = cl.shape
            h = torch.zeros((H, W, LD))
            h[..., :A] = (cl[0::2] - cl[1::2]).permute(1, 2, 0)
            h[..., A:5 * A] = rg.permute(1, 2, 0)
            hs.append(h.reshape(H * W, LD).to(dev).contiguous())
        sel = ops.RpnSelector(base, ag['strides'], c['nms_pre'], c['max_per_img'], c['iou_thr'], c['min_bbox_size'],
                              DeltaXYWHBBoxCoder(), dev)
        out = sel(hs, c['sizes'], LD, torch.tensor([c['img_shape']], dtype=torch.float32, device=dev))
        k = int(out['count'][0])
        assert k == c['scores'].shape[0]
        gb, gs = out['boxes'][0, :k].cpu(), out['scores'][0, :k].cpu()
        assert _err(gs, c['scores']) < 1e-6
        bad = ((gb - c['bboxes']).abs().amax(1) > 2e-3).nonzero()[:, 0].tolist()
        assert len(bad) <= 4                                   # rows may trade places at (near-)equal scores only
        for i in bad:
            d = (c['bboxes'] - gb[i]).abs().amax(1)
            j = int(d.argmin())
            assert float(d[j]) < 2e-3 and abs(float(c['scores'][j]) - float(gs[i])) < 1e-6


def test_bbox_post_many_classes(dev):
    """multiclass_nms of a many-class head (bbox_nms.py:12-105; 500 RoIs x 80 classes = 40000 (RoI, class) pairs, above the
    16384 candidates the NMS sorts in LDS): ops.bbox_post sizes the NMS by the pairs that pass score_thr and the in-memory
    sort / 32-word reduction take over above 16384 of those.  Checked against the oracle's restatement of
    BBoxHead._predict_by_feat_single (itself pinned on the real class, test_oracle_golden.py)."""
    from oracle import glue
    from rsprompter_amd import ops
    from rsprompter_amd.anchor_heads import DeltaXYWHBBoxCoder
    coder = DeltaXYWHBBoxCoder(target_stds=(0.1, 0.1, 0.2, 0.2))
    for seed, (n, nc, thr, temp) in enumerate(((500, 80, 0.05, 3.0), (500, 80, 0.005, 0.7))):
        g = torch.Generator().manual_seed(500 + seed)
        xy = torch.rand(n, 2, generator=g) * 800
        roi = torch.cat([torch.zeros(n, 1), xy, xy + torch.rand(n, 2, generator=g) * 200 + 2], 1)
        cls_score = torch.randn(n, nc + 1, generator=g) * temp
        bbox_pred = torch.randn(n, nc * 4, generator=g)
        LD = (5 * nc + 1 + 3) // 4 * 4
        head = torch.zeros((n, LD))
        head[:, :nc + 1] = cls_score
        head[:, nc + 1:5 * nc + 1] = bbox_pred
        dets, labels, cand = glue.bbox_head_predict_single(roi, cls_score, bbox_pred, (1024, 1024), nc, thr, 0.5, 100)
        n_valid = int((torch.softmax(cls_score, -1)[:, :-1] > thr).sum())
        out = ops.bbox_post(head.to(dev), LD, roi.to(dev), torch.tensor([0, n]),
                            torch.tensor([[1024., 1024.]], device=dev), nc, thr, coder, 0.5, 100)
        k = int(out['count'][0])
        print(f'{n} RoIs x {nc} classes: {n_valid} candidates pass score_thr={thr}, {k} detections')
        assert k == labels.shape[0]
        pairs = match_detections(out['boxes'][0, :k].cpu(), out['scores'][0, :k].cpu(), out['ids'][0, :k].cpu().long(),
                                 dets[:, :4], dets[:, 4], labels)
        assert len(pairs) == k
        ii = torch.tensor([i for i, _ in pairs]); jj = torch.tensor([j for _, j in pairs])
        assert _err(out['boxes'][0, :k].cpu()[ii], dets[:, :4][jj]) < 2e-3
        assert int((ii != jj).sum()) <= 4
    assert n_valid > 16384          # the second case runs the in-memory sort


def test_bbox_post_matches_real_bbox_head_with_and_without_rescale(dev):
    """rsp_bbox_post (+ rsp_scale_boxes, rsp_batched_nms) on the golden vectors of the REAL BBoxHead._predict_by_feat_single
    (bbox_head.py:476-571; tests/golden/make_golden_heads.py): rescale=True multiplies by fp32(1 / scale_factor) before the
    NMS -- labels and the kept set are exact, boxes and scores to fp32 rounding of the decode."""
    from rsprompter_amd import ops
    from rsprompter_amd.anchor_heads import DeltaXYWHBBoxCoder
    g = torch.load(os.path.join(HERE, 'golden', 'reference_vectors_heads.pt'), weights_only=False)
    for key in ('bbox_head_predict_single', 'bbox_head_predict_single_rescale'):
        for c in g[key]:
            n, nc = c['roi'].shape[0], c['num_classes']
            LD = (5 * nc + 1 + 3) // 4 * 4
            head = torch.zeros((n, LD))
            head[:, :nc + 1] = c['cls_score']
            head[:, nc + 1:5 * nc + 1] = c['bbox_pred']
            sf = c.get('scale_factor')
            out = ops.bbox_post(head.to(dev), LD, c['roi'].to(dev), torch.tensor([0, n]), torch.tensor([c['img_shape']],
                                dtype=torch.float32, device=dev), nc, c['score_thr'],
                                DeltaXYWHBBoxCoder(target_stds=(0.1, 0.1, 0.2, 0.2)), c['iou_thr'], c['max_per_img'],
                                scale_factors=None if sf is None else [sf])
            k = int(out['count'][0])
            assert k == c['labels'].shape[0], (key, sf)
            pairs = match_detections(out['boxes'][0, :k].cpu(), out['scores'][0, :k].cpu(), out['ids'][0, :k].cpu().long(),
                                     c['bboxes'], c['scores'], c['labels'])
            assert len(pairs) == k
            ii = torch.tensor([i for i, _ in pairs]); jj = torch.tensor([j for _, j in pairs])
            assert _err(out['boxes'][0, :k].cpu()[ii], c['bboxes'][jj]) < 2e-3 and _err(out['scores'][0, :k].cpu()[ii], c['scores'][jj]) < 1e-6
            assert int((ii != jj).sum()) <= 4                      # rank swaps only at score ties


def test_resnet50_fpn_modules_match_real_classes(dev):
    """the HIP ResNet-50 + FPN modules on the inputs / weights of the golden run of the REAL mmdet classes."""
    from rsprompter_amd.samdet import FPN, ResNet
    from rsprompter_amd.synth import synth_state_dict
    g = torch.load(os.path.join(HERE, 'golden', 'reference_vectors_samdet.pt'), weights_only=False)['resnet_fpn']
    net = ResNet(depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                 norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, style='pytorch',
                 init_cfg=dict(type='Pretrained', checkpoint='torchvision://resnet50'))
    neck = FPN(in_channels=[256, 512, 1024, 2048], out_channels=256, num_outs=5)
    assert sorted((k, tuple(v.shape)) for k, v in net.state_dict().items()) == sorted(g['backbone_keys'])
    assert sorted((k, tuple(v.shape)) for k, v in neck.state_dict().items()) == sorted(g['neck_keys'])
    net.load_state_dict(synth_state_dict(net, g['seed'][0]), strict=True)
    neck.load_state_dict(synth_state_dict(neck, g['seed'][1]), strict=True)
    net, neck = net.to(dev), neck.to(dev)
    seed, shape = g['x']
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(seed))
    c = net(x.to(dev))
    p = neck(c)
    cs, fs = g['strides']
    assert [tuple(t.shape) for t in c] == g['c_shapes'] and [tuple(t.shape) for t in p] == g['p_shapes']
    for i, (got, want) in enumerate(zip(c, g['c'])):
        e, s = _err(got[:, ::cs], want), float(want.abs().max())
        print(f'C{i + 2}: err {e:.2e} (range {s:.1f})')
        assert e < 2e-4 * max(1.0, s)
    for i, (got, want) in enumerate(zip(p, g['p'])):
        e, s = _err(got[:, ::fs], want), float(want.abs().max())
        print(f'P{i + 2}: err {e:.2e} (range {s:.1f})')
        assert e < 2e-4 * max(1.0, s)


def test_samdet_end_to_end(dev):
    import rsprompter_amd as ra
    from oracle import glue
    from oracle.samdet import SAMDetOracle
    from rsprompter_amd.default_configs import samdet
    from rsprompter_amd.structures import DetDataSample, InstanceData
    from rsprompter_amd.synth import synth_images, synth_metas, synth_state_dict
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = ra.build_model(samdet('base', 10))
    oracle = SAMDetOracle('base', 10)
    oracle.load_state_dict(synth_state_dict(oracle, 0))
    sd = oracle.state_dict()
    res = model.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    model = model.to(dev)
    imgs = synth_images(2)
    metas = synth_metas(2, ori_shape=(400, 400), scale_factor=(2.56, 2.56))
    x = glue.data_preprocess(imgs, MEAN, STD, True, 32)
    ref, tr = oracle.predict(x, metas)
    # stage checks on the oracle's intermediates, then the free-running pipeline
    fp = model.detector.extract_feat(x.to(dev))
    for i, (a, b) in enumerate(zip(fp, tr['fpn'])):
        e, s = _err(a, b), float(b.abs().max())
        print(f'FPN level {i}: err {e:.2e} (range {s:.1f})')
        assert e < 2e-4 * max(1.0, s)
    out = model.test_step(dict(inputs=[i.to(dev) for i in imgs],
                               data_samples=[DetDataSample(metainfo=dict(m)) for m in metas]))
    total = 0
    for b in range(2):
        pi, r = out[b].pred_instances, ref[b]
        assert tuple(pi.masks.shape[1:]) == (400, 400) and pi.masks.dtype == torch.bool
        pairs = match_detections(pi.bboxes, pi.scores, pi.labels, r['bboxes'], r['scores'], r['labels'])
        assert len(pairs) >= r['labels'].shape[0] - 2 and abs(pi.labels.shape[0] - r['labels'].shape[0]) <= 2
        ii = torch.tensor([i for i, _ in pairs]); jj = torch.tensor([j for _, j in pairs])
        total += len(pairs)
        if len(pairs):
            eb = _err(pi.bboxes[ii], r['bboxes'][jj])
            mism = float((pi.masks.cpu()[ii] != r['masks'][jj]).float().mean())
            print(f'SAMDet img {b}: {pi.labels.shape[0]} dets, {len(pairs)} matched, box err {eb:.2e}, mask mismatch {mism:.2e}')
            assert eb < 5e-2 and mism < 1e-3
    assert total > 0
    # ground-truth boxes as prompts (`oracle_on`, models.py:1090-1153)
    model.test_cfg = dict(oracle_on=True)
    gt = [torch.tensor([[20.0, 30.0, 200.0, 260.0], [300.0, 100.0, 390.0, 380.0], [0.0, 0.0, 399.0, 399.0]]),
          torch.zeros((0, 4))]
    samples = []
    for m, gb in zip(metas, gt):
        s = DetDataSample(metainfo=dict(m))
        s.gt_instances = InstanceData()
        s.gt_instances.bboxes, s.gt_instances.labels = gb, torch.arange(gb.shape[0])
        samples.append(s)
    out = model.test_step(dict(inputs=[i.to(dev) for i in imgs], data_samples=samples))
    ref, tr = oracle.predict(x, metas, gt_boxes=gt)
    low = model._last_seg['low_res']
    e = _err(low, tr['seg'][0]['low_res'])
    print(f'SAM low-res mask logits err {e:.2e} (range {float(tr["seg"][0]["low_res"].abs().max()):.1f})')
    assert e < 2e-3
    assert float((out[0].pred_instances.masks.cpu() != ref[0]['masks']).float().mean()) < 1e-3
    assert tuple(out[1].pred_instances.masks.shape) == (0, 400, 400)
