"""Golden vectors for the DeltaXYWHBBoxCoder branches no RSPrompter config uses (target_means != 0, clip_border=False,
add_ctr_clamp=True), produced by executing the REAL reference sources with the stub modules of make_golden.py:
  delta2bbox / DeltaXYWHBBoxCoder.decode          mmdet/models/task_modules/coders/delta_xywh_bbox_coder.py:71-131,264-361
  RPNHead._predict_by_feat_single                 mmdet/models/dense_heads/rpn_head.py:134-304     (coder of the RPN)
  BBoxHead._predict_by_feat_single                mmdet/models/roi_heads/bbox_heads/bbox_head.py:476-571 (coder of the head)
mmcv's `batched_nms` is the oracle's restatement, as in make_golden_heads.py.  Includes the reference's own known-answer
test of the centre clamp (tests/test_models/test_task_modules/test_coder/test_delta_xywh_bbox_coder.py:44-57).
Run in the build container:  python tests/golden/make_golden_coder.py  ->  tests/golden/reference_vectors_coder.pt"""
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import make_golden as mg  # noqa: E402
from make_golden_heads import Cfg  # noqa: E402

OUT = os.path.join(HERE, 'reference_vectors_coder.pt')

# keyword sets of the coder; 'plain' repeats the shipped configuration as a control
CODERS = dict(
    plain=dict(target_means=(0., 0., 0., 0.), target_stds=(0.1, 0.1, 0.2, 0.2)),
    means=dict(target_means=(0.1, -0.05, 0.2, -0.1), target_stds=(0.5, 0.4, 0.9, 1.1)),
    noclip=dict(target_means=(0., 0., 0., 0.), target_stds=(0.1, 0.1, 0.2, 0.2), clip_border=False),
    ctr=dict(target_means=(0., 0., 0., 0.), target_stds=(1., 1., 1., 1.), add_ctr_clamp=True, ctr_clamp=32),
    ctr_means_noclip=dict(target_means=(0.05, 0.05, -0.1, 0.1), target_stds=(0.6, 0.6, 1.2, 1.2), clip_border=False,
                          add_ctr_clamp=True, ctr_clamp=9),
)


def main():
    mg._install_stubs()
    from oracle import build as oracle_build
    oracle_build.build()
    from oracle import glue
    from rsprompter_amd.structures import InstanceData

    def batched_nms(boxes, scores, idxs, nms_cfg, class_agnostic=False):
        return glue.batched_nms(boxes, scores, idxs, nms_cfg['iou_threshold'])

    cm = mg._load('mmdet/models/task_modules/coders/delta_xywh_bbox_coder.py', '_ref_coder')
    cm.get_box_tensor = lambda b: b

    def real_coder(kw):
        c = cm.DeltaXYWHBBoxCoder(**kw)
        c.use_box_type = False
        c.encode_size = 4              # BaseBBoxCoder.encode_size (the stubbed base class)
        return c

    out = {'coders': CODERS}
    # ---- the reference's own known-answer test of add_ctr_clamp
    rois = torch.Tensor([[0., 0., 6., 6.], [0., 0., 1., 1.], [0., 0., 1., 1.], [5., 5., 5., 5.]])
    deltas = torch.Tensor([[1., 1., 2., 2.], [1., 1., 1., 1.], [0., 0., 2., -1.], [0.7, -1.9, -0.5, 0.3]])
    kw = dict(add_ctr_clamp=True, ctr_clamp=2)
    out['kat_ctr_clamp'] = dict(rois=rois, deltas=deltas, coder=kw, max_shape=(32, 32),
                                expected=torch.Tensor([[0.0000, 0.0000, 27.1672, 27.1672], [0.1409, 0.1409, 2.8591, 2.8591],
                                                       [0.0000, 0.3161, 4.1945, 0.6839], [5.0000, 5.0000, 5.0000, 5.0000]]),
                                out=real_coder(kw).decode(rois, deltas, max_shape=(32, 32)))
    # ---- decode of random boxes under every coder
    g = torch.Generator().manual_seed(7)
    xy = torch.rand(300, 2, generator=g) * 900
    r2 = torch.cat([xy, xy + torch.rand(300, 2, generator=g) * 300 + 1], 1)
    d2 = torch.randn(300, 12, generator=g) * 2
    out['decode'] = {name: dict(rois=r2, deltas=d2, max_shape=(1024, 1000),
                                out=real_coder(kw).decode(r2, d2, max_shape=(1024, 1000))) for name, kw in CODERS.items()}

    # ---- RPN with each coder
    ag = mg._load('mmdet/models/task_modules/prior_generators/anchor_generator.py', '_ref_anchor')
    rpn = mg._load('mmdet/models/dense_heads/rpn_head.py', '_ref_rpn')
    rpn.batched_nms = batched_nms
    rpn.InstanceData = InstanceData
    rpn.cat_boxes = torch.cat
    rpn.get_box_tensor = lambda b: b
    rpn.get_box_wh = lambda b: (b[:, 2] - b[:, 0], b[:, 3] - b[:, 1])
    rpn.empty_box_as = lambda b: b.new_zeros((0, 4))
    gen = ag.AnchorGenerator(strides=[4, 8, 16, 32, 64], ratios=[0.5, 1.0, 2.0], scales=[4, 8])
    sizes = [(32, 40), (16, 20), (8, 10), (4, 5), (2, 3)]
    priors = gen.grid_priors(sizes, device='cpu')
    cases = {}
    for seed, (name, kw) in enumerate(CODERS.items()):
        fake = types.SimpleNamespace(bbox_coder=real_coder(kw), cls_out_channels=1, use_sigmoid_cls=True, test_cfg=None)
        fake._bbox_post_process = (lambda f: lambda **k: rpn.RPNHead._bbox_post_process(f, **k))(fake)
        g = torch.Generator().manual_seed(300 + seed)
        cls = [torch.randn(6, h, w, generator=g) * 2 for h, w in sizes]
        reg = [torch.randn(24, h, w, generator=g) * 0.8 for h, w in sizes]
        cfg = Cfg(nms_pre=300, max_per_img=200, nms=Cfg(type='nms', iou_threshold=0.7), min_bbox_size=2)
        r = rpn.RPNHead._predict_by_feat_single(fake, cls, reg, None, priors, dict(img_shape=(128, 160)), cfg, rescale=False)
        cases[name] = dict(cls=cls, reg=reg, sizes=sizes, img_shape=(128, 160), nms_pre=300, max_per_img=200, min_bbox_size=2,
                           iou_thr=0.7, bboxes=r.bboxes, scores=r.scores, labels=r.labels)
    out['rpn_predict_single'] = cases

    # ---- R-CNN box head with each coder
    nmsm = mg._load('mmdet/models/layers/bbox_nms.py', '_ref_bbox_nms')
    nmsm.batched_nms = batched_nms
    bh = mg._load('mmdet/models/roi_heads/bbox_heads/bbox_head.py', '_ref_bbox_head')
    bh.multiclass_nms = nmsm.multiclass_nms
    bh.InstanceData = InstanceData
    bh.get_box_tensor = lambda b: b
    bb = {}
    for seed, (name, kw) in enumerate(CODERS.items()):
        n, nc = 250, 6
        g = torch.Generator().manual_seed(400 + seed)
        fake_b = types.SimpleNamespace(bbox_coder=real_coder(kw), custom_cls_channels=False, reg_class_agnostic=False,
                                       num_classes=nc, predict_box_type='hbox')
        xy = torch.rand(n, 2, generator=g) * 800
        roi = torch.cat([torch.zeros(n, 1), xy, xy + torch.rand(n, 2, generator=g) * 200 + 2], 1)
        cls_score = torch.randn(n, nc + 1, generator=g) * 3
        bbox_pred = torch.randn(n, nc * 4, generator=g)
        cfg = Cfg(score_thr=0.05, nms=Cfg(type='nms', iou_threshold=0.5), max_per_img=100)
        r = bh.BBoxHead._predict_by_feat_single(fake_b, roi, cls_score, bbox_pred, dict(img_shape=(1024, 1024)),
                                                rescale=False, rcnn_test_cfg=cfg)
        bb[name] = dict(roi=roi, cls_score=cls_score, bbox_pred=bbox_pred, img_shape=(1024, 1024), num_classes=nc,
                        score_thr=0.05, iou_thr=0.5, max_per_img=100, bboxes=r.bboxes, scores=r.scores, labels=r.labels)
    out['bbox_head_predict_single'] = bb
    torch.save(out, OUT)
    print('wrote', OUT, {k: (len(v) if hasattr(v, '__len__') else v) for k, v in out.items()})


if __name__ == '__main__':
    main()
