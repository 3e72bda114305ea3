"""Golden vectors for the detection glue of the anchor path, produced by executing the REAL reference sources
(stub modules of make_golden.py) with ONE injected dependency: mmcv's `batched_nms` (not under /root/reference) is
replaced by the oracle's restatement of it (oracle/glue.py::batched_nms -> oracle/mmcv_ops.c nms).  What gets pinned is
everything AROUND the NMS call -- per-level top-k, delta decode, min-size filter, level-wise NMS bookkeeping, score
threshold, flat (roi, class) indexing, max_per_img -- i.e. the code that decides which indices come out:
  RPNHead._predict_by_feat_single / _bbox_post_process   mmdet/models/dense_heads/rpn_head.py:134-304
  multiclass_nms                                         mmdet/models/layers/bbox_nms.py:12-105
  BBoxHead._predict_by_feat_single                       mmdet/models/roi_heads/bbox_heads/bbox_head.py:476-571
  SingleRoIExtractor.map_roi_levels                      mmdet/models/roi_heads/roi_extractors/single_level_roi_extractor.py:44-63
Run in the build container:  python tests/golden/make_golden_heads.py  ->  tests/golden/reference_vectors_heads.pt
Inputs are continuous random numbers (no exact score ties), so the reference's unstable sort is deterministic here."""
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import make_golden as mg  # noqa: E402

OUT = os.path.join(HERE, 'reference_vectors_heads.pt')


class Cfg(dict):
    """ConfigDict stand-in: attribute access + .get()."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def main():
    mg._install_stubs()
    from oracle import build as oracle_build
    oracle_build.build()
    from oracle import glue
    from rsprompter_amd.structures import InstanceData

    def batched_nms(boxes, scores, idxs, nms_cfg, class_agnostic=False):
        return glue.batched_nms(boxes, scores, idxs, nms_cfg['iou_threshold'])

    out = {}
    # ------------------------------------------------------------------ RPN
    coder_mod = mg._load('mmdet/models/task_modules/coders/delta_xywh_bbox_coder.py', '_ref_coder')
    coder = coder_mod.DeltaXYWHBBoxCoder(target_means=[.0, .0, .0, .0], target_stds=[1.0, 1.0, 1.0, 1.0])
    coder.use_box_type = False
    coder.encode_size = 4              # BaseBBoxCoder.encode_size (base_bbox_coder.py), the stubbed base class
    ag = mg._load('mmdet/models/task_modules/prior_generators/anchor_generator.py', '_ref_anchor')
    rpn = mg._load('mmdet/models/dense_heads/rpn_head.py', '_ref_rpn')
    rpn.batched_nms = batched_nms
    rpn.InstanceData = InstanceData
    rpn.cat_boxes = torch.cat
    rpn.get_box_tensor = lambda b: b
    rpn.get_box_wh = lambda b: (b[:, 2] - b[:, 0], b[:, 3] - b[:, 1])
    rpn.empty_box_as = lambda b: b.new_zeros((0, 4))
    fake = types.SimpleNamespace(bbox_coder=coder, cls_out_channels=1, use_sigmoid_cls=True, test_cfg=None)
    fake._bbox_post_process = lambda **kw: rpn.RPNHead._bbox_post_process(fake, **kw)
    gen = ag.AnchorGenerator(strides=[4, 8, 16, 32, 64], ratios=[0.5, 1.0, 2.0], scales=[4, 8])
    sizes = [(32, 40), (16, 20), (8, 10), (4, 5), (2, 3)]
    priors = gen.grid_priors(sizes, device='cpu')
    cases = []
    for seed, (nms_pre, max_per_img, min_size) in enumerate(((300, 200, 0), (1000, 1000, 0), (50, 20, 4))):
        g = torch.Generator().manual_seed(60 + seed)
        cls = [torch.randn(6, h, w, generator=g) * 2 for h, w in sizes]
        reg = [torch.randn(24, h, w, generator=g) * 0.5 for h, w in sizes]
        cfg = Cfg(nms_pre=nms_pre, max_per_img=max_per_img, nms=Cfg(type='nms', iou_threshold=0.7), min_bbox_size=min_size)
        r = rpn.RPNHead._predict_by_feat_single(fake, cls, reg, None, priors, dict(img_shape=(128, 160)), cfg, rescale=False)
        cases.append(dict(cls=cls, reg=reg, sizes=sizes, img_shape=(128, 160), nms_pre=nms_pre, max_per_img=max_per_img,
                          min_bbox_size=min_size, iou_thr=0.7, bboxes=r.bboxes, scores=r.scores, labels=r.labels))
    out['rpn_predict_single'] = cases
    # softmax objectness (anchor_head.py:73-77 use_sigmoid=False -> [fg, bg] per anchor; rpn_head.py:193-200)
    fake2 = types.SimpleNamespace(bbox_coder=coder, cls_out_channels=2, use_sigmoid_cls=False, test_cfg=None)
    fake2._bbox_post_process = lambda **kw: rpn.RPNHead._bbox_post_process(fake2, **kw)
    cases = []
    for seed, (nms_pre, max_per_img, min_size) in enumerate(((300, 200, 0), (80, 40, 2))):
        g = torch.Generator().manual_seed(160 + seed)
        cls = [torch.randn(12, h, w, generator=g) * 2 for h, w in sizes]
        reg = [torch.randn(24, h, w, generator=g) * 0.5 for h, w in sizes]
        cfg = Cfg(nms_pre=nms_pre, max_per_img=max_per_img, nms=Cfg(type='nms', iou_threshold=0.7), min_bbox_size=min_size)
        r = rpn.RPNHead._predict_by_feat_single(fake2, cls, reg, None, priors, dict(img_shape=(128, 160)), cfg, rescale=False)
        cases.append(dict(cls=cls, reg=reg, sizes=sizes, img_shape=(128, 160), nms_pre=nms_pre, max_per_img=max_per_img,
                          min_bbox_size=min_size, iou_thr=0.7, bboxes=r.bboxes, scores=r.scores, labels=r.labels))
    out['rpn_predict_single_softmax'] = cases

    # ------------------------------------------------------------------ multiclass_nms
    nmsm = mg._load('mmdet/models/layers/bbox_nms.py', '_ref_bbox_nms')
    nmsm.batched_nms = batched_nms
    mc = []
    for seed, (n, nc, thr, max_num) in enumerate(((200, 10, 0.05, 100), (50, 3, 0.3, 5), (30, 2, 0.99, 100))):
        g = torch.Generator().manual_seed(80 + seed)
        xy = torch.rand(n, nc, 2, generator=g) * 400
        boxes = torch.cat([xy, xy + torch.rand(n, nc, 2, generator=g) * 200 + 1], -1).reshape(n, nc * 4)
        scores = torch.softmax(torch.randn(n, nc + 1, generator=g) * 2, -1)
        dets, labels, inds = nmsm.multiclass_nms(boxes, scores, thr, Cfg(type='nms', iou_threshold=0.5), max_num,
                                                 return_inds=True)
        mc.append(dict(boxes=boxes, scores=scores, score_thr=thr, iou_thr=0.5, max_num=max_num, dets=dets, labels=labels,
                       inds=inds))
    out['multiclass_nms'] = mc

    # ------------------------------------------------------------------ BBoxHead._predict_by_feat_single
    bh = mg._load('mmdet/models/roi_heads/bbox_heads/bbox_head.py', '_ref_bbox_head')
    bh.multiclass_nms = nmsm.multiclass_nms
    bh.InstanceData = InstanceData
    bh.get_box_tensor = lambda b: b
    coder2 = coder_mod.DeltaXYWHBBoxCoder(target_means=[0., 0., 0., 0.], target_stds=[0.1, 0.1, 0.2, 0.2])
    coder2.use_box_type = False
    coder2.encode_size = 4
    bb = []
    for seed, (n, nc) in enumerate(((300, 10), (40, 1), (1000, 10))):
        g = torch.Generator().manual_seed(120 + seed)
        fake_b = types.SimpleNamespace(bbox_coder=coder2, custom_cls_channels=False, reg_class_agnostic=False,
                                       num_classes=nc, predict_box_type='hbox')
        xy = torch.rand(n, 2, generator=g) * 800
        roi = torch.cat([torch.zeros(n, 1), xy, xy + torch.rand(n, 2, generator=g) * 200 + 2], 1)
        cls_score = torch.randn(n, nc + 1, generator=g) * 3
        bbox_pred = torch.randn(n, nc * 4, generator=g)
        cfg = Cfg(score_thr=0.05, nms=Cfg(type='nms', iou_threshold=0.5), max_per_img=100)
        r = bh.BBoxHead._predict_by_feat_single(fake_b, roi, cls_score, bbox_pred, dict(img_shape=(1024, 1024)),
                                                rescale=False, rcnn_test_cfg=cfg)
        bb.append(dict(roi=roi, cls_score=cls_score, bbox_pred=bbox_pred, img_shape=(1024, 1024), num_classes=nc,
                       score_thr=0.05, iou_thr=0.5, max_per_img=100, bboxes=r.bboxes, scores=r.scores, labels=r.labels))
    out['bbox_head_predict_single'] = bb
    # ... with rescale=True (a RoI head without mask branch, e.g. SAMDet's Faster R-CNN detector: bbox_head.py:549-552
    # multiplies the decoded, clipped boxes by fp32(1 / scale_factor) through the REAL scale_boxes BEFORE the NMS)
    import sys as _sys
    _sys.modules['mmdet.structures.bbox'].BaseBoxes = type('BaseBoxes', (), {})
    tr = mg._load('mmdet/structures/bbox/transforms.py', '_ref_bbox_transforms')
    bh.scale_boxes = tr.scale_boxes
    bbr = []
    for seed, (n, nc, sf) in enumerate(((300, 10, (2.56, 2.56)), (200, 1, (1.7, 2.3)), (500, 10, (0.8, 0.8)))):
        g = torch.Generator().manual_seed(140 + seed)
        fake_b = types.SimpleNamespace(bbox_coder=coder2, custom_cls_channels=False, reg_class_agnostic=False,
                                       num_classes=nc, predict_box_type='hbox')
        xy = torch.rand(n, 2, generator=g) * 800
        roi = torch.cat([torch.zeros(n, 1), xy, xy + torch.rand(n, 2, generator=g) * 200 + 2], 1)
        cls_score = torch.randn(n, nc + 1, generator=g) * 3
        bbox_pred = torch.randn(n, nc * 4, generator=g)
        cfg = Cfg(score_thr=0.05, nms=Cfg(type='nms', iou_threshold=0.5), max_per_img=100)
        r = bh.BBoxHead._predict_by_feat_single(fake_b, roi, cls_score, bbox_pred,
                                                dict(img_shape=(1024, 1024), scale_factor=sf), rescale=True, rcnn_test_cfg=cfg)
        bbr.append(dict(roi=roi, cls_score=cls_score, bbox_pred=bbox_pred, img_shape=(1024, 1024), num_classes=nc,
                        score_thr=0.05, iou_thr=0.5, max_per_img=100, scale_factor=sf, bboxes=r.bboxes, scores=r.scores,
                        labels=r.labels))
    out['bbox_head_predict_single_rescale'] = bbr

    # ------------------------------------------------------------------ RoI level mapping
    ext = mg._load('mmdet/models/roi_heads/roi_extractors/single_level_roi_extractor.py', '_ref_roi_ext')
    g = torch.Generator().manual_seed(99)
    wh = torch.cat([torch.tensor([[1., 1.], [111.9, 112.], [112., 112.], [112.01, 112.], [224., 224.], [447.9, 448.1],
                                  [448., 448.], [2000., 1500.]]), torch.rand(200, 2, generator=g) * 900 + 1])
    xy = torch.rand(wh.shape[0], 2, generator=g) * 100
    rois = torch.cat([torch.zeros(wh.shape[0], 1), xy, xy + wh], 1)
    lv = ext.SingleRoIExtractor.map_roi_levels(types.SimpleNamespace(finest_scale=56), rois, 4)
    out['map_roi_levels'] = dict(rois=rois, num_levels=4, finest_scale=56, out=lv)
    torch.save(out, OUT)
    print('wrote', OUT, {k: (len(v) if isinstance(v, list) else sorted(v.keys())) for k, v in out.items()})


if __name__ == '__main__':
    main()
