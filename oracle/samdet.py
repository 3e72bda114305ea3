"""CPU oracle of `SAMDet` (mmdet/rsprompter/models.py:1061-1215, configs/rsprompter/_base_/samdet.py): a Faster R-CNN
R50-FPN detector whose boxes prompt a frozen HF `SamModel`.  TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

Restated from the reference (module tree = the reference's `state_dict` keys, so seeded weights load into both sides):
  ResNet / Bottleneck   mmdet/models/backbones/resnet.py:103-300 (style='pytorch': the stride sits on the 3x3 conv),
                        :371-672 (stem conv1 7x7 s2 + bn1 + relu + MaxPool2d(3, 2, 1), stage strides (1, 2, 2, 2))
  ResLayer              mmdet/models/layers/res_layer.py:12-107 (downsample = 1x1 conv stride s + BN on the first block)
  FPN                   mmdet/models/necks/fpn.py:15-222 (1x1 laterals with bias, nearest top-down, 3x3 output convs,
                        extra level = max_pool2d(k=1, stride=2))
  FasterRCNN.predict    two_stage.py:147-195 -> RPNHead.predict (3 anchors) -> StandardRoIHead.predict_bbox with
                        rescale=True (bbox_head.py:549-552: boxes * fp32(1 / scale_factor) before the NMS)
  SAMDet.predict        models.py:1155-1213 (test_cfg is None in every samdet-*.py, so the detector's boxes are used):
                        per image `segmentor(pixel_values, input_boxes=bboxes * scale_factor, multimask_output=False)`,
                        pred_masks[0].squeeze(1) -> bilinear to img_shape -> crop to int(ori * scale) -> bilinear to
                        ori_shape -> > 0
The SAM model itself is the HF dependency (transformers `SamModel`, pinned 4.38.1 in the reference's environment; the
installed version is used here), built offline from the arch name like oracle/hf_sam.py does.
`tests/golden/make_golden_samdet.py` pins ResNet / FPN on the real resnet.py / fpn.py."""
import torch
import torch.nn.functional as F
from torch import nn

from . import glue, hf_sam
from .anchor import AnchorOracle, BBoxHead, RPNHead


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=False):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
                                            nn.BatchNorm2d(planes * 4))

    def forward(self, x):
        identity = x
        out = F.relu(self.bn1(self.conv1(x)))
        out = F.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return F.relu(out + identity)


class ResNet(nn.Module):
    arch_settings = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}

    def __init__(self, depth=50, out_indices=(0, 1, 2, 3)):
        super().__init__()
        self.out_indices = tuple(out_indices)
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        inplanes = 64
        for i, nb in enumerate(self.arch_settings[depth]):
            planes, stride = 64 * 2 ** i, (1 if i == 0 else 2)
            blocks = [Bottleneck(inplanes, planes, stride, downsample=(stride != 1 or inplanes != planes * 4))]
            inplanes = planes * 4
            blocks += [Bottleneck(inplanes, planes) for _ in range(1, nb)]
            setattr(self, f'layer{i + 1}', nn.Sequential(*blocks))

    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
        outs = []
        for i in range(4):
            x = getattr(self, f'layer{i + 1}')(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)


class _Conv(nn.Module):               # mmcv ConvModule without norm / activation: key `conv.*`
    def __init__(self, cin, cout, k, padding=0):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, padding=padding)

    def forward(self, x):
        return self.conv(x)


class FPN(nn.Module):
    def __init__(self, in_channels=(256, 512, 1024, 2048), out_channels=256, num_outs=5):
        super().__init__()
        self.num_outs = num_outs
        self.lateral_convs = nn.ModuleList([_Conv(c, out_channels, 1) for c in in_channels])
        self.fpn_convs = nn.ModuleList([_Conv(out_channels, out_channels, 3, padding=1) for _ in in_channels])

    def forward(self, inputs):
        laterals = [l(x) for l, x in zip(self.lateral_convs, inputs)]
        for i in range(len(laterals) - 1, 0, -1):
            laterals[i - 1] = laterals[i - 1] + F.interpolate(laterals[i], size=laterals[i - 1].shape[2:], mode='nearest')
        outs = [c(l) for c, l in zip(self.fpn_convs, laterals)]
        for _ in range(self.num_outs - len(outs)):
            outs.append(F.max_pool2d(outs[-1], 1, stride=2))
        return tuple(outs)


class FasterRCNNOracle(nn.Module):
    def __init__(self, num_classes=1, depth=50, test_cfg=None):
        super().__init__()
        self.num_classes = num_classes
        self.backbone = ResNet(depth)
        self.neck = FPN()
        self.rpn_head = RPNHead(num_anchors=3)
        self.roi_head = nn.Module()
        self.roi_head.bbox_head = BBoxHead(num_classes=num_classes)
        self.strides = [4, 8, 16, 32, 64]
        self.anchor_scales, self.anchor_ratios = [8], [0.5, 1.0, 2.0]
        self.test_cfg = test_cfg or dict(
            rpn=dict(nms_pre=1000, max_per_img=1000, iou_threshold=0.7, min_bbox_size=0),
            rcnn=dict(score_thr=0.05, iou_threshold=0.5, max_per_img=100))
        self.eval()

    rpn_predict = AnchorOracle.rpn_predict

    @torch.no_grad()
    def extract_feat(self, batch_inputs):
        c = self.backbone(batch_inputs)
        return self.neck(c), c

    @torch.no_grad()
    def bbox_predict(self, x, proposals, metas, rescale=True):
        """standard_roi_head.py:293-363 + bbox_head.py:476-571."""
        rois = torch.cat([torch.cat([p.new_full((p.shape[0], 1), i), p], 1) for i, p in enumerate(proposals)], 0)
        feats = glue.roi_extract(x[:4], rois, 7, self.strides[:4])
        cls_score, bbox_pred = self.roi_head.bbox_head(feats)
        c = self.test_cfg['rcnn']
        out, start = [], 0
        for p, meta in zip(proposals, metas):
            n = p.shape[0]
            cs, bp, roi = cls_score[start:start + n], bbox_pred[start:start + n], rois[start:start + n]
            start += n
            if n == 0:
                out.append(dict(bboxes=p.new_zeros((0, 4)), scores=p.new_zeros(0), labels=torch.zeros(0, dtype=torch.long),
                                cand=torch.zeros(0, dtype=torch.long)))
                continue
            dets, labels, cand = glue.bbox_head_predict_single(
                roi, cs, bp, meta['img_shape'], self.num_classes, c['score_thr'], c['iou_threshold'], c['max_per_img'],
                scale_factor=meta['scale_factor'] if rescale else None)
            out.append(dict(bboxes=dets[:, :4], scores=dets[:, 4], labels=labels, cand=cand))
        return out, dict(rois=rois, roi_feats=feats, cls_score=cls_score, bbox_pred=bbox_pred)

    @torch.no_grad()
    def predict(self, batch_inputs, metas, rescale=True):
        x, c = self.extract_feat(batch_inputs)
        props, t1 = self.rpn_predict(x, metas)
        dets, t2 = self.bbox_predict(x, [p['bboxes'] for p in props], metas, rescale)
        trace = dict(backbone=c, fpn=x, proposals=props)
        trace.update(t1); trace.update(t2)
        return dets, trace


def build_sam_model(arch):
    """HF `SamModel(SamConfig)` of models.py:727-730 with the arch inferred from the name (no hub access here)."""
    from transformers.models.sam.configuration_sam import SamConfig
    cfg = SamConfig(vision_config=hf_sam.ARCH[arch])
    for c in (cfg, cfg.vision_config, cfg.mask_decoder_config, cfg.prompt_encoder_config):
        c._attn_implementation = 'eager'
    return hf_sam.hf.SamModel(cfg).eval()


def sam_box_masks(sam_model, input_img, bboxes, meta):
    """models.py:1174-1206 for one image: boxes in INPUT pixels -> bool masks [n, ori_h, ori_w] (+ intermediates)."""
    outputs = sam_model(pixel_values=input_img.unsqueeze(0), input_boxes=bboxes.unsqueeze(0), multimask_output=False)
    low = outputs.pred_masks[0].squeeze(1)
    ori_h, ori_w = meta['ori_shape'][:2]
    sf = meta['scale_factor']
    sh, sw = int(ori_h * sf[1]), int(ori_w * sf[0])
    m = F.interpolate(low[:, None], size=tuple(meta['img_shape'][:2]), mode='bilinear', align_corners=False)[:, 0]
    m = m[:, :sh, :sw]
    m = F.interpolate(m[:, None], size=(ori_h, ori_w), mode='bilinear', align_corners=False)[:, 0]
    return m > 0, dict(low_res=low, logits=m)


class SAMDetOracle(nn.Module):
    def __init__(self, arch='base', num_classes=1, depth=50, test_cfg=None):
        super().__init__()
        self.detector = FasterRCNNOracle(num_classes, depth, test_cfg)
        self.segmentor = nn.Module()
        self.segmentor.sam_model = build_sam_model(arch)
        self.eval()

    @torch.no_grad()
    def predict(self, batch_inputs, metas, rescale=True, gt_boxes=None):
        """gt_boxes (list of [n,4] in ORIGINAL-image pixels): `oracle_predict` (models.py:1090-1153, test_cfg.oracle_on)."""
        dets, trace = self.detector.predict(batch_inputs, metas, rescale)
        results, seg = [], []
        for i, (img, d, meta) in enumerate(zip(batch_inputs, dets, metas)):
            if gt_boxes is not None:
                g = gt_boxes[i]
                d = dict(bboxes=g, scores=torch.ones(g.shape[0]), labels=torch.zeros(g.shape[0], dtype=torch.long))
            bboxes = d['bboxes']
            h, w = meta['ori_shape'][:2]
            if bboxes.shape[0] == 0:
                results.append(dict(d, masks=torch.zeros((0, h, w), dtype=torch.bool)))
                seg.append(None)
                continue
            sf = bboxes.new_tensor(meta['scale_factor']).repeat((1, bboxes.size(-1) // 2))
            masks, t = sam_box_masks(self.segmentor.sam_model, img, bboxes * sf, meta)
            results.append(dict(bboxes=bboxes, scores=d['scores'], labels=d['labels'], masks=masks))
            seg.append(t)
        trace.update(dets=dets, seg=seg)
        return results, trace
