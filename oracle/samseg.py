"""CPU oracle of the SAMSeg sibling model `SAMSegMaskRCNN` (mmdet/rsprompter/models.py:1219-1244 over MaskRCNN /
TwoStageDetector.predict two_stage.py:147-195, StandardRoIHead.predict standard_roi_head.py:293-424, FCNMaskHead
fcn_mask_head.py:27-150 and its mask paste :276-480).  TEST INFRASTRUCTURE ONLY.

Backbone / neck / RPN / bbox head are the AnchorOracle's (same classes in the reference); the RPN has 3 anchors per
location (scales [8], configs/rsprompter/_base_/samseg-maskrcnn.py:87-90), the RoI features get NO extra positional
encoding, and the mask branch is the standard FCN head.  `paste_masks` / `fcn_predict_single` restate
`_do_paste_mask` / `FCNMaskHead._predict_by_feat_single` and are pinned on the real file by
tests/golden/make_golden_samseg.py."""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import glue, hf_sam
from .anchor import AnchorOracle, BBoxHead, FeatureAggregator, RPNHead, SimpleFPN, _Wrap


class _ConvRelu(nn.Module):          # mmcv ConvModule(conv + ReLU, no norm): key `conv.*`
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 3, padding=1)

    def forward(self, x):
        return F.relu(self.conv(x))


class FCNMaskHead(nn.Module):
    def __init__(self, num_convs=4, in_channels=256, conv_out_channels=256, num_classes=10, class_agnostic=False):
        super().__init__()
        self.class_agnostic = class_agnostic
        self.convs = nn.ModuleList([_ConvRelu(in_channels if i == 0 else conv_out_channels, conv_out_channels)
                                    for i in range(num_convs)])
        self.upsample = nn.ConvTranspose2d(conv_out_channels, conv_out_channels, 2, 2)
        self.conv_logits = nn.Conv2d(conv_out_channels, 1 if class_agnostic else num_classes, 1)

    def forward(self, x):
        for c in self.convs:
            x = c(x)
        return self.conv_logits(F.relu(self.upsample(x)))


def paste_masks(masks, boxes, img_h, img_w):
    """fcn_mask_head.py:423-480 `_do_paste_mask(skip_empty=False)`: masks [N,1,h,w] probabilities -> [N,img_h,img_w]."""
    x0, y0, x1, y1 = torch.split(boxes, 1, dim=1)
    N = masks.shape[0]
    img_y = torch.arange(0, img_h).to(torch.float32) + 0.5
    img_x = torch.arange(0, img_w).to(torch.float32) + 0.5
    img_y = (img_y - y0) / (y1 - y0) * 2 - 1
    img_x = (img_x - x0) / (x1 - x0) * 2 - 1
    img_x = torch.where(torch.isinf(img_x), torch.zeros_like(img_x), img_x)
    img_y = torch.where(torch.isinf(img_y), torch.zeros_like(img_y), img_y)
    gx = img_x[:, None, :].expand(N, img_y.size(1), img_x.size(1))
    gy = img_y[:, :, None].expand(N, img_y.size(1), img_x.size(1))
    grid = torch.stack([gx, gy], dim=3)
    return F.grid_sample(masks.to(torch.float32), grid, align_corners=False)[:, 0]


def fcn_predict_single(mask_preds, bboxes, labels, img_meta, mask_thr_binary=0.5, rescale=True, class_agnostic=False):
    """fcn_mask_head.py:276-420 (activate_map=False, threshold >= 0).  Returns (bool masks, bboxes (rescaled))."""
    scale_factor = bboxes.new_tensor(img_meta['scale_factor']).repeat((1, 2))
    img_h, img_w = img_meta['ori_shape'][:2]
    mask_preds = mask_preds.sigmoid()
    if rescale:
        bboxes = bboxes / scale_factor
    else:
        w_scale, h_scale = scale_factor[0, 0], scale_factor[0, 1]
        img_h = int(np.round(img_h * h_scale.item()).astype(np.int32))
        img_w = int(np.round(img_w * w_scale.item()).astype(np.int32))
    N = len(mask_preds)
    if not class_agnostic:
        mask_preds = mask_preds[range(N), labels][:, None]
    probs = paste_masks(mask_preds, bboxes, img_h, img_w)
    # CPU path of the reference (:390-404, `skip_empty=device.type == 'cpu'`, one instance per chunk): only the region
    # [floor(x0) - 1, ceil(x1) + 1) x [floor(y0) - 1, ceil(y1) + 1) (clamped to the image) is pasted, the rest stays 0.
    # Inside it the values equal the full paste; it matters only for degenerate boxes (zero width / height: the inf -> 0
    # rule would otherwise smear the mask centre over whole rows / columns, which is what the reference does on a GPU).
    xs, ys = torch.arange(img_w)[None, None, :], torch.arange(img_h)[None, :, None]
    x0i = torch.clamp(bboxes[:, 0].floor() - 1, min=0)[:, None, None]
    y0i = torch.clamp(bboxes[:, 1].floor() - 1, min=0)[:, None, None]
    x1i = torch.clamp(bboxes[:, 2].ceil() + 1, max=img_w)[:, None, None]
    y1i = torch.clamp(bboxes[:, 3].ceil() + 1, max=img_h)[:, None, None]
    region = (xs >= x0i) & (xs < x1i) & (ys >= y0i) & (ys < y1i)
    probs = torch.where(region, probs, torch.zeros_like(probs))
    return probs >= mask_thr_binary, bboxes, probs


class SAMSegMaskRCNNOracle(nn.Module):
    def __init__(self, arch='base', num_classes=10, test_cfg=None):
        super().__init__()
        depth = hf_sam.ARCH[arch]['num_hidden_layers']
        self.arch, self.num_classes = arch, num_classes
        self.backbone = _Wrap('vision_encoder', hf_sam.build_vision_encoder(arch))
        self.neck = nn.Module()
        self.neck.feature_aggregator = FeatureAggregator(arch, 32, 256, list(range(1, depth + 1, 2)))
        self.neck.feature_spliter = SimpleFPN()
        self.rpn_head = RPNHead(num_anchors=3)
        self.roi_head = nn.Module()
        self.roi_head.bbox_head = BBoxHead(num_classes=num_classes)
        self.roi_head.mask_head = FCNMaskHead(num_classes=num_classes)
        self.strides = [4, 8, 16, 32, 64]
        self.anchor_scales, self.anchor_ratios = [8], [0.5, 1.0, 2.0]
        self.test_cfg = test_cfg or dict(
            rpn=dict(nms_pre=1000, max_per_img=1000, iou_threshold=0.7, min_bbox_size=0),
            rcnn=dict(score_thr=0.05, iou_threshold=0.5, max_per_img=100, mask_thr_binary=0.5))
        self.eval()

    rpn_predict = AnchorOracle.rpn_predict
    bbox_predict = AnchorOracle.bbox_predict

    @torch.no_grad()
    def extract_feat(self, batch_inputs):
        _, hidden = hf_sam.run_vision_encoder(self.backbone.vision_encoder, batch_inputs)
        return self.neck.feature_spliter(self.neck.feature_aggregator(hidden))

    @torch.no_grad()
    def predict(self, batch_inputs, metas, rescale=True):
        x = self.extract_feat(batch_inputs)
        props, t1 = self.rpn_predict(x, metas)
        dets, t2 = self.bbox_predict(x, [p['bboxes'] for p in props], metas)
        bboxes = [d['bboxes'] for d in dets]
        rois = torch.cat([torch.cat([b.new_full((b.shape[0], 1), i), b], 1) for i, b in enumerate(bboxes)], 0)
        results, trace = [], dict(fpn=x, proposals=props, dets=dets, mask_rois=rois)
        trace.update(t1); trace.update(t2)
        if rois.shape[0] == 0:
            for d, meta in zip(dets, metas):
                h, w = meta['ori_shape'][:2]
                results.append(dict(d, masks=torch.zeros((0, h, w), dtype=torch.bool)))
            return results, trace
        feats = glue.roi_extract(x[:4], rois, 14, self.strides[:4])
        logits = self.roi_head.mask_head(feats)
        trace.update(mask_feats=feats, mask_logits=logits)
        start = 0
        for d, meta in zip(dets, metas):
            n = d['bboxes'].shape[0]
            if n == 0:
                h, w = meta['ori_shape'][:2]
                results.append(dict(d, masks=torch.zeros((0, h, w), dtype=torch.bool)))
                continue
            masks, bb, _ = fcn_predict_single(logits[start:start + n], d['bboxes'], d['labels'], meta,
                                              self.test_cfg['rcnn']['mask_thr_binary'], rescale)
            start += n
            results.append(dict(bboxes=bb, scores=d['scores'], labels=d['labels'], masks=masks))
        return results, trace


# ----------------------------------------------------------------------------------------------------------------------
# SAMSegMask2Former (models.py:1247-1274): SAM encoder -> RSFPN -> the STANDARD mmdet Mask2FormerHead / fusion head
# ----------------------------------------------------------------------------------------------------------------------
class Mask2FormerHead(nn.Module):
    """mmdet Mask2FormerHead (mask2former_head.py:62-156 ctor, :340-380 `_forward_head`, :382-460 `forward`) with the
    configuration of configs/rsprompter/_base_/samseg-mask2former.py:88-150: feat 256, 8 heads, 3-layer deformable
    encoder (FFN 1024), 9 decoder layers (FFN 2048), sine PE num_feats 128, single-Linear class head."""

    def __init__(self, num_classes, num_queries, feat=256, out=256, dec_layers=9, dec_ffn=2048, enc_ffn=1024):
        super().__init__()
        from .query import DecLayer, PixelDecoder
        self.num_classes, self.num_queries, self.num_heads, self.n_dec = num_classes, num_queries, 8, dec_layers
        self.pe_feats = 128
        self.pixel_decoder = PixelDecoder(feat, out, ffn=enc_ffn, enc_layers=3)
        self.pixel_decoder.pe_feats = 128
        self.transformer_decoder = nn.Module()
        self.transformer_decoder.layers = nn.ModuleList([DecLayer(feat, dec_ffn) for _ in range(dec_layers)])
        self.transformer_decoder.post_norm = nn.LayerNorm(feat)
        self.query_embed = nn.Embedding(num_queries, feat)
        self.query_feat = nn.Embedding(num_queries, feat)
        self.level_embed = nn.Embedding(3, feat)
        self.cls_embed = nn.Linear(feat, num_classes + 1)
        self.mask_embed = nn.Sequential(nn.Linear(feat, feat), nn.ReLU(inplace=True), nn.Linear(feat, feat),
                                        nn.ReLU(inplace=True), nn.Linear(feat, out))

    def _forward_head(self, decoder_out, mask_feature, attn_size):
        decoder_out = self.transformer_decoder.post_norm(decoder_out)
        cls_pred = self.cls_embed(decoder_out)
        mask_pred = torch.einsum('bqc,bchw->bqhw', self.mask_embed(decoder_out), mask_feature)
        attn_mask = F.interpolate(mask_pred, attn_size, mode='bilinear', align_corners=False)
        attn_mask = attn_mask.flatten(2).unsqueeze(1).repeat((1, self.num_heads, 1, 1)).flatten(0, 1)
        return cls_pred, mask_pred, attn_mask.sigmoid() < 0.5

    @torch.no_grad()
    def forward(self, x):
        bs = x[0].shape[0]
        mask_features, mem = self.pixel_decoder(x)
        dec_in, dec_pos = [], []
        for i in range(3):
            dec_in.append(mem[i].flatten(2).permute(0, 2, 1) + self.level_embed.weight[i].view(1, 1, -1))
            pe = glue.sine_positional_encoding(bs, mem[i].shape[-2], mem[i].shape[-1], num_feats=self.pe_feats)
            dec_pos.append(pe.flatten(2).permute(0, 2, 1))
        qf = self.query_feat.weight.unsqueeze(0).repeat((bs, 1, 1))
        qe = self.query_embed.weight.unsqueeze(0).repeat((bs, 1, 1))
        cls, mask, attn_mask = self._forward_head(qf, mask_features, mem[0].shape[-2:])
        trace = dict(mask_features=mask_features, memory=mem, attn_masks=[], query_feats=[qf], cls_pred_all=[cls],
                     mask_pred_all=[mask])
        for i in range(self.n_dec):
            lvl = i % 3
            attn_mask = attn_mask & (attn_mask.sum(-1) != attn_mask.shape[-1]).unsqueeze(-1)   # :432-434
            trace['attn_masks'].append(attn_mask)
            qf = self.transformer_decoder.layers[i](qf, dec_in[lvl], dec_in[lvl], qe, dec_pos[lvl], attn_mask)
            cls, mask, attn_mask = self._forward_head(qf, mask_features, mem[(i + 1) % 3].shape[-2:])
            trace['query_feats'].append(qf)
            trace['cls_pred_all'].append(cls)
            trace['mask_pred_all'].append(mask)
        trace.update(cls_pred=cls, mask_pred=mask)
        return cls, mask, trace


class SAMSegMask2FormerOracle(nn.Module):
    """SAMSegMask2Former.predict = MaskFormer.predict (maskformer.py:83-151): extract_feat (models.py:1262-1274, the
    neck only) -> MaskFormerHead.predict (maskformer_head.py:569-604: last stage, bilinear up to batch_input_shape) ->
    MaskFormerFusionHead.predict (maskformer_fusion_head.py:184-270, same body as the RS fusion head)."""

    def __init__(self, arch='base', num_classes=10, num_queries=70, select_layers=None, max_per_image=None):
        super().__init__()
        depth = hf_sam.ARCH[arch]['num_hidden_layers']
        select_layers = list(select_layers) if select_layers is not None else list(range(1, depth + 1, 2))
        self.num_classes = num_classes
        self.max_per_image = max_per_image if max_per_image is not None else num_queries
        self.backbone = _Wrap('vision_encoder', hf_sam.build_vision_encoder(arch))
        self.neck = nn.Module()
        self.neck.feature_aggregator = FeatureAggregator(arch, 32, 256, select_layers)
        self.neck.feature_spliter = SimpleFPN()
        self.panoptic_head = Mask2FormerHead(num_classes, num_queries)
        self.eval()

    @torch.no_grad()
    def extract_feat(self, batch_inputs):
        _, hidden = hf_sam.run_vision_encoder(self.backbone.vision_encoder, batch_inputs)
        return self.neck.feature_spliter(self.neck.feature_aggregator(hidden))

    @torch.no_grad()
    def predict(self, batch_inputs, metas, rescale=True):
        from .query import fusion_predict
        x = self.extract_feat(batch_inputs)
        cls, mask, trace = self.panoptic_head(x)
        img_shape = metas[0]['batch_input_shape']
        mask_up = F.interpolate(mask, size=(img_shape[0], img_shape[1]), mode='bilinear', align_corners=False)
        results = fusion_predict(cls, mask_up, metas, self.num_classes, self.max_per_image, rescale)
        trace.update(fpn=x)
        return results, trace
