"""CPU oracle of `RSPrompterAnchor.predict` (reference mmdet/rsprompter/models.py:53-170 and
the classes it wires together).  TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

The module tree reproduces the reference's `state_dict` key layout so the
same seeded synthetic weights load into this oracle and into the HIP modules.
"""
import einops
import torch
import torch.nn.functional as F
from torch import nn

from . import glue, hf_sam


class LN2d(nn.Module):
    """models.py:33-50."""

    def __init__(self, c, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.eps = eps

    def forward(self, x):
        u = x.mean(1, keepdim=True)
        s = (x - u).pow(2).mean(1, keepdim=True)
        x = (x - u) / torch.sqrt(s + self.eps)
        return self.weight[:, None, None] * x + self.bias[:, None, None]


class ConvLN(nn.Module):
    """mmcv ConvModule(conv -> norm, act_cfg=None): bias=False because a norm follows
    (App. B); the norm sub-module of a non-torch norm class is named `norm_layer`."""

    def __init__(self, cin, cout, k, padding=0, eps=1e-5):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, padding=padding, bias=False)
        self.norm_layer = LN2d(cout, eps)

    def forward(self, x):
        return self.norm_layer(self.conv(x))


class FeatureAggregator(nn.Module):
    """models.py:988-1057."""
    in_channels_dict = {'base': [768] * 13, 'large': [1024] * 25, 'huge': [1280] * 33}

    def __init__(self, arch, hidden_channels=32, out_channels=256, select_layers=range(1, 13, 2)):
        super().__init__()
        self.in_channels = self.in_channels_dict[arch]
        self.select_layers = list(select_layers)
        h = hidden_channels
        self.downconvs = nn.ModuleList([
            nn.Sequential(nn.Conv2d(self.in_channels[i], h, 1), nn.BatchNorm2d(h), nn.ReLU(inplace=True),
                          nn.Conv2d(h, h, 3, padding=1), nn.BatchNorm2d(h), nn.ReLU(inplace=True))
            for i in self.select_layers])
        self.hidden_convs = nn.ModuleList([
            nn.Sequential(nn.Conv2d(h, h, 3, padding=1), nn.BatchNorm2d(h), nn.ReLU(inplace=True))
            for _ in self.select_layers])
        self.fusion_conv = nn.Sequential(
            nn.Conv2d(h, out_channels, 1), nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True),
            nn.Conv2d(out_channels, out_channels, 3, padding=1), nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True), nn.Conv2d(out_channels, out_channels, 3, padding=1))

    def forward(self, inputs):
        assert len(inputs) == len(self.in_channels)
        inputs = [einops.rearrange(x, 'b h w c -> b c h w') for x in inputs]
        feats = [self.downconvs[i](inputs[l]) for i, l in enumerate(self.select_layers)]
        x = None
        for hs, conv in zip(feats, self.hidden_convs):
            if x is not None:
                hs = x + hs
            x = hs + conv(hs)
        return self.fusion_conv(x)


class PseudoAggregator(nn.Module):
    """models.py:943-984."""

    def __init__(self, cin=256, hidden=512, cout=256):
        super().__init__()
        from .vitsam import _LN2d
        self.channel_fusion = nn.Sequential(
            nn.Conv2d(cin, hidden, 1, bias=False), _LN2d(hidden, eps=1e-6),
            nn.Conv2d(hidden, hidden, 3, padding=1, bias=False), _LN2d(hidden, eps=1e-6),
            nn.Conv2d(hidden, cout, 3, padding=1, bias=False), _LN2d(cout, eps=1e-6))

    def forward(self, inputs):
        assert len(inputs) == 1
        return self.channel_fusion(inputs[0])


class SimpleFPN(nn.Module):
    """models.py:1278-1363 with norm_cfg=LN2d, act_cfg=None, num_outs=5.  The LN2d layers are built through mmcv's
    `build_norm_layer`, which sets eps=1e-5 when the norm_cfg carries none (mmcv/cnn/bricks/norm.py), overriding
    LN2d's own default of 1e-6 -- pinned by tests/golden/make_golden_forwards.py, which runs the real class."""

    def __init__(self, backbone_channel=256, in_channels=(64, 128, 256, 256), out_channels=256, num_outs=5):
        super().__init__()
        c = backbone_channel
        self.num_ins, self.num_outs = len(in_channels), num_outs
        self.fpn1 = nn.Sequential(nn.ConvTranspose2d(c, c // 2, 2, 2), LN2d(c // 2, 1e-5), nn.GELU(),
                                  nn.ConvTranspose2d(c // 2, c // 4, 2, 2))
        self.fpn2 = nn.Sequential(nn.ConvTranspose2d(c, c // 2, 2, 2))
        self.fpn3 = nn.Sequential(nn.Identity())
        self.fpn4 = nn.Sequential(nn.MaxPool2d(kernel_size=2, stride=2))
        self.lateral_convs = nn.ModuleList([ConvLN(ci, out_channels, 1) for ci in in_channels])
        self.fpn_convs = nn.ModuleList([ConvLN(out_channels, out_channels, 3, padding=1) for _ in in_channels])

    def forward(self, x):
        ins = [self.fpn1(x), self.fpn2(x), self.fpn3(x), self.fpn4(x)]
        lats = [l(ins[i]) for i, l in enumerate(self.lateral_convs)]
        outs = [self.fpn_convs[i](lats[i]) for i in range(self.num_ins)]
        for _ in range(self.num_outs - self.num_ins):
            outs.append(F.max_pool2d(outs[-1], 1, stride=2))
        return tuple(outs)


class RPNHead(nn.Module):
    """rpn_head.py:45-97 (num_convs=1): 3x3 conv + ReLU, 1x1 cls (A*1), 1x1 reg (A*4)."""

    def __init__(self, in_channels=256, feat_channels=256, num_anchors=6):
        super().__init__()
        self.rpn_conv = nn.Conv2d(in_channels, feat_channels, 3, padding=1)
        self.rpn_cls = nn.Conv2d(feat_channels, num_anchors, 1)
        self.rpn_reg = nn.Conv2d(feat_channels, num_anchors * 4, 1)

    def forward(self, feats):
        cls, reg = [], []
        for x in feats:
            x = F.relu(self.rpn_conv(x))
            cls.append(self.rpn_cls(x))
            reg.append(self.rpn_reg(x))
        return cls, reg


class BBoxHead(nn.Module):
    """Shared2FCBBoxHead: convfc_bbox_head.py:163-233 (2 shared FCs + fc_cls/fc_reg)."""

    def __init__(self, in_channels=256, fc_out=1024, roi_feat_size=7, num_classes=10):
        super().__init__()
        self.num_classes = num_classes
        self.shared_fcs = nn.ModuleList([nn.Linear(in_channels * roi_feat_size ** 2, fc_out),
                                         nn.Linear(fc_out, fc_out)])
        self.fc_cls = nn.Linear(fc_out, num_classes + 1)
        self.fc_reg = nn.Linear(fc_out, 4 * num_classes)

    def forward(self, x):
        x = x.flatten(1)
        for fc in self.shared_fcs:
            x = F.relu(fc(x))
        return self.fc_cls(x), self.fc_reg(x)


class _Wrap(nn.Module):
    def __init__(self, name, mod):
        super().__init__()
        self.add_module(name, mod)


class MaskHead(nn.Module):
    """RSPrompterAnchorMaskHead: models.py:1597-1698."""

    def __init__(self, in_channels=256, roi_feat_size=14, per_pointset_point=5, with_sincos=True):
        super().__init__()
        self.per_pointset_point, self.with_sincos = per_pointset_point, with_sincos
        self.mask_decoder = _Wrap('mask_decoder', hf_sam.build_mask_decoder())
        self.no_mask_embed = nn.Embedding(1, 256)
        ns = 2 if with_sincos else 1
        c = in_channels
        self.point_emb = nn.Sequential(
            nn.Conv2d(c, c, 3, stride=2, padding=1), nn.BatchNorm2d(c), nn.ReLU(inplace=True), nn.Flatten(),
            nn.Linear(c * roi_feat_size ** 2 // 4, c), nn.ReLU(inplace=True), nn.Linear(c, c),
            nn.ReLU(inplace=True), nn.Linear(c, c * ns * per_pointset_point))

    def forward(self, x, image_embeddings, image_positional_embeddings, roi_img_ids):
        img_bs, roi_bs = image_embeddings.shape[0], x.shape[0]
        size = image_embeddings.shape[-2:]
        pe = self.point_emb(x)
        pe = einops.rearrange(pe, 'b (n c) -> b n c', n=self.per_pointset_point)
        if self.with_sincos:
            pe = torch.sin(pe[..., ::2]) + pe[..., 1::2]
        sparse = pe.unsqueeze(1)
        num_roi = torch.bincount(roi_img_ids.long())
        num_roi = torch.cat([num_roi, torch.zeros(img_bs - len(num_roi), dtype=num_roi.dtype)])
        dense = self.no_mask_embed.weight.reshape(1, -1, 1, 1).expand(roi_bs, -1, size[0], size[1])
        img = image_embeddings.repeat_interleave(num_roi, dim=0)
        ipe = image_positional_embeddings.repeat_interleave(num_roi, dim=0)
        # transformers>=5 returns (masks, iou); 4.38.1 also took output_attentions (models.py:1685-1694)
        masks, iou = self.mask_decoder.mask_decoder(
            image_embeddings=img, image_positional_embeddings=ipe, sparse_prompt_embeddings=sparse,
            dense_prompt_embeddings=dense, multimask_output=False)
        h, w = masks.shape[-2:]
        return masks.reshape(roi_bs, -1, h, w), iou.reshape(roi_bs, -1), sparse


class RoIHead(nn.Module):
    def __init__(self, num_classes):
        super().__init__()
        self.bbox_head = BBoxHead(num_classes=num_classes)
        self.mask_head = MaskHead()


class AnchorOracle(nn.Module):
    """RSPrompterAnchor (MaskRCNN) predict path, configs/rsprompter/_base_/rsprompter_anchor.py."""

    def __init__(self, arch='base', num_classes=10, select_layers=None, hidden_channels=32,
                 test_cfg=None, peft512=False):
        """peft512=True: the rsprompter_anchor-*-peft-512.py tree (MMPretrainSamVisionEncoder at 512 px with
        LoRA on qkv + PseudoFeatureAggregator(hidden 512)), models.py:812-878,943-984."""
        super().__init__()
        depth = hf_sam.ARCH[arch]['num_hidden_layers']
        select_layers = list(select_layers) if select_layers is not None else list(range(1, depth + 1, 2))
        self.arch, self.num_classes, self.peft512 = arch, num_classes, peft512
        self.shared_image_embedding = _Wrap('shared_image_embedding', hf_sam.build_positional_embedding(arch))
        self.neck = nn.Module()
        if peft512:
            from .vitsam import PeftWrapped, ViTSAM
            self.backbone = _Wrap('vision_encoder', PeftWrapped(ViTSAM(arch, 512, lora=True)))
            self.neck.feature_aggregator = PseudoAggregator(256, 512, 256)
        else:
            self.backbone = _Wrap('vision_encoder', hf_sam.build_vision_encoder(arch))
            self.neck.feature_aggregator = FeatureAggregator(arch, hidden_channels, 256, select_layers)
        self.neck.feature_spliter = SimpleFPN()
        self.rpn_head = RPNHead()
        self.roi_head = RoIHead(num_classes)
        self.strides = [4, 8, 16, 32, 64]
        self.anchor_scales, self.anchor_ratios = [4, 8], [0.5, 1.0, 2.0]
        self.test_cfg = test_cfg or dict(
            rpn=dict(nms_pre=1000, max_per_img=1000, iou_threshold=0.7, min_bbox_size=0),
            rcnn=dict(score_thr=0.05, iou_threshold=0.5, max_per_img=100, mask_thr_binary=0.5))
        self.eval()

    # ---- stages (each returns plain tensors so tests can compare stage by stage) ----
    @torch.no_grad()
    def extract_feat(self, batch_inputs):
        """models.py:97-114."""
        if self.peft512:
            hidden = self.backbone.vision_encoder(batch_inputs)      # 1-tuple (models.py:102-104)
            emb = hidden[0]
        else:
            emb, hidden = hf_sam.run_vision_encoder(self.backbone.vision_encoder, batch_inputs)
        G = self.shared_image_embedding.shared_image_embedding.positional_embedding
        ipe = glue.image_wide_positional_embeddings(G, emb.shape[-1]).repeat(emb.shape[0], 1, 1, 1)
        agg = self.neck.feature_aggregator(hidden)
        x = self.neck.feature_spliter(agg)
        return x, emb, ipe, dict(hidden_states=hidden, aggregated=agg)

    @torch.no_grad()
    def rpn_predict(self, x, metas):
        """base_dense_head.py:171-289 + rpn_head.py:134-304, rescale=False."""
        cls, reg = self.rpn_head(x[:5])
        sizes = [c.shape[-2:] for c in cls]
        priors = glue.grid_priors(sizes, self.strides, self.anchor_scales, self.anchor_ratios)
        c = self.test_cfg['rpn']
        out = []
        for i, meta in enumerate(metas):
            out.append(glue.rpn_predict_single([s[i] for s in cls], [r[i] for r in reg], priors,
                                               meta['img_shape'], c['nms_pre'], c['max_per_img'],
                                               c['iou_threshold'], c['min_bbox_size']))
        return out, dict(cls=cls, reg=reg)

    @torch.no_grad()
    def add_extra_pe(self, x):
        """models.py:1566-1574: PE computed at level-0 size, bilinearly resized per level."""
        bs, _, h, w = x[0].shape
        pe = glue.sine_positional_encoding(bs, h, w, num_feats=128)
        return tuple(xi + F.interpolate(pe, size=xi.shape[-2:], mode='bilinear', align_corners=False)
                     for xi in x)

    @torch.no_grad()
    def bbox_predict(self, x_pe, proposals, metas):
        """standard_roi_head.py:293-363 + bbox_head.py:476-571 (rescale=False because with_mask)."""
        rois = torch.cat([torch.cat([p.new_full((p.shape[0], 1), i), p], 1)
                          for i, p in enumerate(proposals)], 0)
        feats = glue.roi_extract(x_pe[:4], rois, 7, self.strides[:4])
        cls_score, bbox_pred = self.roi_head.bbox_head(feats)
        c = self.test_cfg['rcnn']
        out, start = [], 0
        for i, (p, meta) in enumerate(zip(proposals, metas)):
            n = p.shape[0]
            cs, bp, roi = cls_score[start:start + n], bbox_pred[start:start + n], rois[start:start + n]
            start += n
            if n == 0:
                out.append(dict(bboxes=p.new_zeros((0, 4)), scores=p.new_zeros(0),
                                labels=torch.zeros(0, dtype=torch.long), cand=torch.zeros(0, dtype=torch.long)))
                continue
            dets, labels, cand = glue.bbox_head_predict_single(roi, cs, bp, meta['img_shape'], self.num_classes,
                                                               c['score_thr'], c['iou_threshold'], c['max_per_img'])
            out.append(dict(bboxes=dets[:, :4], scores=dets[:, 4], labels=labels, cand=cand))
        return out, dict(rois=rois, roi_feats=feats, cls_score=cls_score, bbox_pred=bbox_pred)

    @torch.no_grad()
    def mask_predict(self, x_pe, dets, metas, emb, ipe, rescale=True):
        """models.py:1511-1550, 1383-1409, 1659-1698, 1746-1784."""
        bboxes = [d['bboxes'] for d in dets]
        rois = torch.cat([torch.cat([b.new_full((b.shape[0], 1), i), b], 1) for i, b in enumerate(bboxes)], 0)
        results, trace = [], dict(mask_rois=rois)
        if rois.shape[0] == 0:
            for d, meta in zip(dets, metas):
                h, w = meta['ori_shape'][:2]
                results.append(dict(d, masks=torch.zeros((0, h, w), dtype=torch.bool)))
            return results, trace
        feats = glue.roi_extract(x_pe[:4], rois, 14, self.strides[:4])
        low_res, iou, sparse = self.roi_head.mask_head(feats, emb, ipe, rois[:, 0])
        trace.update(mask_feats=feats, low_res_masks=low_res, iou_predictions=iou, sparse_embeddings=sparse)
        start = 0
        probs = []
        for d, meta in zip(dets, metas):
            n = d['bboxes'].shape[0]
            mp = low_res[start:start + n]
            start += n
            if n == 0:
                h, w = meta['ori_shape'][:2]
                results.append(dict(d, masks=torch.zeros((0, h, w), dtype=torch.bool)))
                probs.append(None)
                continue
            masks, bb, prob = glue.mask_postprocess_single(
                mp, d['bboxes'], meta, self.test_cfg['rcnn']['mask_thr_binary'], rescale)
            results.append(dict(bboxes=bb, scores=d['scores'], labels=d['labels'], masks=masks))
            probs.append(prob)
        trace['mask_probs'] = probs
        return results, trace

    @torch.no_grad()
    def predict(self, batch_inputs, metas, rescale=True):
        """models.py:148-170.  returns (per-image result dicts, trace of every stage boundary)."""
        x, emb, ipe, t0 = self.extract_feat(batch_inputs)
        props, t1 = self.rpn_predict(x, metas)
        x_pe = self.add_extra_pe(x)
        dets, t2 = self.bbox_predict(x_pe, [p['bboxes'] for p in props], metas)
        results, t3 = self.mask_predict(x_pe, dets, metas, emb, ipe, rescale)
        trace = dict(fpn=x, image_embeddings=emb, image_pe=ipe, proposals=props, dets=dets, x_pe=x_pe)
        for t in (t0, t1, t2, t3):
            trace.update(t)
        return results, trace
