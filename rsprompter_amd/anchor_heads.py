"""Anchor prompter of RSPrompter on HIP kernels: RPN head, RoI extractors, box head,
prompt (mask) head.  Registry names / ctor kwargs follow the reference configs
(configs/rsprompter/_base_/rsprompter_anchor.py:86-200).

Reference code mirrored here:
  AnchorGenerator            mmdet/models/task_modules/prior_generators/anchor_generator.py:69-301
  DeltaXYWHBBoxCoder         mmdet/models/task_modules/coders/delta_xywh_bbox_coder.py:71-131,264-361
  RPNHead                    mmdet/models/dense_heads/rpn_head.py:22-304, base_dense_head.py:171-289
  SingleRoIExtractor         mmdet/models/roi_heads/roi_extractors/single_level_roi_extractor.py:44-119
  Shared2FCBBoxHead          mmdet/models/roi_heads/bbox_heads/convfc_bbox_head.py:163-233, bbox_head.py:476-571
  RSPrompterAnchorRoIPromptHead / RSPrompterAnchorMaskHead   mmdet/rsprompter/models.py:1366-1784
"""
import math

import numpy as np
import torch

from . import debug, ops
from .necks import conv3x3_weight, fold_bn
from .nnutil import HIPModule, add_param, nchw_view, nhwc_view
from .registry import MODELS, TASK_UTILS
from .structures import InstanceData


# ----------------------------------------------------------------------------- task utils
@TASK_UTILS.register_module()
class AnchorGenerator:
    def __init__(self, strides, ratios, scales=None, base_sizes=None, scale_major=True,
                 octave_base_scale=None, scales_per_octave=None, centers=None, center_offset=0.,
                 use_box_type=False):
        if center_offset != 0:
            assert centers is None
        if not (0 <= center_offset <= 1):
            raise ValueError(f'center_offset should be in range [0, 1], {center_offset} is given.')
        pair = lambda s: tuple(s) if isinstance(s, (tuple, list)) else (s, s)  # noqa: E731
        self.strides = [pair(s) for s in strides]
        self.base_sizes = [min(s) for s in self.strides] if base_sizes is None else list(base_sizes)
        assert len(self.base_sizes) == len(self.strides)
        assert ((octave_base_scale is not None and scales_per_octave is not None) ^ (scales is not None))
        if scales is not None:
            self.scales = torch.Tensor(scales)
        else:
            octave_scales = np.array([2 ** (i / scales_per_octave) for i in range(scales_per_octave)])
            self.scales = torch.Tensor(octave_scales * octave_base_scale)
        self.ratios = torch.Tensor(ratios)
        self.scale_major, self.centers, self.center_offset = scale_major, centers, center_offset
        self.base_anchors = self.gen_base_anchors()

    @property
    def num_base_priors(self):
        return [b.size(0) for b in self.base_anchors]

    num_base_anchors = num_base_priors

    @property
    def num_levels(self):
        return len(self.strides)

    def gen_base_anchors(self):
        return [self.gen_single_level_base_anchors(bs, self.scales, self.ratios,
                                                   None if self.centers is None else self.centers[i])
                for i, bs in enumerate(self.base_sizes)]

    def gen_single_level_base_anchors(self, base_size, scales, ratios, center=None):
        w = h = base_size
        if center is None:
            x_c, y_c = self.center_offset * w, self.center_offset * h
        else:
            x_c, y_c = center
        h_ratios = torch.sqrt(ratios)
        w_ratios = 1 / h_ratios
        if self.scale_major:
            ws = (w * w_ratios[:, None] * scales[None, :]).view(-1)
            hs = (h * h_ratios[:, None] * scales[None, :]).view(-1)
        else:
            ws = (w * scales[:, None] * w_ratios[None, :]).view(-1)
            hs = (h * scales[:, None] * h_ratios[None, :]).view(-1)
        return torch.stack([x_c - 0.5 * ws, y_c - 0.5 * hs, x_c + 0.5 * ws, y_c + 0.5 * hs], dim=-1)

    def single_level_grid_priors(self, featmap_size, level_idx, dtype=torch.float32, device='cpu'):
        base = self.base_anchors[level_idx].to(device).to(dtype)
        fh, fw = featmap_size
        sw, sh = self.strides[level_idx]
        sx = torch.arange(0, fw, device=device).to(dtype) * sw
        sy = torch.arange(0, fh, device=device).to(dtype) * sh
        xx = sx.repeat(fh)
        yy = sy.view(-1, 1).repeat(1, fw).view(-1)
        shifts = torch.stack([xx, yy, xx, yy], dim=-1)
        return (base[None, :, :] + shifts[:, None, :]).view(-1, 4)

    def grid_priors(self, featmap_sizes, dtype=torch.float32, device='cpu'):
        assert self.num_levels == len(featmap_sizes)
        return [self.single_level_grid_priors(featmap_sizes[i], i, dtype, device)
                for i in range(self.num_levels)]


@TASK_UTILS.register_module()
class DeltaXYWHBBoxCoder:
    encode_size = 4

    def __init__(self, target_means=(0., 0., 0., 0.), target_stds=(1., 1., 1., 1.), clip_border=True,
                 add_ctr_clamp=False, ctr_clamp=32, use_box_type=False):
        self.means, self.stds = tuple(target_means), tuple(target_stds)
        self.clip_border, self.add_ctr_clamp, self.ctr_clamp = clip_border, add_ctr_clamp, ctr_clamp
        assert len(self.means) == 4 and len(self.stds) == 4
        if use_box_type:
            raise NotImplementedError('DeltaXYWHBBoxCoder(use_box_type=True): the heads hand tensors to the decode kernels')
        self.max_ratio = float(np.float32(np.abs(np.log(16 / 1000))))


def _img_hw(metas, device):
    return torch.tensor([[float(m['img_shape'][0]), float(m['img_shape'][1])] for m in metas],
                        dtype=torch.float32, device=device)


def _metas_of(batch_data_samples):
    return [s.metainfo if hasattr(s, 'metainfo') else s for s in batch_data_samples]


# ----------------------------------------------------------------------------- RPN
@MODELS.register_module()
class RPNHead(HIPModule):
    def __init__(self, in_channels, num_classes=1, feat_channels=256, anchor_generator=None, bbox_coder=None,
                 num_convs=1, loss_cls=None, loss_bbox=None, train_cfg=None, test_cfg=None, init_cfg=None,
                 reg_decoded_bbox=False, **kwargs):
        super().__init__()
        assert num_classes == 1 and num_convs == 1
        self.in_channels, self.feat_channels = in_channels, feat_channels
        self.prior_generator = TASK_UTILS.build(anchor_generator)
        self.bbox_coder = TASK_UTILS.build(bbox_coder or dict(type='DeltaXYWHBBoxCoder'))
        self.num_base_priors = self.prior_generator.num_base_priors[0]
        # anchor_head.py:60-77: the default loss_cls is the sigmoid one, a given loss_cls without the key means softmax;
        # sigmoid objectness = one channel per anchor, softmax = [fg, bg] per anchor
        self.use_sigmoid_cls = True if loss_cls is None else bool(loss_cls.get('use_sigmoid', False))
        self.cls_out_channels = 1 if self.use_sigmoid_cls else 2
        self.test_cfg = test_cfg
        A, co = self.num_base_priors, self.cls_out_channels
        add_param(self, 'rpn_conv.weight', (feat_channels, in_channels, 3, 3))
        add_param(self, 'rpn_conv.bias', (feat_channels,))
        add_param(self, 'rpn_cls.weight', (A * co, feat_channels, 1, 1))
        add_param(self, 'rpn_cls.bias', (A * co,))
        add_param(self, 'rpn_reg.weight', (A * 4, feat_channels, 1, 1))
        add_param(self, 'rpn_reg.bias', (A * 4,))
        # columns of the packed 1x1 head: [0, A) objectness logit, [A, 5A) deltas, softmax only: [5A, 7A) the raw [fg, bg] scores
        cols = A * 5 + (0 if self.use_sigmoid_cls else 2 * A)
        self.LD = (cols + 31) // 32 * 32

    def _pack(self):
        A, fc = self.num_base_priors, self.feat_channels
        w = torch.zeros((self.LD, fc), dtype=torch.float32, device=self.rpn_cls.weight.device)
        b = torch.zeros((self.LD,), dtype=torch.float32, device=w.device)
        wc, bc = self.rpn_cls.weight.detach().reshape(-1, fc), self.rpn_cls.bias.detach()
        if self.use_sigmoid_cls:
            w[:A], b[:A] = wc, bc
        else:
            # rpn_head.py:193-197 `cls_score.softmax(-1)[:, :-1]` over [fg, bg] (channel a * 2 + k, :186-187) is
            # sigmoid(fg - bg): the difference is folded into the 1x1 convolution, the selection kernels stay the sigmoid ones
            w[:A], b[:A] = wc[0::2] - wc[1::2], bc[0::2] - bc[1::2]
            w[5 * A:7 * A], b[5 * A:7 * A] = wc, bc
        w[A:5 * A] = self.rpn_reg.weight.detach().reshape(4 * A, fc)
        b[A:5 * A] = self.rpn_reg.bias.detach()
        self._packed = dict(conv=ops.PackedWeight(conv3x3_weight(self.rpn_conv.weight.detach()), self.rpn_conv.bias),
                            head=ops.PackedWeight(w, b), selector=None)

    def _heads(self, x):
        """per level: [B*H*W, LD] with objectness in cols [0,A) and deltas in [A,5A) (rpn_head.py:80-97)."""
        if self._packed is None:
            self._pack()
        P = self._packed
        heads, sizes = [], []
        for xi in x:
            f = nhwc_view(xi)
            B, H, W, C = f.shape
            t = ops.gemm(f, P['conv'], act=ops.ACT_RELU, conv=(3, 1, 1))
            heads.append(ops.gemm(t, P['head']))
            sizes.append((H, W))
        return heads, sizes

    def forward(self, x):
        heads, sizes = self._heads(x)
        A = self.num_base_priors
        B = x[0].shape[0]
        c0, c1 = (0, A) if self.use_sigmoid_cls else (5 * A, 7 * A)
        cls = [h.view(B, H, W, self.LD)[..., c0:c1].permute(0, 3, 1, 2) for h, (H, W) in zip(heads, sizes)]
        reg = [h.view(B, H, W, self.LD)[..., A:5 * A].permute(0, 3, 1, 2) for h, (H, W) in zip(heads, sizes)]
        return cls, reg

    def predict(self, x, batch_data_samples, rescale=False):
        """base_dense_head.py:171-199 -> rpn_head.py:134-304.  Returns a list of InstanceData
        (bboxes [k,4], scores [k], labels [k]); `rescale` must be False (models.py:156-157)."""
        assert not rescale
        metas = _metas_of(batch_data_samples)
        heads, sizes = self._heads(x)
        out = self.select(heads, sizes, metas)
        return self._to_instances(out)

    def select(self, heads, sizes, metas, cfg=None):
        cfg = cfg or self.test_cfg
        dev = heads[0].device
        P = self._packed
        nms_cfg = cfg['nms'] if isinstance(cfg, dict) else cfg.nms
        key = (cfg['nms_pre'], cfg['max_per_img'], nms_cfg['iou_threshold'], cfg.get('min_bbox_size', -1))
        if P['selector'] is None or P['selector'][0] != key:
            base = torch.stack(self.prior_generator.base_anchors, 0)
            strides = [s[0] for s in self.prior_generator.strides]
            sel = ops.RpnSelector(base, strides, key[0], key[1], key[2], key[3], self.bbox_coder, dev)
            P['selector'] = (key, sel)
        return P['selector'][1](heads, sizes, self.LD, _img_hw(metas, dev))

    @staticmethod
    def _to_instances(out):
        counts = out['count'].tolist()         # the one host sync of the RPN stage
        res = []
        for b, n in enumerate(counts):
            r = InstanceData()
            r.bboxes = out['boxes'][b, :n]
            r.scores = out['scores'][b, :n]
            r.labels = torch.zeros((n,), dtype=torch.long, device=out['boxes'].device)
            r.level_ids = out['ids'][b, :n]
            r.anchor_index = out['src'][b, :n]
            res.append(r)
        return res


# ----------------------------------------------------------------------------- RoI extractor
@MODELS.register_module()
class SingleRoIExtractor(HIPModule):
    def __init__(self, roi_layer, out_channels, featmap_strides, finest_scale=56, init_cfg=None):
        super().__init__()
        cfg = dict(roi_layer)
        typ = cfg.pop('type', 'RoIAlign')
        if typ != 'RoIAlign':
            raise NotImplementedError(f'roi layer {typ}')
        out_size = cfg.pop('output_size')
        self.output_size = out_size if isinstance(out_size, int) else out_size[0]
        if cfg.pop('sampling_ratio', 0) != 0 or not cfg.pop('aligned', True) or cfg.pop('pool_mode', 'avg') != 'avg':
            raise NotImplementedError('only RoIAlign(sampling_ratio=0, aligned=True, avg) is implemented')
        self.out_channels, self.featmap_strides, self.finest_scale = out_channels, list(featmap_strides), finest_scale

    @property
    def num_inputs(self):
        return len(self.featmap_strides)

    def forward(self, feats, rois, roi_scale_factor=None, pes=None):
        """feats: logical-NCHW channels-last levels; returns logical [K, C, P, P] (channels-last view)."""
        assert roi_scale_factor is None
        f = [nhwc_view(x) for x in feats]
        out = ops.roi_align(f, pes, rois.to(torch.float32).contiguous(), self.output_size, self.featmap_strides,
                            self.finest_scale)
        return out.permute(0, 3, 1, 2)


# ----------------------------------------------------------------------------- box head
@MODELS.register_module()
class Shared2FCBBoxHead(HIPModule):
    def __init__(self, in_channels=256, fc_out_channels=1024, roi_feat_size=7, num_classes=80, bbox_coder=None,
                 reg_class_agnostic=False, loss_cls=None, loss_bbox=None, with_avg_pool=False, init_cfg=None,
                 **kwargs):
        super().__init__()
        if reg_class_agnostic or with_avg_pool:
            raise NotImplementedError
        self.in_channels, self.fc_out, self.roi_feat_size = in_channels, fc_out_channels, roi_feat_size
        self.num_classes = num_classes
        self.bbox_coder = TASK_UTILS.build(bbox_coder or dict(type='DeltaXYWHBBoxCoder',
                                                             target_stds=(0.1, 0.1, 0.2, 0.2)))
        k0 = in_channels * roi_feat_size * roi_feat_size
        add_param(self, 'shared_fcs.0.weight', (fc_out_channels, k0))
        add_param(self, 'shared_fcs.0.bias', (fc_out_channels,))
        add_param(self, 'shared_fcs.1.weight', (fc_out_channels, fc_out_channels))
        add_param(self, 'shared_fcs.1.bias', (fc_out_channels,))
        add_param(self, 'fc_cls.weight', (num_classes + 1, fc_out_channels))
        add_param(self, 'fc_cls.bias', (num_classes + 1,))
        add_param(self, 'fc_reg.weight', (4 * num_classes, fc_out_channels))
        add_param(self, 'fc_reg.bias', (4 * num_classes,))
        n = 5 * num_classes + 1
        self.LD = (n + 31) // 32 * 32

    def _pack(self):
        s, c = self.roi_feat_size, self.in_channels
        fc0 = getattr(self.shared_fcs, '0')
        fc1 = getattr(self.shared_fcs, '1')
        # RoI features arrive as (y, x, c); torch flattens [c, y, x] -> permute the weight columns once
        w0 = fc0.weight.detach().view(self.fc_out, c, s, s).permute(0, 2, 3, 1).reshape(self.fc_out, -1)
        nc = self.num_classes
        w = torch.zeros((self.LD, self.fc_out), dtype=torch.float32, device=w0.device)
        b = torch.zeros((self.LD,), dtype=torch.float32, device=w0.device)
        w[:nc + 1] = self.fc_cls.weight.detach()
        w[nc + 1:5 * nc + 1] = self.fc_reg.weight.detach()
        b[:nc + 1] = self.fc_cls.bias.detach()
        b[nc + 1:5 * nc + 1] = self.fc_reg.bias.detach()
        self._packed = dict(fc0=ops.PackedWeight(w0, fc0.bias), fc1=ops.PackedWeight(fc1.weight, fc1.bias),
                            head=ops.PackedWeight(w, b))

    def _head(self, roi_feats):
        """roi_feats: logical [K, C, s, s] channels-last -> [K, LD] (cls logits | per-class deltas)."""
        if self._packed is None:
            self._pack()
        P = self._packed
        x = nhwc_view(roi_feats)
        K = x.shape[0]
        x = ops.gemm(x.reshape(K, -1), P['fc0'], act=ops.ACT_RELU)
        x = ops.gemm(x, P['fc1'], act=ops.ACT_RELU)
        return ops.gemm(x, P['head'])

    def forward(self, x):
        h = self._head(x)
        nc = self.num_classes
        return h[:, :nc + 1], h[:, nc + 1:5 * nc + 1]


# ----------------------------------------------------------------------------- prompt (mask) head
@MODELS.register_module()
class RSPrompterAnchorMaskHead(HIPModule):
    def __init__(self, mask_decoder, in_channels, roi_feat_size=14, per_pointset_point=5, with_sincos=True,
                 multimask_output=False, attention_similarity=None, target_embedding=None, output_attentions=None,
                 class_agnostic=False, loss_mask=None, init_cfg=None, *args, **kwargs):
        super().__init__()
        self.in_channels, self.roi_feat_size = in_channels, roi_feat_size
        self.per_pointset_point, self.with_sincos = per_pointset_point, with_sincos
        self.multimask_output, self.class_agnostic = multimask_output, class_agnostic
        self.mask_decoder = MODELS.build(mask_decoder)
        # the reference builds a whole RSSamPromptEncoder and keeps its no_mask_embed (models.py:1628-1635)
        add_param(self, 'no_mask_embed.weight', (1, 256))
        c, ns = in_channels, (2 if with_sincos else 1)
        add_param(self, 'point_emb.0.weight', (c, c, 3, 3))
        add_param(self, 'point_emb.0.bias', (c,))
        add_param(self, 'point_emb.1.weight', (c,), 1.0)
        add_param(self, 'point_emb.1.bias', (c,))
        add_param(self, 'point_emb.1.running_mean', (c,), buffer=True)
        add_param(self, 'point_emb.1.running_var', (c,), 1.0, buffer=True)
        add_param(self, 'point_emb.1.num_batches_tracked', buffer=True, tensor=torch.zeros((), dtype=torch.long))
        add_param(self, 'point_emb.4.weight', (c, c * roi_feat_size ** 2 // 4))
        add_param(self, 'point_emb.4.bias', (c,))
        add_param(self, 'point_emb.6.weight', (c, c))
        add_param(self, 'point_emb.6.bias', (c,))
        add_param(self, 'point_emb.8.weight', (c * ns * per_pointset_point, c))
        add_param(self, 'point_emb.8.bias', (c * ns * per_pointset_point,))

    def _pack(self):
        pe = self.point_emb
        c, s2 = self.in_channels, self.roi_feat_size // 2
        w, b = fold_bn(getattr(pe, '0').weight, getattr(pe, '0').bias, getattr(pe, '1'))
        fc4 = getattr(pe, '4')
        w4 = fc4.weight.detach().view(c, c, s2, s2).permute(0, 2, 3, 1).reshape(c, -1)   # (c,y,x) -> (y,x,c)
        self._packed = dict(conv=ops.PackedWeight(conv3x3_weight(w), b),
                            fc4=ops.PackedWeight(w4, fc4.bias),
                            fc6=ops.PackedWeight(getattr(pe, '6').weight, getattr(pe, '6').bias),
                            fc8=ops.PackedWeight(getattr(pe, '8').weight, getattr(pe, '8').bias))

    def point_embeddings(self, x):
        """models.py:1669-1672: [R, C, 14, 14] -> sparse prompts [R, n_points, 256]."""
        if self._packed is None:
            self._pack()
        P = self._packed
        f = nhwc_view(x)
        R = f.shape[0]
        y = ops.gemm(f, P['conv'], act=ops.ACT_RELU, conv=(3, 2, 1))         # conv s2 + BN + ReLU
        y = ops.gemm(y.view(R, -1), P['fc4'], act=ops.ACT_RELU)
        y = ops.gemm(y, P['fc6'], act=ops.ACT_RELU)
        y = ops.gemm(y, P['fc8'])
        n = self.per_pointset_point
        if self.with_sincos:
            y = ops.sincos_pairs(y)          # sin(x[..., ::2]) + x[..., 1::2] on interleaved pairs
        return y.view(R, n, -1)

    def forward(self, x, image_embeddings, image_positional_embeddings, roi_img_ids=None):
        """models.py:1659-1698.  roi_img_ids must be sorted by image (they are: bbox2roi order)."""
        roi_bs = x.shape[0]
        sparse = self.point_embeddings(x)
        roi_img = roi_img_ids.to(torch.int32).contiguous()
        dec = self.mask_decoder.mask_decoder
        # multimask_output=True (HF:537-542): three masks / iou scores per prompt set instead of one
        low_res, iou = dec.decode(image_embeddings, image_positional_embeddings, sparse,
                                  self.no_mask_embed.weight.reshape(-1), roi_img,
                                  multimask_output=bool(self.multimask_output))
        h, w = low_res.shape[-2:]
        return low_res.reshape(roi_bs, -1, h, w), iou.reshape(roi_bs, -1)

    def predict_by_feat(self, mask_preds, results_list, batch_img_metas, rcnn_test_cfg, rescale=False,
                        activate_map=False):
        """fcn_mask_head.py:218-276 + models.py:1746-1784."""
        assert len(mask_preds) == len(results_list) == len(batch_img_metas)
        for img_id, (mp, results, meta) in enumerate(zip(mask_preds, results_list, batch_img_metas)):
            h, w = meta['ori_shape'][:2]
            if results.bboxes.shape[0] == 0:
                results.masks = torch.zeros((0, h, w), dtype=torch.bool, device=results.bboxes.device)
                continue
            results.masks = self._predict_by_feat_single(mp, results, meta, rcnn_test_cfg, rescale)
        return results_list

    def _predict_by_feat_single(self, mask_preds, results, img_meta, rcnn_test_cfg, rescale=False,
                                want_prob=False):
        sf_w, sf_h = img_meta['scale_factor']
        img_h, img_w = img_meta['ori_shape'][:2]
        if rescale:
            results.bboxes = ops.div_boxes(results.bboxes, (sf_w, sf_h, sf_w, sf_h))
        else:
            img_h = np.round(img_h * np.float32(sf_h)).astype(np.int32)
            img_w = np.round(img_w * np.float32(sf_w)).astype(np.int32)
        thr = rcnn_test_cfg['mask_thr_binary'] if isinstance(rcnn_test_cfg, dict) else rcnn_test_cfg.mask_thr_binary
        if mask_preds.shape[1] != 1:
            # the reference's own post-processing is written for one mask per instance: with the three masks of
            # multimask_output=True its `.squeeze(1)` is a no-op and the second F.interpolate gets a 5-D tensor
            # (models.py:1771-1778 raises); forward() / mode='tensor' return the three masks
            raise ValueError(f'mask post-processing expects one mask per instance, got {mask_preds.shape[1]} '
                             '(multimask_output=True has no defined predict path in the reference either)')
        Hb, Wb = img_meta['batch_input_shape']
        crop = (min(int(img_h * sf_h), Hb), min(int(img_w * sf_w), Wb))
        h, w = img_meta['ori_shape'][:2]
        if thr < 0:
            # models.py:1779-1783 "for visualization and debugging": the resized probabilities as uint8, (p * 255) truncated
            _, prob = ops.mask_post(mask_preds[:, 0].contiguous(), (Hb, Wb), crop, (h, w), 0.5, True)
            soft = (prob * 255).to(torch.uint8)
            return (soft, prob) if want_prob else soft
        return ops.mask_post(mask_preds[:, 0].contiguous(), (Hb, Wb), crop, (h, w), float(thr), want_prob)


# ----------------------------------------------------------------------------- RoI head
def _sine_pe(h, w, num_feats, temperature=10000, scale=2 * math.pi, eps=1e-6):
    """SinePositionalEncoding(normalize=True) on an all-valid mask (positional_encoding.py:60-110):
    input independent, evaluated once on the host with the reference's fp32 expression."""
    not_mask = torch.ones((1, h, w), dtype=torch.int)
    y_embed = not_mask.cumsum(1, dtype=torch.float32)
    x_embed = not_mask.cumsum(2, dtype=torch.float32)
    y_embed = y_embed / (y_embed[:, -1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(num_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / num_feats)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).view(1, h, w, -1)
    pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).view(1, h, w, -1)
    return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


@MODELS.register_module()
class RSPrompterAnchorRoIPromptHead(HIPModule):
    def __init__(self, with_extra_pe=False, bbox_roi_extractor=None, bbox_head=None, mask_roi_extractor=None,
                 mask_head=None, shared_head=None, train_cfg=None, test_cfg=None, init_cfg=None):
        super().__init__()
        assert shared_head is None
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.bbox_roi_extractor = MODELS.build(bbox_roi_extractor)
        self.bbox_head = MODELS.build(bbox_head)
        self.mask_roi_extractor = MODELS.build(mask_roi_extractor) if mask_roi_extractor is not None else None
        self.share_roi_extractor = mask_roi_extractor is None
        self.mask_head = MODELS.build(mask_head) if mask_head is not None else None
        self.with_extra_pe = with_extra_pe
        self._pe_cache = {}

    @property
    def with_bbox(self):
        return self.bbox_head is not None

    @property
    def with_mask(self):
        return self.mask_head is not None

    def _apply(self, fn, *a, **kw):
        self._pe_cache = {}
        return super()._apply(fn, *a, **kw)

    def extra_pe_tables(self, x):
        """models.py:1566-1574: PE of the level-0 grid, bilinearly resized to every level; returned
        as per-level NHWC [H, W, C] device tables that RoIAlign adds on the fly."""
        if not self.with_extra_pe:
            return None
        sizes = tuple(tuple(xi.shape[-2:]) for xi in x)
        key = (sizes, str(x[0].device))
        if key not in self._pe_cache:
            c = self.bbox_roi_extractor.out_channels
            pe = _sine_pe(sizes[0][0], sizes[0][1], c // 2)
            tabs = []
            for s in sizes:
                t = torch.nn.functional.interpolate(pe, size=s, mode='bilinear', align_corners=False)
                tabs.append(t[0].permute(1, 2, 0).contiguous().to(x[0].device))
            self._pe_cache = {key: tabs}
        return self._pe_cache[key]

    @staticmethod
    def _rois(box_list):
        """bbox2roi (structures/bbox/transforms.py:82-102)."""
        parts = []
        for i, b in enumerate(box_list):
            idx = torch.full((b.shape[0], 1), float(i), dtype=torch.float32, device=b.device)
            parts.append(torch.cat([idx, b[:, :4]], 1))
        return torch.cat(parts, 0) if parts else torch.zeros((0, 5))

    def predict_bbox(self, x, batch_img_metas, rpn_results_list, rcnn_test_cfg, rescale=False, pes=None):
        """standard_roi_head.py:293-363 + bbox_head.py:476-571 + multiclass_nms."""
        proposals = [r.bboxes for r in rpn_results_list]
        rois = self._rois(proposals)
        dev = rois.device
        nc = self.bbox_head.num_classes
        counts = [int(p.shape[0]) for p in proposals]
        if rois.shape[0] == 0:
            return [self._empty_det(dev) for _ in proposals]
        n = self.bbox_roi_extractor.num_inputs
        feats = self.bbox_roi_extractor(x[:n], rois, pes=None if pes is None else pes[:n])
        head = self.bbox_head._head(feats)
        roi_start = torch.tensor([0] + list(np.cumsum(counts)), dtype=torch.int64)
        nms = rcnn_test_cfg['nms']
        out = ops.bbox_post(head, self.bbox_head.LD, rois, roi_start, _img_hw(batch_img_metas, dev), nc,
                            float(rcnn_test_cfg['score_thr']), self.bbox_head.bbox_coder, float(nms['iou_threshold']),
                            int(rcnn_test_cfg['max_per_img']),
                            scale_factors=[m['scale_factor'] for m in batch_img_metas] if rescale else None)
        kept = out['count'].tolist()            # host sync of the R-CNN stage
        res = []
        for b, k in enumerate(kept):
            r = InstanceData()
            r.bboxes = out['boxes'][b, :k]
            r.scores = out['scores'][b, :k]
            r.labels = out['ids'][b, :k].to(torch.long)
            r.cand_index = out['src'][b, :k]
            res.append(r)
        self._last_bbox_trace = debug.keep(lambda: dict(rois=rois, roi_feats=feats, head=head))   # tests only
        return res

    @staticmethod
    def _empty_det(dev):
        r = InstanceData()
        r.bboxes = torch.zeros((0, 4), device=dev)
        r.scores = torch.zeros((0,), device=dev)
        r.labels = torch.zeros((0,), dtype=torch.long, device=dev)
        return r

    def _mask_forward(self, x, rois, image_embeddings=None, image_positional_embeddings=None, pes=None):
        """models.py:1383-1409."""
        n = self.mask_roi_extractor.num_inputs
        mask_feats = self.mask_roi_extractor(x[:n], rois, pes=None if pes is None else pes[:n])
        mask_preds, iou = self.mask_head(mask_feats, image_embeddings=image_embeddings,
                                         image_positional_embeddings=image_positional_embeddings,
                                         roi_img_ids=rois[:, 0])
        return dict(mask_preds=mask_preds, mask_feats=mask_feats, iou_predictions=iou)

    def predict_mask(self, x, batch_img_metas, results_list, rescale=False, image_embeddings=None,
                     image_positional_embeddings=None, pes=None):
        """models.py:1511-1550."""
        bboxes = [res.bboxes for res in results_list]
        mask_rois = self._rois(bboxes)
        if mask_rois.shape[0] == 0:
            for res, meta in zip(results_list, batch_img_metas):
                h, w = meta['ori_shape'][:2]
                res.masks = torch.zeros((0, h, w), dtype=torch.bool, device=res.bboxes.device)
            return results_list
        mr = self._mask_forward(x, mask_rois, image_embeddings, image_positional_embeddings, pes)
        self._last_mask_trace = debug.keep(lambda: dict(mr, mask_rois=mask_rois))                  # tests only
        mask_preds = mr['mask_preds'].split([len(r) for r in results_list], 0)
        return self.mask_head.predict_by_feat(mask_preds, results_list, batch_img_metas, self.test_cfg,
                                              rescale=rescale)

    def forward(self, x, rpn_results_list, batch_data_samples=None, image_embeddings=None,
                image_positional_embeddings=None):
        """StandardRoIHead.forward (standard_roi_head.py:60-92): raw `(cls_score, bbox_pred, mask_preds)` over the
        proposals of the whole batch, mask branch on the first 100 RoIs, with the extra PE of models.py:1566-1574 and
        the image embeddings the RSPrompter mask head needs (see RSPrompterAnchor._forward)."""
        pes = self.extra_pe_tables(x)
        rois = self._rois([r.bboxes for r in rpn_results_list])
        results = ()
        if self.with_bbox:
            n = self.bbox_roi_extractor.num_inputs
            feats = self.bbox_roi_extractor(x[:n], rois, pes=None if pes is None else pes[:n])
            cls_score, bbox_pred = self.bbox_head(feats)
            results = results + (cls_score, bbox_pred)
        if self.with_mask:
            mr = self._mask_forward(x, rois[:100].contiguous(), image_embeddings, image_positional_embeddings, pes)
            results = results + (mr['mask_preds'],)
        return results

    def predict(self, x, rpn_results_list, batch_data_samples, rescale=False, image_embeddings=None,
                image_positional_embeddings=None):
        """models.py:1553-1593."""
        metas = _metas_of(batch_data_samples)
        pes = self.extra_pe_tables(x)
        bbox_rescale = rescale if not self.with_mask else False
        results = self.predict_bbox(x, metas, rpn_results_list, self.test_cfg, rescale=bbox_rescale, pes=pes)
        if self.with_mask:
            results = self.predict_mask(x, metas, results, rescale=rescale, image_embeddings=image_embeddings,
                                        image_positional_embeddings=image_positional_embeddings, pes=pes)
        return results


# ----------------------------------------------------------------------------- SAMSeg sibling model
@MODELS.register_module()
class FCNMaskHead(HIPModule):
    """mmdet/models/roi_heads/mask_heads/fcn_mask_head.py:27-150 (the mask head of configs/rsprompter/_base_/
    samseg-maskrcnn.py:117-124): num_convs x (3x3 conv + ReLU) -> ConvTranspose2d(k2, s2) + ReLU -> 1x1 conv to
    num_classes logits at 2 x roi_feat_size; `_predict_by_feat_single` pastes them into the image (:276-420)."""

    def __init__(self, num_convs=4, roi_feat_size=14, in_channels=256, conv_kernel_size=3, conv_out_channels=256,
                 num_classes=80, class_agnostic=False, upsample_cfg=None, conv_cfg=None, norm_cfg=None,
                 predictor_cfg=None, loss_mask=None, init_cfg=None):
        super().__init__()
        up = dict(upsample_cfg or dict(type='deconv', scale_factor=2))
        if up.get('type') != 'deconv' or up.get('scale_factor', 2) != 2 or conv_kernel_size != 3 or norm_cfg is not None:
            raise NotImplementedError('FCNMaskHead: 3x3 convs without norm + deconv x2 (the SAMSeg configuration)')
        self.num_convs, self.in_channels, self.conv_out_channels = num_convs, in_channels, conv_out_channels
        self.num_classes, self.class_agnostic = num_classes, class_agnostic
        c = conv_out_channels
        for i in range(num_convs):
            add_param(self, f'convs.{i}.conv.weight', (c, in_channels if i == 0 else c, 3, 3))
            add_param(self, f'convs.{i}.conv.bias', (c,))
        add_param(self, 'upsample.weight', (c if num_convs > 0 else in_channels, c, 2, 2))
        add_param(self, 'upsample.bias', (c,))
        nout = 1 if class_agnostic else num_classes
        add_param(self, 'conv_logits.weight', (nout, c, 1, 1))
        add_param(self, 'conv_logits.bias', (nout,))

    def _pack(self):
        from .necks import conv3x3_weight, convt_weights
        P = dict(convs=[])
        for i in range(self.num_convs):
            cv = _get(self, f'convs.{i}.conv')
            P['convs'].append(ops.PackedWeight(conv3x3_weight(cv.weight.detach()), cv.bias))
        P['up'] = convt_weights(self.upsample.weight, self.upsample.bias)
        nout = self.conv_logits.weight.shape[0]
        P['logits'] = ops.PackedWeight(self.conv_logits.weight.detach().reshape(nout, -1), self.conv_logits.bias)
        self._packed = P

    def forward(self, x):
        """[K, C, s, s] (channels-last) -> mask logits [K, nc, 2s, 2s] (a channels-last view of [K, 2s, 2s, nc])."""
        if self._packed is None:
            self._pack()
        P = self._packed
        y = nhwc_view(x)
        K, h, w, _ = y.shape
        for pw in P['convs']:
            y = ops.gemm(y, pw, act=ops.ACT_RELU, conv=(3, 1, 1)).view(K, h, w, -1)
        y = ops.conv_transpose2x2(y, P['up'][0], P['up'][1], act=ops.ACT_RELU)
        lg = ops.gemm(y.reshape(K * 4 * h * w, -1), P['logits'])
        return nchw_view(lg.view(K, 2 * h, 2 * w, -1))

    def predict_by_feat(self, mask_preds, results_list, batch_img_metas, rcnn_test_cfg, rescale=False,
                        activate_map=False):
        """fcn_mask_head.py:218-276."""
        assert len(mask_preds) == len(results_list) == len(batch_img_metas) and not activate_map
        for mp, results, meta in zip(mask_preds, results_list, batch_img_metas):
            if results.bboxes.shape[0] == 0:
                h, w = meta['ori_shape'][:2]
                if not rescale:
                    sf_w, sf_h = meta['scale_factor']
                    h, w = int(np.round(h * np.float32(sf_h))), int(np.round(w * np.float32(sf_w)))
                results.masks = torch.zeros((0, h, w), dtype=torch.bool, device=results.bboxes.device)
                continue
            results.masks = self._predict_by_feat_single(mp, results, meta, rcnn_test_cfg, rescale)
        return results_list

    def _predict_by_feat_single(self, mask_preds, results, img_meta, rcnn_test_cfg, rescale=False):
        """fcn_mask_head.py:276-420: boxes to the output image's coordinates (in place, like the reference), then
        sigmoid + per-box bilinear paste + threshold in one kernel."""
        sf_w, sf_h = img_meta['scale_factor']
        img_h, img_w = img_meta['ori_shape'][:2]
        if rescale:
            results.bboxes = ops.div_boxes(results.bboxes, (sf_w, sf_h, sf_w, sf_h))
        else:
            img_h = int(np.round(img_h * np.float32(sf_h)))
            img_w = int(np.round(img_w * np.float32(sf_w)))
        thr = rcnn_test_cfg['mask_thr_binary'] if isinstance(rcnn_test_cfg, dict) else rcnn_test_cfg.mask_thr_binary
        lg = nhwc_view(mask_preds)
        labels = None if self.class_agnostic else results.labels
        return ops.paste_masks(lg, labels, results.bboxes, (img_h, img_w), float(thr))


def _get(root, dotted):
    for part in dotted.split('.'):
        root = getattr(root, part)
    return root


@MODELS.register_module()
class StandardRoIHead(RSPrompterAnchorRoIPromptHead):
    """mmdet/models/roi_heads/standard_roi_head.py:19-424 (inference): the RSPrompter RoI head without the extra
    positional encoding and with a mask head that consumes RoI features only (FCNMaskHead)."""

    def __init__(self, bbox_roi_extractor=None, bbox_head=None, mask_roi_extractor=None, mask_head=None,
                 shared_head=None, train_cfg=None, test_cfg=None, init_cfg=None):
        super().__init__(with_extra_pe=False, bbox_roi_extractor=bbox_roi_extractor, bbox_head=bbox_head,
                         mask_roi_extractor=mask_roi_extractor, mask_head=mask_head, shared_head=shared_head,
                         train_cfg=train_cfg, test_cfg=test_cfg, init_cfg=init_cfg)

    def _mask_forward(self, x, rois, image_embeddings=None, image_positional_embeddings=None, pes=None):
        """standard_roi_head.py:257-291."""
        ext = self.mask_roi_extractor if self.mask_roi_extractor is not None else self.bbox_roi_extractor
        n = ext.num_inputs
        mask_feats = ext(x[:n], rois)
        return dict(mask_preds=self.mask_head(mask_feats), mask_feats=mask_feats)
