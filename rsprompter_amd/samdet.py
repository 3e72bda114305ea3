"""`SAMDet` on HIP kernels: a Faster R-CNN R50-FPN detector whose boxes prompt SAM.

Reference: SAMDet mmdet/rsprompter/models.py:1061-1215, RSSamModel :719-741, configs/rsprompter/_base_/samdet.py
(detector :56-167, model :169-178); ResNet mmdet/models/backbones/resnet.py:103-300 / :371-672, ResLayer
mmdet/models/layers/res_layer.py:12-107, FPN mmdet/models/necks/fpn.py:15-222, FasterRCNN / TwoStageDetector
two_stage.py:23-195; HF SamModel / SamPromptEncoder (transformers 4.38.1 modeling_sam.py:595-700, :1165-1330).
Same registry names, ctor kwargs and `state_dict` keys.  Feature maps are channels-last ([B*H*W, C] matrices): 1x1 convs
are GEMMs, 3x3 and strided 1x1 convs implicit GEMMs (rsp_gemm), eval-mode BatchNorm is folded into the weights at pack
time, ReLU (and the ReLU after the shortcut, RSP_ACT_RELU_POST) runs in the GEMM epilogue; the stem, the max pooling,
the FPN top-down step and the box prompt have their own kernels (csrc/resnet.hip)."""
import torch

from . import debug, ops
from .detectors import BaseDetectorHIP, SAMSegMaskRCNN
from .necks import conv3x3_weight, fold_bn
from .nnutil import HIPModule, add_param, infer_sam_arch, load_checkpoint_into, nchw_view, nhwc_view
from .registry import MODELS
from .sam_decoder import SamMaskDecoderHIP, _PosEmb, _PromptEncoder, image_wide_table
from .sam_encoder import SamVisionEncoderHIP
from .structures import InstanceData


def _add_bn(root, name, c):
    add_param(root, name + '.weight', (c,), 1.0)
    add_param(root, name + '.bias', (c,))
    add_param(root, name + '.running_mean', (c,), buffer=True)
    add_param(root, name + '.running_var', (c,), 1.0, buffer=True)
    add_param(root, name + '.num_batches_tracked', buffer=True, tensor=torch.zeros((), dtype=torch.long))


def _g(root, dotted):
    for p in dotted.split('.'):
        root = getattr(root, p)
    return root


@MODELS.register_module()
class ResNet(HIPModule):
    """Bottleneck ResNets (depth 50 / 101 / 152) as the reference configures them: plain 7x7 stem, BN, no DCN / plugins.
    `norm_eval` / `frozen_stages` only matter for training; inference always uses the running statistics."""
    arch_settings = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}

    def __init__(self, depth, in_channels=3, stem_channels=None, base_channels=64, num_stages=4, strides=(1, 2, 2, 2),
                 dilations=(1, 1, 1, 1), out_indices=(0, 1, 2, 3), style='pytorch', deep_stem=False, avg_down=False,
                 frozen_stages=-1, conv_cfg=None, norm_cfg=None, norm_eval=True, dcn=None,
                 stage_with_dcn=(False, False, False, False), plugins=None, with_cp=False, zero_init_residual=True,
                 pretrained=None, init_cfg=None):
        super().__init__()
        if depth not in self.arch_settings:
            raise NotImplementedError(f'ResNet depth {depth}: only the Bottleneck depths {sorted(self.arch_settings)} '
                                      'are implemented (the reference configures depth 50)')
        norm_type = (norm_cfg or dict(type='BN'))['type']
        if (in_channels != 3 or (stem_channels or base_channels) != 64 or base_channels != 64 or deep_stem or avg_down
                or dcn is not None or plugins is not None or conv_cfg is not None or norm_type not in ('BN', 'SyncBN')
                or any(d != 1 for d in dilations) or style != 'pytorch'):
            raise NotImplementedError('ResNet: only the configuration of configs/rsprompter/_base_/samdet.py:58-68 '
                                      '(7x7 stem, 64 base channels, BN, style="pytorch", no dilation / DCN / plugins) is implemented')
        assert 1 <= num_stages <= 4 and max(out_indices) < num_stages
        self.depth, self.style, self.num_stages = depth, style, num_stages
        self.out_indices, self.strides = tuple(out_indices), tuple(strides[:num_stages])
        self.stage_blocks = self.arch_settings[depth][:num_stages]
        add_param(self, 'conv1.weight', (64, 3, 7, 7))
        _add_bn(self, 'bn1', 64)
        inplanes = 64
        self.block_cfg = []                                      # (name, inplanes, planes, stride, has_downsample)
        for i, nb in enumerate(self.stage_blocks):
            planes, stride = 64 * 2 ** i, self.strides[i]
            for j in range(nb):
                s = stride if j == 0 else 1
                down = j == 0 and (s != 1 or inplanes != planes * 4)
                name = f'layer{i + 1}.{j}'
                add_param(self, f'{name}.conv1.weight', (planes, inplanes, 1, 1))
                _add_bn(self, f'{name}.bn1', planes)
                add_param(self, f'{name}.conv2.weight', (planes, planes, 3, 3))
                _add_bn(self, f'{name}.bn2', planes)
                add_param(self, f'{name}.conv3.weight', (planes * 4, planes, 1, 1))
                _add_bn(self, f'{name}.bn3', planes * 4)
                if down:
                    add_param(self, f'{name}.downsample.0.weight', (planes * 4, inplanes, 1, 1))
                    _add_bn(self, f'{name}.downsample.1', planes * 4)
                self.block_cfg.append((name, inplanes, planes, s, down))
                inplanes = planes * 4

    def _pack(self):
        w, b = fold_bn(self.conv1.weight, None, self.bn1)
        P = dict(stem=(w.reshape(64, -1).t().contiguous(), b.contiguous()), blocks=[])
        for name, _, _, _, down in self.block_cfg:
            blk = _g(self, name)
            w1, b1 = fold_bn(blk.conv1.weight, None, blk.bn1)
            w2, b2 = fold_bn(blk.conv2.weight, None, blk.bn2)
            w3, b3 = fold_bn(blk.conv3.weight, None, blk.bn3)
            pk = dict(c1=ops.PackedWeight(w1.reshape(w1.shape[0], -1), b1), c2=ops.PackedWeight(conv3x3_weight(w2), b2),
                      c3=ops.PackedWeight(w3.reshape(w3.shape[0], -1), b3), down=None)
            if down:
                wd, bd = fold_bn(getattr(blk.downsample, '0').weight, None, getattr(blk.downsample, '1'))
                pk['down'] = ops.PackedWeight(wd.reshape(wd.shape[0], -1), bd)
            P['blocks'].append(pk)
        self._packed = P

    @staticmethod
    def _conv1x1(x, w, stride, **kw):
        """x NHWC -> ([rows, N], (B, Ho, Wo))."""
        B, H, W, C = x.shape
        if stride == 1:
            return ops.gemm(x.view(B * H * W, C), w, **kw), (B, H, W)
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        return ops.gemm(x, w, conv=(1, stride, 0), **kw), (B, Ho, Wo)

    def _block(self, x, pk, planes, stride):
        """resnet.py:268-300: relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1 x))))))) + identity)."""
        s1, s2 = 1, stride                  # style='pytorch': the stride sits on the 3x3 conv (resnet.py:135-140)
        o, (B, H1, W1) = self._conv1x1(x, pk['c1'], s1, act=ops.ACT_RELU)
        H2, W2 = (H1 - 1) // s2 + 1, (W1 - 1) // s2 + 1
        o = ops.gemm(o.view(B, H1, W1, planes), pk['c2'], act=ops.ACT_RELU, conv=(3, s2, 1))
        if pk['down'] is not None:
            idt, _ = self._conv1x1(x, pk['down'], stride)
        else:
            idt = x.view(-1, x.shape[-1])
        o = ops.gemm(o, pk['c3'], res=idt, act=ops.ACT_RELU_POST)
        return o.view(B, H2, W2, planes * 4)

    def forward(self, x):
        if self._packed is None:
            self._pack()
        P = self._packed
        if not x.is_contiguous():
            x = x.contiguous()
        y = ops.resnet_stem(x, *P['stem'])                       # conv1 + bn1 + relu (resnet.py:640-647), NHWC
        y = ops.maxpool_nhwc(y, 3, 2, 1)
        outs, k = [], 0
        for i, nb in enumerate(self.stage_blocks):
            for _ in range(nb):
                _, _, planes, stride, _ = self.block_cfg[k]
                y = self._block(y, P['blocks'][k], planes, stride)
                k += 1
            if i in self.out_indices:
                outs.append(nchw_view(y))
        return tuple(outs)


@MODELS.register_module()
class FPN(HIPModule):
    """necks/fpn.py:15-222 as the Faster R-CNN configs use it: 1x1 laterals and 3x3 output convs with bias and no norm /
    activation, nearest top-down pathway, extra levels by stride-2 subsampling (`F.max_pool2d(x, 1, stride=2)`)."""

    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 relu_before_extra_convs=False, no_norm_on_lateral=False, conv_cfg=None, norm_cfg=None, act_cfg=None,
                 upsample_cfg=None, init_cfg=None):
        super().__init__()
        upsample_cfg = dict(upsample_cfg or dict(mode='nearest'))
        if (start_level != 0 or end_level not in (-1, len(in_channels) - 1) or add_extra_convs or conv_cfg is not None
                or norm_cfg is not None or act_cfg is not None or upsample_cfg != dict(mode='nearest')):
            raise NotImplementedError('FPN: only the configuration of configs/rsprompter/_base_/samdet.py:70-74 '
                                      '(all levels, max-pool extra levels, no norm / activation, nearest upsampling)')
        self.in_channels, self.out_channels = list(in_channels), out_channels
        self.num_ins, self.num_outs = len(in_channels), num_outs
        assert num_outs >= self.num_ins
        for i, c in enumerate(self.in_channels):
            add_param(self, f'lateral_convs.{i}.conv.weight', (out_channels, c, 1, 1))
            add_param(self, f'lateral_convs.{i}.conv.bias', (out_channels,))
            add_param(self, f'fpn_convs.{i}.conv.weight', (out_channels, out_channels, 3, 3))
            add_param(self, f'fpn_convs.{i}.conv.bias', (out_channels,))

    def _pack(self):
        P = dict(lat=[], out=[])
        for i in range(self.num_ins):
            lc, fc = _g(self, f'lateral_convs.{i}.conv'), _g(self, f'fpn_convs.{i}.conv')
            lw = lc.weight.detach()
            P['lat'].append(ops.PackedWeight(lw.reshape(lw.shape[0], -1), lc.bias.detach()))
            P['out'].append(ops.PackedWeight(conv3x3_weight(fc.weight.detach()), fc.bias.detach()))
        self._packed = P

    def forward(self, inputs):
        assert len(inputs) == self.num_ins
        if self._packed is None:
            self._pack()
        P, co = self._packed, self.out_channels
        lats = []
        for i, x in enumerate(inputs):
            x = nhwc_view(x)
            B, H, W, C = x.shape
            lats.append(ops.gemm(x.view(B * H * W, C), P['lat'][i]).view(B, H, W, co))
        for i in range(self.num_ins - 1, 0, -1):                 # fpn.py:190-204
            ops.upsample_nearest_add_(lats[i - 1], lats[i])
        outs = []
        for i, l in enumerate(lats):
            B, H, W, _ = l.shape
            outs.append(ops.gemm(l, P['out'][i], conv=(3, 1, 1)).view(B, H, W, co))
        for _ in range(self.num_outs - self.num_ins):
            outs.append(ops.pool2(outs[-1], 1))                  # fpn.py:213-214
        self._last_laterals = debug.keep(lats)
        return tuple(nchw_view(o) for o in outs)


@MODELS.register_module()
class FasterRCNN(SAMSegMaskRCNN):
    """mmdet/models/detectors/faster_rcnn.py over TwoStageDetector (two_stage.py:23-195): backbone -> neck -> RPN ->
    StandardRoIHead (bbox branch only, so `predict(rescale=True)` rescales the boxes, bbox_head.py:549-552)."""

    def extract_feat(self, batch_inputs):
        x = self.backbone(batch_inputs)
        return self.neck(x) if self.neck is not None else x


class SamImageSegmentationOutput:
    """the two fields of HF's SamImageSegmentationOutput this path produces."""

    def __init__(self, iou_scores, pred_masks):
        self.iou_scores, self.pred_masks = iou_scores, pred_masks

    def __getitem__(self, i):
        return (self.iou_scores, self.pred_masks)[i]


class _SamPromptEncoderFull(_PromptEncoder):
    """HF SamPromptEncoder parameters (modeling_sam.py:595-646); the box path (:647-656) is what SAMDet uses."""

    def __init__(self, hid=256):
        super().__init__()
        add_param(self, 'shared_embedding.positional_embedding', (2, hid // 2))
        for i in range(4):
            add_param(self, f'point_embed.{i}.weight', (1, hid))
        add_param(self, 'not_a_point_embed.weight', (1, hid))


class SamModelHIP(HIPModule):
    """HF `SamModel` (modeling_sam.py:1165-1330) for the call SAMDet makes: `pixel_values` (or `image_embeddings`) +
    `input_boxes`, `multimask_output=False`.  Points / mask prompts / multimask outputs are not on the reference's path."""

    def __init__(self, arch='huge', image_size=1024):
        super().__init__()
        self.image_size = image_size
        self.shared_image_embedding = _PosEmb()
        self.vision_encoder = SamVisionEncoderHIP(arch=arch, image_size=image_size)
        self.prompt_encoder = _SamPromptEncoderFull()
        self.mask_decoder = SamMaskDecoderHIP()
        self._pe_cache = {}

    def _apply(self, fn, *a, **kw):
        self._pe_cache = {}
        return super()._apply(fn, *a, **kw)

    def get_image_wide_positional_embeddings(self):
        G = self.shared_image_embedding.positional_embedding
        size = self.vision_encoder.grid
        key = (size, G.data_ptr(), G._version)
        if key not in self._pe_cache:
            self._pe_cache = {key: image_wide_table(G, size)}
        return self._pe_cache[key]

    def get_image_embeddings(self, pixel_values):
        return self.vision_encoder(pixel_values, output_hidden_states=False)[0]

    @torch.no_grad()
    def forward(self, pixel_values=None, input_points=None, input_labels=None, input_boxes=None, input_masks=None,
                image_embeddings=None, multimask_output=True, attention_similarity=None, target_embedding=None,
                output_attentions=None, output_hidden_states=None, return_dict=None, **kwargs):
        if pixel_values is None and image_embeddings is None:
            raise ValueError('Either pixel_values or image_embeddings must be provided.')
        if pixel_values is not None and image_embeddings is not None:
            raise ValueError('Only one of pixel_values and image_embeddings can be provided.')
        if (input_points is not None or input_labels is not None or input_masks is not None or multimask_output
                or attention_similarity is not None or target_embedding is not None or input_boxes is None):
            raise NotImplementedError('SamModel on HIP: only box prompts with multimask_output=False '
                                      '(the call of SAMDet.predict, models.py:1174-1178)')
        if input_boxes.dim() != 3 or input_boxes.shape[-1] != 4:
            raise ValueError('The input_boxes must be a 3D tensor. Of shape `batch_size`, `nb_boxes`, `4`.')
        if image_embeddings is None:
            image_embeddings = self.get_image_embeddings(pixel_values)
        B, nb = input_boxes.shape[:2]
        if image_embeddings.shape[0] != B:
            raise ValueError('You should provide as many bounding boxes as input_points (batch_size)')
        pe = self.prompt_encoder
        sparse = ops.sam_embed_boxes(input_boxes.reshape(B * nb, 4).to(torch.float32),
                                     pe.shared_embedding.positional_embedding, getattr(pe.point_embed, '2').weight,
                                     getattr(pe.point_embed, '3').weight, (self.image_size, self.image_size))
        roi_img = torch.arange(B, dtype=torch.int32, device=sparse.device).repeat_interleave(nb)
        masks, iou = self.mask_decoder.decode(image_embeddings, self.get_image_wide_positional_embeddings(), sparse,
                                              pe.no_mask_embed.weight.reshape(-1), roi_img)
        h, w = masks.shape[-2:]
        return SamImageSegmentationOutput(iou.view(B, nb, 1), masks.view(B, nb, 1, h, w))


@MODELS.register_module()
class RSSamModel(HIPModule):
    """models.py:719-741."""

    def __init__(self, hf_pretrain_name, extra_config=None, init_cfg=None):
        super().__init__()
        if extra_config:
            raise NotImplementedError('RSSamModel: extra_config overrides are not used by any samdet config')
        self.sam_model = SamModelHIP(arch=infer_sam_arch(hf_pretrain_name))
        if init_cfg is not None:
            load_checkpoint_into(self.sam_model, init_cfg.get('checkpoint'), revise_keys=[(r'^module\.', '')])
            self.sam_model.is_init = True

    def forward(self, *args, **kwargs):
        return self.sam_model(*args, **kwargs)


@MODELS.register_module()
class SAMDet(BaseDetectorHIP):
    """models.py:1061-1215 (inference).  The reference calls the segmentor once per image; here the SAM encoder runs once
    on the whole batch and ONE decoder pass serves the boxes of every image (same per-image arithmetic)."""

    def __init__(self, detector, segmentor, data_preprocessor=None, test_cfg=None, init_cfg=None):
        super().__init__()
        self.data_preprocessor = MODELS.build(data_preprocessor or dict(type='DetDataPreprocessor'))
        self.detector = MODELS.build(detector)
        self.segmentor = MODELS.build(segmentor)
        self.test_cfg = test_cfg
        self.eval()

    def extract_feat(self, batch_inputs):
        pass

    def _forward(self, batch_inputs, batch_data_samples=None):
        pass

    @torch.no_grad()
    def predict(self, batch_inputs, batch_data_samples, rescale=True):
        oracle_on = self.test_cfg is not None and self.test_cfg.get('oracle_on', True)      # models.py:1159
        batch_data_samples = self.detector.predict(batch_inputs, batch_data_samples, rescale=rescale)
        dev = batch_inputs.device
        boxes_in, counts = [], []
        for s in batch_data_samples:
            if oracle_on:                                                                   # models.py:1100-1104
                inst = InstanceData()
                inst.bboxes = s.gt_instances.bboxes.to(device=dev, dtype=torch.float32)
                inst.labels = s.gt_instances.labels.to(dev)
                inst.scores = torch.ones_like(inst.labels, dtype=torch.float32)
                s.pred_instances = inst
            bboxes = s.pred_instances.bboxes
            counts.append(int(bboxes.shape[0]))
            if counts[-1]:
                sf = s.metainfo['scale_factor']
                boxes_in.append(ops.scale_boxes(bboxes, (sf[0], sf[1], sf[0], sf[1])))      # models.py:1169-1172
        low = None
        if sum(counts):
            sam = self.segmentor.sam_model if hasattr(self.segmentor, 'sam_model') else self.segmentor
            # the reference skips the segmentor for an image without boxes (models.py:1166): only the images that have
            # prompts go through the ViT; prompt set r belongs to row roi_img[r] of that sub-batch
            have = [i for i, n in enumerate(counts) if n]
            sub = batch_inputs if len(have) == len(counts) else batch_inputs[torch.tensor(have, device=dev)]
            emb = sam.get_image_embeddings(sub)
            # one decoder pass over all images
            pe = sam.prompt_encoder
            allb = torch.cat(boxes_in, 0)
            sparse = ops.sam_embed_boxes(allb, pe.shared_embedding.positional_embedding,
                                         getattr(pe.point_embed, '2').weight, getattr(pe.point_embed, '3').weight,
                                         (sam.image_size, sam.image_size))
            roi_img = torch.repeat_interleave(torch.arange(len(have), dtype=torch.int32),
                                              torch.tensor([counts[i] for i in have])).to(dev)
            low, _ = sam.mask_decoder.decode(emb, sam.get_image_wide_positional_embeddings(), sparse,
                                             pe.no_mask_embed.weight.reshape(-1), roi_img, want_iou=False)
            low = low[:, 0]                                                                 # [R, 256, 256]
            self._last_seg = debug.keep(lambda: dict(embeddings=emb, boxes=allb, sparse=sparse, low_res=low))
        start = 0
        for s, n in zip(batch_data_samples, counts):
            meta = s.metainfo
            oh, ow = meta['ori_shape'][:2]
            if n == 0:
                s.pred_instances.masks = torch.zeros((0, oh, ow), dtype=torch.bool, device=dev)
                continue
            sf = meta['scale_factor']
            crop = (int(oh * sf[1]), int(ow * sf[0]))                                       # models.py:1182-1183
            s.pred_instances.masks = ops.mask_post_logits(low[start:start + n].contiguous(), tuple(meta['img_shape'][:2]),
                                                          crop, (oh, ow), 0.0)
            start += n
        return batch_data_samples
