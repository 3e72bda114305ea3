"""Detector facades with the reference's registry names and forward API.

Reference: RSPrompterAnchor mmdet/rsprompter/models.py:53-170 (MaskRCNN -> TwoStageDetector
two_stage.py:23-113 -> BaseDetector base.py:17-156), DetDataPreprocessor
mmdet/models/data_preprocessors/data_preprocessor.py:110-149 (+ mmengine ImgDataPreprocessor).
Training-only members (`loss`, assigners, samplers, loss_* dicts) are accepted by the
constructors and ignored: this package implements the inference hot path only.
"""
import copy

import torch

from . import debug, ops
from .nnutil import HIPModule
from .registry import MODELS
from .structures import DetDataSample, InstanceData


@MODELS.register_module()
class BatchFixedSizePad:
    """training-only batch augment (data_preprocessor.py:300-377, applied only when training)."""

    def __init__(self, size, img_pad_value=0, pad_mask=False, mask_pad_value=0, pad_seg=False, seg_pad_value=255):
        self.size = tuple(size)


@MODELS.register_module()
class DetDataPreprocessor(HIPModule):
    def __init__(self, mean=None, std=None, pad_size_divisor=1, pad_value=0, pad_mask=False, mask_pad_value=0,
                 pad_seg=False, seg_pad_value=255, bgr_to_rgb=False, rgb_to_bgr=False, boxtype2tensor=True,
                 non_blocking=False, batch_augments=None):
        super().__init__()
        assert not (bgr_to_rgb and rgb_to_bgr)
        self.mean = list(mean) if mean is not None else [0., 0., 0.]
        self.std = list(std) if std is not None else [1., 1., 1.]
        self.swap = bool(bgr_to_rgb or rgb_to_bgr)
        self.pad_size_divisor, self.pad_value = pad_size_divisor, pad_value
        self.batch_augments = batch_augments     # training only
        self.register_buffer('_dev', torch.zeros(1), persistent=False)

    @property
    def device(self):
        return self._dev.device

    def forward(self, data, training=False):
        """data: dict(inputs=list[uint8/float [3,H,W]] or [B,3,H,W], data_samples=list) -> same dict with
        `inputs` = normalised fp32 [B,3,Hp,Wp] on the device and metas updated (data_preprocessor.py:129-134)."""
        assert not training, 'inference path only'
        inputs = data['inputs']
        imgs = list(inputs) if not isinstance(inputs, torch.Tensor) else [im for im in inputs]
        batch = ops.preprocess(imgs, self.mean, self.std, self.swap, self.pad_size_divisor, float(self.pad_value),
                               device=self.device)
        samples = data.get('data_samples')
        if samples is not None:
            shape = tuple(batch.shape[-2:])
            for s, im in zip(samples, imgs):
                s.set_metainfo({'batch_input_shape': shape, 'pad_shape': shape})
        return dict(inputs=batch, data_samples=samples)


class BaseDetectorHIP(HIPModule):
    """BaseDetector.forward / test_step / add_pred_to_datasample (base.py:58-156)."""

    def forward(self, inputs, data_samples=None, mode='tensor'):
        if mode == 'predict':
            return self.predict(inputs, data_samples)
        if mode == 'loss':
            raise NotImplementedError('training (`loss`) is outside the scope of the MI355X inference path')
        if mode == 'tensor':
            return self._forward(inputs, data_samples)
        raise RuntimeError(f'Invalid mode "{mode}". Only supports loss, predict and tensor mode')

    @torch.no_grad()
    def test_step(self, data):
        data = self.data_preprocessor(data, False)
        return self.forward(data['inputs'], data['data_samples'], mode='predict')

    val_step = test_step

    @staticmethod
    def add_pred_to_datasample(data_samples, results_list):
        for s, r in zip(data_samples, results_list):
            s.pred_instances = r
        return data_samples


@MODELS.register_module()
class RSPrompterAnchor(BaseDetectorHIP):
    def __init__(self, shared_image_embedding, decoder_freeze=True, backbone=None, neck=None, rpn_head=None,
                 roi_head=None, train_cfg=None, test_cfg=None, data_preprocessor=None, init_cfg=None):
        super().__init__()
        self.data_preprocessor = MODELS.build(data_preprocessor or dict(type='DetDataPreprocessor'))
        self.backbone = MODELS.build(backbone)
        self.neck = MODELS.build(neck) if neck is not None else None
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        if rpn_head is not None:   # two_stage.py:40-52
            rpn_head_ = copy.deepcopy(dict(rpn_head))
            rpn_head_.update(train_cfg=(train_cfg or {}).get('rpn'), test_cfg=test_cfg['rpn'])
            if rpn_head_.get('num_classes') is None:
                rpn_head_.update(num_classes=1)
            self.rpn_head = MODELS.build(rpn_head_)
        if roi_head is not None:   # two_stage.py:54-61
            roi_head_ = copy.deepcopy(dict(roi_head))
            roi_head_.update(train_cfg=(train_cfg or {}).get('rcnn'), test_cfg=test_cfg['rcnn'])
            self.roi_head = MODELS.build(roi_head_)
        self.shared_image_embedding = MODELS.build(shared_image_embedding)
        self.decoder_freeze = decoder_freeze
        self.eval()

    @property
    def with_rpn(self):
        return hasattr(self, 'rpn_head') and self.rpn_head is not None

    def get_image_wide_positional_embeddings(self, size):
        """models.py:85-95."""
        return self.shared_image_embedding.image_wide(size)

    def extract_feat(self, batch_inputs):
        """models.py:97-114."""
        vision_outputs = self.backbone(batch_inputs)
        if hasattr(vision_outputs, 'hidden_states') and vision_outputs.hidden_states is not None:
            image_embeddings, vision_hidden_states = vision_outputs[0], vision_outputs[1]
        elif isinstance(vision_outputs, tuple):
            image_embeddings, vision_hidden_states = vision_outputs[0], vision_outputs
        else:
            raise NotImplementedError
        pe = self.get_image_wide_positional_embeddings(size=image_embeddings.shape[-1])
        # the reference repeats the table per image (models.py:110-111); it is input independent, so a
        # broadcast view is handed on and the decoder reads batch entry 0
        image_positional_embeddings = pe.expand(image_embeddings.shape[0], -1, -1, -1)
        x = self.neck(vision_hidden_states)
        return x, image_embeddings, image_positional_embeddings

    @torch.no_grad()
    def predict(self, batch_inputs, batch_data_samples, rescale=True):
        """models.py:148-170."""
        x, image_embeddings, image_positional_embeddings = self.extract_feat(batch_inputs)
        self._last_embeddings = debug.keep(image_embeddings)       # parity tests only (rsprompter_amd/debug.py)
        if batch_data_samples[0].get('proposals', None) is None:
            rpn_results_list = self.rpn_head.predict(x, batch_data_samples, rescale=False)
        else:
            rpn_results_list = [s.proposals for s in batch_data_samples]
        results_list = self.roi_head.predict(x, rpn_results_list, batch_data_samples, rescale=rescale,
                                             image_embeddings=image_embeddings,
                                             image_positional_embeddings=image_positional_embeddings)
        return self.add_pred_to_datasample(batch_data_samples, results_list)

    @torch.no_grad()
    def _forward(self, batch_inputs, batch_data_samples=None):
        """`mode='tensor'`: TwoStageDetector._forward (two_stage.py:115-145) -> StandardRoIHead.forward
        (standard_roi_head.py:60-92): raw head outputs without post-processing, `(cls_score, bbox_pred, mask_preds)` over
        the RPN proposals of the whole batch, masks for the first 100 RoIs.
        Reference quirk (not reproduced): RSPrompterAnchor.extract_feat returns a 3-tuple (models.py:97-114) which the
        inherited `_forward` hands to the RPN as if it were the feature pyramid, and `_mask_forward` is called without
        the image embeddings -- the reference's own tensor mode raises.  This is the evident intent of those lines with
        the tuple unpacked and the embeddings passed on."""
        x, image_embeddings, image_positional_embeddings = self.extract_feat(batch_inputs)
        if self.with_rpn:
            samples = batch_data_samples
            if samples is None:       # `tensor` mode is also what FLOP counters call, without data samples
                shape = tuple(batch_inputs.shape[-2:])
                samples = [DetDataSample(metainfo=dict(img_shape=shape, batch_input_shape=shape, pad_shape=shape,
                                                       ori_shape=shape, scale_factor=(1.0, 1.0)))
                           for _ in range(batch_inputs.shape[0])]
            rpn_results_list = self.rpn_head.predict(x, samples, rescale=False)
        else:
            assert batch_data_samples[0].get('proposals', None) is not None
            rpn_results_list = [s.proposals for s in batch_data_samples]
        roi_outs = self.roi_head.forward(x, rpn_results_list, batch_data_samples, image_embeddings=image_embeddings,
                                         image_positional_embeddings=image_positional_embeddings)
        return (roi_outs,)


@MODELS.register_module()
class SAMSegMaskRCNN(RSPrompterAnchor):
    """mmdet/rsprompter/models.py:1219-1244: Mask R-CNN on the SAM encoder + RSFPN (configs/rsprompter/_base_/
    samseg-maskrcnn.py) -- the same backbone / neck / RPN / bbox-head kernels as RSPrompterAnchor with the standard
    `StandardRoIHead` + `FCNMaskHead`; no prompt generator, no SAM mask decoder, no image-wide PE."""

    def __init__(self, backbone=None, neck=None, rpn_head=None, roi_head=None, train_cfg=None, test_cfg=None,
                 data_preprocessor=None, init_cfg=None):
        BaseDetectorHIP.__init__(self)
        self.data_preprocessor = MODELS.build(data_preprocessor or dict(type='DetDataPreprocessor'))
        self.backbone = MODELS.build(backbone)
        self.neck = MODELS.build(neck) if neck is not None else None
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        rpn_head_ = copy.deepcopy(dict(rpn_head))
        rpn_head_.update(train_cfg=(train_cfg or {}).get('rpn'), test_cfg=test_cfg['rpn'])
        if rpn_head_.get('num_classes') is None:
            rpn_head_.update(num_classes=1)
        self.rpn_head = MODELS.build(rpn_head_)
        roi_head_ = copy.deepcopy(dict(roi_head))
        roi_head_.update(train_cfg=(train_cfg or {}).get('rcnn'), test_cfg=test_cfg['rcnn'])
        self.roi_head = MODELS.build(roi_head_)
        self.eval()

    def extract_feat(self, batch_inputs):
        """models.py:1233-1244: hidden states -> neck; only the pyramid is returned."""
        vo = self.backbone(batch_inputs)
        if hasattr(vo, 'hidden_states') and vo.hidden_states is not None:
            hs = vo[1]
        elif isinstance(vo, tuple):
            hs = vo
        else:
            raise NotImplementedError
        return self.neck(hs)

    @torch.no_grad()
    def predict(self, batch_inputs, batch_data_samples, rescale=True):
        """TwoStageDetector.predict (two_stage.py:147-195)."""
        x = self.extract_feat(batch_inputs)
        if batch_data_samples[0].get('proposals', None) is None:
            rpn_results_list = self.rpn_head.predict(x, batch_data_samples, rescale=False)
        else:
            rpn_results_list = [s.proposals for s in batch_data_samples]
        results_list = self.roi_head.predict(x, rpn_results_list, batch_data_samples, rescale=rescale)
        return self.add_pred_to_datasample(batch_data_samples, results_list)

    @torch.no_grad()
    def _forward(self, batch_inputs, batch_data_samples=None):
        x = self.extract_feat(batch_inputs)
        shape = tuple(batch_inputs.shape[-2:])
        samples = batch_data_samples or [DetDataSample(metainfo=dict(img_shape=shape, batch_input_shape=shape,
                                                                      pad_shape=shape, ori_shape=shape,
                                                                      scale_factor=(1.0, 1.0)))
                                         for _ in range(batch_inputs.shape[0])]
        rpn_results_list = self.rpn_head.predict(x, samples, rescale=False)
        return (self.roi_head.forward(x, rpn_results_list, samples),)
