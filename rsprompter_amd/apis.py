"""Caller APIs of the inference path: `init_detector`, `inference_detector` and a `DetInferencer`
with the reference's call surface, on top of a GPU front end for the test pipeline.

Reference: mmdet/apis/inference.py:31-119 (`init_detector`), :122-193 (`inference_detector`),
mmdet/apis/det_inferencer.py:298-417 (`DetInferencer.__call__`: chunked `preprocess -> forward -> postprocess`),
and the test pipeline every RSPrompter config declares (configs/rsprompter/_base_/rsprompter_anchor.py:231-241):
    LoadImageFromFile(to_float32=True) -> Resize(scale=crop_size, keep_ratio=True)
    -> Pad(size=crop_size, pad_val=dict(img=(0.406*255, 0.456*255, 0.485*255), masks=0)) -> [LoadAnnotations]
    -> PackDetInputs(meta_keys=('img_id', 'img_path', 'ori_shape', 'img_shape', 'scale_factor', ...))

What runs where: the image file is decoded on the host (PIL here; the reference decodes with cv2 -- decoders are not
part of the hot path and may differ by a grey level on some JPEG blocks); the decoded HWC array goes to the device as
uint8 and everything else -- float conversion, bilinear resize with cv2's INTER_LINEAR arithmetic, constant padding
-- is ONE kernel (`rsp_resize_pad`), after which `model.test_step` runs the existing DetDataPreprocessor kernel.
Annotation loading / visualisation are not part of the hot path and are skipped (a pipeline entry that is not one of
the transforms above and is not annotation-related raises).
"""
import copy
import os

import numpy as np
import torch

from . import ops
from .config import Config
from .structures import DetDataSample

_SKIPPED = ('LoadAnnotations', 'mmdet.LoadAnnotations')


def rescale_size(old_wh, scale):
    """mmcv.image.geometric.rescale_size (mmcv 2.1, the `Resize(keep_ratio=True)` path): the largest size that fits
    inside `scale` keeping the aspect ratio; mmcv rounds with +0.5 (`_scale_size`)."""
    w, h = old_wh
    if isinstance(scale, (int, float)):
        sf = float(scale)
    else:
        sf = min(max(scale) / max(h, w), min(scale) / min(h, w))
    return (int(w * float(sf) + 0.5), int(h * float(sf) + 0.5)), sf


class TestPipeline:
    """The test pipeline as a device front end.  Built from the pipeline cfg list; callable on dict(img=ndarray) or
    dict(img_path=str) like mmcv's Compose; returns dict(inputs=fp32 [3, H, W] device tensor (BGR order, 0..255 range,
    exactly what PackDetInputs would hand on), data_samples=DetDataSample with the metainfo keys of `meta_keys`)."""

    def __init__(self, pipeline, device='cuda:0'):
        self.device = torch.device(device)
        self.scale, self.keep_ratio, self.pad_size, self.pad_val = None, True, None, (0.0, 0.0, 0.0)
        self.meta_keys = ('img_id', 'img_path', 'ori_shape', 'img_shape', 'scale_factor')
        self.to_float32 = False
        for t in pipeline:
            name = str(t['type']).split('.')[-1]
            if name in ('LoadImageFromFile', 'LoadImageFromNDArray', 'LoadImageFromWebcam'):
                self.to_float32 = bool(t.get('to_float32', False))
            elif name == 'Resize':
                self.scale, self.keep_ratio = tuple(t['scale']), bool(t.get('keep_ratio', False))
                if t.get('interpolation', 'bilinear') != 'bilinear' or t.get('backend', 'cv2') != 'cv2':
                    raise NotImplementedError('Resize: only cv2 bilinear (the mmcv default) is implemented')
            elif name == 'Pad':
                if t.get('size') is None or t.get('padding_mode', 'constant') != 'constant':
                    raise NotImplementedError('Pad: only a fixed `size` with constant padding is implemented')
                self.pad_size = tuple(t['size'])                      # (w, h)
                pv = t.get('pad_val', dict(img=0))
                pv = pv.get('img', 0) if isinstance(pv, dict) else pv
                self.pad_val = tuple(float(v) for v in pv) if isinstance(pv, (tuple, list)) else (float(pv),) * 3
            elif name == 'PackDetInputs':
                self.meta_keys = tuple(t.get('meta_keys', self.meta_keys))
            elif name in ('LoadAnnotations',):
                continue                                              # ground truth: not an input of predict
            else:
                raise NotImplementedError(f'test pipeline transform {t["type"]} is not implemented by the HIP front end')

    @staticmethod
    def _decode(path):
        from PIL import Image
        with Image.open(path) as im:
            rgb = np.asarray(im.convert('RGB'))
        return np.ascontiguousarray(rgb[:, :, ::-1])                  # BGR like cv2.imread / mmcv.imfrombytes

    def __call__(self, data):
        data = dict(data)
        img = data.get('img')
        if img is None:
            img = self._decode(data['img_path'])
        if isinstance(img, np.ndarray):
            img = torch.from_numpy(np.ascontiguousarray(img))
        if img.dim() != 3 or img.shape[2] != 3:
            raise ValueError('expected an [H, W, 3] image')
        h, w = int(img.shape[0]), int(img.shape[1])
        dimg = img.to(self.device, non_blocking=True)
        if self.scale is not None:
            if self.keep_ratio:
                (nw, nh), _ = rescale_size((w, h), self.scale)
            else:
                nw, nh = int(self.scale[0]), int(self.scale[1])
        else:
            nw, nh = w, h
        pw, ph = self.pad_size if self.pad_size is not None else (nw, nh)
        pw, ph = max(pw, nw), max(ph, nh)                             # mmcv.impad never crops
        inputs = ops.resize_pad(dimg, (nh, nw), (ph, pw), self.pad_val)
        meta = dict(img_id=data.get('img_id', 0), img_path=data.get('img_path'), ori_shape=(h, w),
                    # Resize sets img_shape to the resized size, mmcv's Pad then overwrites it with the padded one
                    img_shape=(ph, pw) if self.pad_size is not None else (nh, nw),
                    scale_factor=(nw / w, nh / h), pad_shape=(ph, pw, 3), keep_ratio=self.keep_ratio)
        sample = DetDataSample(metainfo={k: meta[k] for k in self.meta_keys if k in meta})
        return dict(inputs=inputs, data_samples=sample)


def get_test_pipeline_cfg(cfg):
    """mmdet/utils/misc.py::get_test_pipeline_cfg: the pipeline of the test dataloader's dataset."""
    ds = cfg['test_dataloader']['dataset']
    while 'dataset' in ds and 'pipeline' not in ds:
        ds = ds['dataset']
    return copy.deepcopy(ds['pipeline'])


def init_detector(config, checkpoint=None, palette='none', device='cuda:0', cfg_options=None):
    """mmdet/apis/inference.py:31-119: build the model of a config file (or Config), load a checkpoint, attach `cfg`,
    move to `device`, eval mode."""
    from . import build_model
    from .nnutil import load_checkpoint_into
    if isinstance(config, (str, os.PathLike)):
        config = Config.fromfile(str(config))
    elif not isinstance(config, dict):
        raise TypeError(f'config must be a filename or Config object, but got {type(config)}')
    if cfg_options is not None:
        config.merge_from_dict(cfg_options)
    # init_cfg of the sub-modules points at the pretrained SAM files; without them on disk the loaders are skipped
    model = build_model(config)
    if checkpoint is not None:
        load_checkpoint_into(model, checkpoint)
    model.cfg = config
    model.to(device)
    model.eval()
    return model


def inference_detector(model, imgs, test_pipeline=None, text_prompt=None, custom_entities=False):
    """mmdet/apis/inference.py:122-193: str / ndarray or a list of them -> DetDataSample (or a list of them)."""
    if text_prompt:
        raise NotImplementedError('text prompts belong to grounding detectors, not to RSPrompter')
    is_batch = isinstance(imgs, (list, tuple))
    if not is_batch:
        imgs = [imgs]
    if test_pipeline is None:
        dev = next(model.parameters()).device
        test_pipeline = TestPipeline(get_test_pipeline_cfg(model.cfg), device=dev)
    result_list = []
    for img in imgs:
        data_ = dict(img=img, img_id=0) if isinstance(img, (np.ndarray, torch.Tensor)) else dict(img_path=img, img_id=0)
        data_ = test_pipeline(data_)
        data_['inputs'] = [data_['inputs']]
        data_['data_samples'] = [data_['data_samples']]
        with torch.no_grad():
            result_list.append(model.test_step(data_)[0])
    return result_list if is_batch else result_list[0]


class DetInferencer:
    """mmdet/apis/det_inferencer.py: `DetInferencer(model=cfg_or_path, weights=..., device=...)(inputs, batch_size=1)`
    -> dict(predictions=[...], visualization=[]).  Inputs: path / ndarray / list of them / a directory.  Prediction
    dicts follow `pred2dict` (det_inferencer.py:573-627): labels, scores, bboxes (+ masks as COCO RLE)."""

    def __init__(self, model=None, weights=None, device='cuda:0', scope='mmdet', palette='none', show_progress=False):
        if isinstance(model, torch.nn.Module):
            self.model = model
        else:
            self.model = init_detector(model, weights, device=device)
        self.pipeline = TestPipeline(get_test_pipeline_cfg(self.model.cfg), device=next(self.model.parameters()).device)

    @staticmethod
    def _inputs_to_list(inputs):
        if isinstance(inputs, str) and os.path.isdir(inputs):
            exts = ('.jpg', '.jpeg', '.png', '.bmp', '.tif', '.tiff')
            return [os.path.join(inputs, f) for f in sorted(os.listdir(inputs)) if f.lower().endswith(exts)]
        return list(inputs) if isinstance(inputs, (list, tuple)) else [inputs]

    def pred2dict(self, sample, with_rle=True):
        from .rle import encode_mask_results
        p = sample.pred_instances
        out = dict(labels=p.labels.tolist(), scores=p.scores.tolist(), bboxes=p.bboxes.tolist())
        if with_rle and hasattr(p, 'masks') and p.masks is not None:
            out['masks'] = encode_mask_results(p.masks) if len(p.labels) else []
        return out

    @torch.no_grad()
    def __call__(self, inputs, batch_size=1, return_datasamples=False, no_save_pred=True, **kwargs):
        items = self._inputs_to_list(inputs)
        preds = []
        for i in range(0, len(items), batch_size):
            chunk = [self.pipeline(dict(img=x, img_id=i + j) if isinstance(x, (np.ndarray, torch.Tensor))
                                   else dict(img_path=x, img_id=i + j)) for j, x in enumerate(items[i:i + batch_size])]
            data = dict(inputs=[c['inputs'] for c in chunk], data_samples=[c['data_samples'] for c in chunk])
            for s in self.model.test_step(data):
                preds.append(s if return_datasamples else self.pred2dict(s))
        return dict(predictions=preds, visualization=[])
