"""-m gpu: caller APIs on the device: the resize + pad front-end kernel against the oracle's
cv2 / mmcv restatement, and `inference_detector` end to end on a real image (a crop of the reference's
tests/data/color.jpg, tests/golden/color_jpg_crop_bgr.npz) against the CPU oracle fed by the CPU pipeline."""
import os
import sys
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _match import match_detections  # noqa: E402

MEAN = [123.675, 116.28, 103.53]
STD = [58.395, 57.12, 57.375]
PAD = (0.406 * 255, 0.456 * 255, 0.485 * 255)


def _image():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'color_jpg_crop_bgr.npz'))['bgr']


def test_resize_pad_kernel_matches_cv2_restatement(dev):
    from oracle import pipeline as op
    from rsprompter_amd import ops
    from rsprompter_amd.apis import rescale_size
    img = _image()
    cases = [(img, (1024, 1024))]
    g = np.random.default_rng(5)
    for (h, w) in ((37, 53), (600, 401), (1500, 900), (64, 64)):
        cases.append((g.integers(0, 256, size=(h, w, 3), dtype=np.uint8), (512, 512)))
    for arr, scale in cases:
        h, w = arr.shape[:2]
        (nw, nh), _ = rescale_size((w, h), scale)
        ref, meta = op.run_test_pipeline(arr, scale=scale, pad_size=scale, pad_val=PAD)
        got = ops.resize_pad(torch.from_numpy(arr).to(dev), (nh, nw), (scale[1], scale[0]), PAD).cpu().numpy()
        err = float(np.abs(got - ref).max())
        print(f'resize_pad {w}x{h} -> {nw}x{nh} in {scale}: max abs err {err:.2e} (0..255 scale)')
        assert got.shape == ref.shape and err < 1e-3
        # float32 input (to_float32 images) gives the same result as uint8
        got_f = ops.resize_pad(torch.from_numpy(arr.astype(np.float32)).to(dev), (nh, nw), (scale[1], scale[0]), PAD)
        assert float((got_f.cpu() - torch.from_numpy(got)).abs().max()) == 0.0
        # fused DetDataPreprocessor arithmetic: BGR -> RGB, (x - mean) / std
        fused = ops.resize_pad(torch.from_numpy(arr).to(dev), (nh, nw), (scale[1], scale[0]), PAD,
                               normalise=(MEAN, STD, True)).cpu()
        want = (torch.from_numpy(ref)[[2, 1, 0]] - torch.tensor(MEAN).view(3, 1, 1)) / torch.tensor(STD).view(3, 1, 1)
        assert float((fused - want).abs().max()) < 1e-4


def test_inference_detector_on_real_image_matches_oracle(dev, tmp_path):
    """init-free variant of mmdet/apis/inference.py:122-193: model built from the default config with seeded weights,
    image handed over (a) as a file on disk, (b) as an ndarray; oracle = CPU pipeline -> data_preprocess -> predict."""
    import rsprompter_amd as ra
    from PIL import Image
    from oracle import glue
    from oracle import pipeline as op
    from oracle.anchor import AnchorOracle
    from rsprompter_amd import apis
    from rsprompter_amd.config import Config
    from rsprompter_amd.default_configs import rsprompter_anchor
    from rsprompter_amd.synth import synth_state_dict
    bgr = _image()
    path = str(tmp_path / 'crop.png')
    Image.fromarray(np.ascontiguousarray(bgr[:, :, ::-1])).save(path)           # lossless, so both sides see the same pixels
    cfg = Config(dict(
        model=rsprompter_anchor('base', 10),
        test_dataloader=dict(dataset=dict(pipeline=[
            dict(type='LoadImageFromFile', backend_args=None, to_float32=True),
            dict(type='Resize', scale=(1024, 1024), keep_ratio=True),
            dict(type='Pad', size=(1024, 1024), pad_val=dict(img=PAD, masks=0)),
            dict(type='LoadAnnotations', with_bbox=True, with_mask=True),
            dict(type='PackDetInputs', meta_keys=('img_id', 'img_path', 'ori_shape', 'img_shape', 'scale_factor'))]))))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = ra.build_model(cfg)
    oracle = AnchorOracle('base', 10)
    sd = synth_state_dict(oracle, seed=0)
    oracle.load_state_dict(sd)
    model.load_state_dict(sd, strict=True)
    model.cfg = cfg
    model = model.to(dev)
    # ---- oracle side
    inp, meta = op.run_test_pipeline(bgr)
    x = glue.data_preprocess([torch.from_numpy(inp)], MEAN, STD, True, 32)
    m = dict(meta, batch_input_shape=(1024, 1024), img_id=0)
    ref, _ = oracle.predict(x, [m])
    r = ref[0]
    for src in (path, bgr):
        out = apis.inference_detector(model, src)
        pi = out.pred_instances
        assert out.metainfo['ori_shape'] == (160, 256) and out.metainfo['scale_factor'] == (4.0, 4.0)
        assert tuple(pi.masks.shape[1:]) == (160, 256) and pi.labels.shape[0] == r['labels'].shape[0]
        pairs = match_detections(pi.bboxes, pi.scores, pi.labels, r['bboxes'], r['scores'], r['labels'])
        ii = torch.tensor([i for i, _ in pairs]); jj = torch.tensor([j for _, j in pairs])
        mism = float((pi.masks.cpu()[ii] != r['masks'][jj]).float().mean())
        print(f'inference_detector({type(src).__name__}): {pi.labels.shape[0]} dets, {len(pairs)} matched, mask mismatch {mism:.2e}')
        assert mism < 1e-3
    # DetInferencer surface (det_inferencer.py:298-417): predictions as plain dicts with RLE masks
    inf = apis.DetInferencer(model=model)
    res = inf([path, bgr], batch_size=2)
    assert len(res['predictions']) == 2 and res['visualization'] == []
    p0 = res['predictions'][0]
    assert set(p0) == {'labels', 'scores', 'bboxes', 'masks'} and len(p0['masks']) == len(p0['labels'])
    assert p0['masks'][0]['size'] == [160, 256] and isinstance(p0['masks'][0]['counts'], bytes)


def test_init_detector_from_config_file_checkpoints_and_jpeg(dev, tmp_path):
    """SURVEY.md §8 f3 + f2 on the device (mmdet/apis/inference.py:26-193, loader semantics models.py:777-783, 840-851):
    a config FILE (python, `_base_` inheritance) whose SAM sub-modules point at an HF-layout `model.safetensors` through
    init_cfg=Pretrained, a trained-detector mmengine `.pth` on top, `init_detector(config, checkpoint, 'cuda:0')`, and
    `inference_detector` on a JPEG file that both sides decode -- against the oracle loaded with the same tensors."""
    import rsprompter_amd as ra
    from PIL import Image
    from safetensors.torch import save_file
    from oracle import glue
    from oracle import pipeline as op
    from oracle.anchor import AnchorOracle
    from rsprompter_amd import apis
    from rsprompter_amd.default_configs import rsprompter_anchor
    from rsprompter_amd.synth import synth_state_dict
    # ---- the tensors: detector weights (seed 0), but the SAM encoder as "pretrained" HF weights of ANOTHER seed, so the
    # test tells whether init_cfg loaded them (they are then overwritten by the trained checkpoint, like the reference)
    oracle = AnchorOracle('base', 10)
    sd = synth_state_dict(oracle, seed=0)
    oracle.load_state_dict(sd)
    sd_pre = synth_state_dict(oracle, seed=5)
    hf = {k[len('backbone.'):]: v.contiguous() for k, v in sd_pre.items() if k.startswith('backbone.vision_encoder.')}
    hf_dir = tmp_path / 'sam-vit-base'
    hf_dir.mkdir()
    save_file(hf, str(hf_dir / 'model.safetensors'))                       # HF layout: keys start with vision_encoder.
    ckpt = str(tmp_path / 'epoch_1.pth')
    torch.save(dict(meta=dict(epoch=1, dataset_meta=dict(classes=[f'c{i}' for i in range(10)])),
                    state_dict={('module.' + k if i % 2 else k): v for i, (k, v) in enumerate(sd.items())}), ckpt)
    # ---- the config as files: a base file + a child that overrides the pretrained path (mmengine `_base_` semantics)
    model_cfg = rsprompter_anchor('base', 10)
    (tmp_path / 'base_cfg.py').write_text(
        'crop_size = (1024, 1024)\n'
        f'model = {model_cfg!r}\n'
        'test_dataloader = dict(dataset=dict(pipeline=[\n'
        "    dict(type='LoadImageFromFile', backend_args=None, to_float32=True),\n"
        "    dict(type='Resize', scale=crop_size, keep_ratio=True),\n"
        f"    dict(type='Pad', size=crop_size, pad_val=dict(img={PAD!r}, masks=0)),\n"
        "    dict(type='PackDetInputs', meta_keys=('img_id', 'img_path', 'ori_shape', 'img_shape', 'scale_factor'))]))\n")
    (tmp_path / 'child_cfg.py').write_text(
        "_base_ = ['base_cfg.py']\n"
        f"model = dict(backbone=dict(init_cfg=dict(type='Pretrained', checkpoint={str(hf_dir)!r})))\n")
    # ---- (1) init_cfg alone: the encoder must carry the seed-5 tensors, everything else its constructor defaults
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m_pre = apis.init_detector(str(tmp_path / 'child_cfg.py'), None, device=dev)
    got = m_pre.state_dict()
    k0 = 'backbone.vision_encoder.layers.3.attn.qkv.weight'
    assert torch.equal(got[k0].cpu(), sd_pre[k0]) and not torch.equal(got[k0].cpu(), sd[k0])
    # ---- (2) with the trained checkpoint (module.-prefixed keys mixed in): every tensor equals the oracle's
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = apis.init_detector(str(tmp_path / 'child_cfg.py'), ckpt, device=dev)
    got = model.state_dict()
    assert all(torch.equal(got[k].cpu(), v) for k, v in sd.items())
    assert next(model.parameters()).device.type == 'cuda' and not model.training and model.cfg is not None
    # ---- (3) a JPEG on disk, decoded by the product's loader; the oracle gets the pixels of the same file
    bgr = _image()
    jpg = str(tmp_path / 'crop.jpg')
    Image.fromarray(np.ascontiguousarray(bgr[:, :, ::-1])).save(jpg, quality=92)
    with Image.open(jpg) as im:
        dec = np.ascontiguousarray(np.asarray(im.convert('RGB'))[:, :, ::-1])
    assert dec.shape == bgr.shape and int(np.abs(dec.astype(int) - bgr.astype(int)).max()) > 0      # really lossy
    inp, meta = op.run_test_pipeline(dec)
    x = glue.data_preprocess([torch.from_numpy(inp)], MEAN, STD, True, 32)
    ref, _ = oracle.predict(x, [dict(meta, batch_input_shape=(1024, 1024), img_id=0)])
    r = ref[0]
    out = apis.inference_detector(model, jpg)
    pi = out.pred_instances
    assert pi.labels.shape[0] == r['labels'].shape[0] and out.metainfo['img_path'] == jpg
    pairs = match_detections(pi.bboxes, pi.scores, pi.labels, r['bboxes'], r['scores'], r['labels'])
    ii = torch.tensor([i for i, _ in pairs]); jj = torch.tensor([j for _, j in pairs])
    mism = float((pi.masks.cpu()[ii] != r['masks'][jj]).float().mean())
    print(f'init_detector(config file, .pth) + inference_detector(JPEG): {pi.labels.shape[0]} dets, {len(pairs)} matched, '
          f'mask mismatch {mism:.2e}')
    assert mism < 1e-3
