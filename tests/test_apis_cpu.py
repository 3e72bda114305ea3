"""CPU: caller APIs and checkpoint formats -- host logic only; the HIP resize kernel is checked
against the same oracle in tests/test_gpu_apis.py."""
import os
import sys
import warnings

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'


def test_cv2_linear_restatement_matches_torch_interpolate():
    """the oracle's cv2.resize(INTER_LINEAR, float32) restatement vs torch's own bilinear (align_corners=False): the
    same sampling rule implemented independently -- up- and down-scaling, odd sizes."""
    import torch.nn.functional as F
    from oracle import pipeline as op
    g = np.random.default_rng(3)
    for (h, w), (nh, nw) in (((288, 512), (576, 1024)), ((37, 53), (111, 160)), ((600, 400), (300, 200)),
                             ((50, 70), (33, 91))):
        img = g.uniform(0, 255, size=(h, w, 3)).astype(np.float32)
        got = op.cv2_resize_linear_f32(img, nw, nh)
        ref = F.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None], size=(nh, nw), mode='bilinear',
                            align_corners=False)[0].permute(1, 2, 0).numpy()
        assert got.shape == ref.shape and float(np.abs(got - ref).max()) < 5e-3     # torch keeps the scale in fp32, cv2 in double


def test_rescale_size_and_metainfo_follow_mmcv():
    from oracle import pipeline as op
    from rsprompter_amd.apis import rescale_size
    for wh in ((512, 288), (640, 427), (1024, 1024), (3000, 2000), (333, 1000)):
        assert rescale_size(wh, (1024, 1024))[0] == op.rescale_size(wh, (1024, 1024))[0]
    assert rescale_size((512, 288), (1024, 1024))[0] == (1024, 576)
    assert rescale_size((640, 427), (1024, 1024))[0] == (1024, 683)            # 427 * 1.6 = 683.2 -> int(683.7)


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference checkout not available')
def test_test_pipeline_from_reference_config_on_reference_image(monkeypatch):
    """the pipeline cfg of configs/rsprompter/rsprompter_anchor-nwpu.py applied to the reference's tests/data/color.jpg
    (512 x 288 -> resized to 1024 x 576 -> padded to 1024 x 1024 with the BGR mean)."""
    import torch_ops_mock as mock
    import rsprompter_amd as ra
    import rsprompter_amd.apis as apis
    from oracle import pipeline as op
    monkeypatch.setattr(apis, 'ops', mock)
    cfg = ra.Config.fromfile(os.path.join(REF, 'configs/rsprompter/rsprompter_anchor-nwpu.py'))
    pipe = apis.TestPipeline(apis.get_test_pipeline_cfg(cfg), device='cpu')
    assert pipe.scale == (1024, 1024) and pipe.keep_ratio and pipe.pad_size == (1024, 1024)
    assert np.allclose(pipe.pad_val, (0.406 * 255, 0.456 * 255, 0.485 * 255))
    path = os.path.join(REF, 'tests/data/color.jpg')
    out = pipe(dict(img_path=path, img_id=7))
    meta = out['data_samples'].metainfo
    assert tuple(out['inputs'].shape) == (3, 1024, 1024)
    assert meta['ori_shape'] == (288, 512) and meta['img_shape'] == (1024, 1024) and meta['img_id'] == 7
    assert meta['scale_factor'] == (2.0, 2.0) and meta['img_path'] == path
    bgr = apis.TestPipeline._decode(path)
    ref, rmeta = op.run_test_pipeline(bgr)
    assert rmeta['scale_factor'] == meta['scale_factor']
    assert float(np.abs(out['inputs'].numpy() - ref).max()) == 0.0
    # padding region carries the per-channel pad value, the image region does not
    assert np.allclose(out['inputs'][:, 600:, :].numpy().reshape(3, -1).T, pipe.pad_val)


def test_checkpoint_formats_round_trip(tmp_path):
    """HF layout (`vision_encoder.` prefix, flat .bin and .safetensors, sharded index), mmengine .pth wrapper,
    DeepSpeed zero_to_fp32 (`module.` prefix) and the ConvModule norm-name alias (`ln` <-> `norm_layer`)."""
    from rsprompter_amd import checkpoint as ck
    from rsprompter_amd.necks import RSSimpleFPN
    from rsprompter_amd.sam_encoder import RSSamVisionEncoder
    from rsprompter_amd.synth import synth_state_dict
    from safetensors.torch import save_file
    import json
    enc = RSSamVisionEncoder('sam_vit_base', extra_config=dict(output_hidden_states=True))
    want = synth_state_dict(enc.vision_encoder, seed=4)
    # --- HF single files: keys `vision_encoder.<...>` next to unrelated ones (prompt encoder, mask decoder)
    hf = {'vision_encoder.' + k: v for k, v in want.items()}
    hf['mask_decoder.iou_token.weight'] = torch.zeros(1, 256)
    torch.save(hf, tmp_path / 'pytorch_model.bin')
    save_file({k: v.contiguous() for k, v in hf.items()}, str(tmp_path / 'model.safetensors'))
    rk = [(r'^module\.', ''), (r'^vision_encoder\.', '')]
    for src in ('pytorch_model.bin', 'model.safetensors'):
        m = RSSamVisionEncoder('sam_vit_base').vision_encoder
        assert ck.load_checkpoint_into(m, str(tmp_path / src), revise_keys=rk)
        assert all(torch.equal(m.state_dict()[k], v) for k, v in want.items())
        assert m._last_load_report['missing'] == []
    # --- sharded safetensors + index json, addressed through the directory
    shard_dir = tmp_path / 'sharded'
    shard_dir.mkdir()
    keys = sorted(hf)
    half = len(keys) // 2
    parts = {'model-00001-of-00002.safetensors': keys[:half], 'model-00002-of-00002.safetensors': keys[half:]}
    for fn, ks in parts.items():
        save_file({k: hf[k].contiguous() for k in ks}, str(shard_dir / fn))
    with open(shard_dir / 'model.safetensors.index.json', 'w') as f:
        json.dump(dict(metadata={}, weight_map={k: fn for fn, ks in parts.items() for k in ks}), f)
    m = RSSamVisionEncoder('sam_vit_base').vision_encoder
    assert ck.load_checkpoint_into(m, str(shard_dir), revise_keys=rk)
    assert all(torch.equal(m.state_dict()[k], v) for k, v in want.items())
    # --- the constructor path: init_cfg=dict(type='Pretrained', checkpoint=<HF file>) (models.py:777-783)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        e2 = RSSamVisionEncoder('sam_vit_base', init_cfg=dict(type='Pretrained', checkpoint=str(tmp_path / 'model.safetensors')))
    assert torch.equal(e2.vision_encoder.state_dict()['layers.3.attn.qkv.weight'], want['layers.3.attn.qkv.weight'])
    # --- mmengine .pth (dict(meta, state_dict)) and zero_to_fp32 output (flat, `module.` prefix) of a sub-module
    fpn = RSSimpleFPN(256, [64, 128, 256, 256], 256, 5, norm_cfg=dict(type='LN2d', requires_grad=True))
    w_fpn = synth_state_dict(fpn, seed=5)
    ck.save_checkpoint(torch.nn.ModuleDict(), str(tmp_path / 'empty.pth'))      # writer works on an empty module
    torch.save(dict(meta=dict(epoch=3, dataset_meta=dict(classes=('a',))), state_dict=w_fpn, optimizer=dict(lr=1e-4)),
               tmp_path / 'epoch_3.pth')
    torch.save({'module.' + k: v for k, v in w_fpn.items()}, tmp_path / 'zero_fp32.bin')
    for src in ('epoch_3.pth', 'zero_fp32.bin'):
        f2 = RSSimpleFPN(256, [64, 128, 256, 256], 256, 5, norm_cfg=dict(type='LN2d', requires_grad=True))
        assert ck.load_checkpoint_into(f2, str(tmp_path / src), strict=True)
        assert all(torch.equal(f2.state_dict()[k], v) for k, v in w_fpn.items())
    # --- norm-name alias: a checkpoint that spells the ConvModule norm `ln` instead of `norm_layer`
    alias = {k.replace('.norm_layer.', '.ln.'): v for k, v in w_fpn.items()}
    assert any('.ln.' in k for k in alias)
    torch.save(alias, tmp_path / 'alias.pth')
    f3 = RSSimpleFPN(256, [64, 128, 256, 256], 256, 5, norm_cfg=dict(type='LN2d', requires_grad=True))
    assert ck.load_checkpoint_into(f3, str(tmp_path / 'alias.pth'), strict=True)
    assert all(torch.equal(f3.state_dict()[k], v) for k, v in w_fpn.items())
    # --- writer/reader symmetry
    for name in ('rt.safetensors', 'rt.pth', 'rt.bin'):
        ck.save_checkpoint(fpn.__class__(256, [64, 128, 256, 256], 256, 5) if False else f3, str(tmp_path / name))
        back = ck.read_state_dict(str(tmp_path / name))
        assert set(back) == set(w_fpn) and all(torch.equal(back[k], w_fpn[k]) for k in back)
    # strict load reports what does not belong
    torch.save(dict(w_fpn, **{'stranger.weight': torch.zeros(1)}), tmp_path / 'extra.pth')
    with pytest.raises(RuntimeError):
        ck.load_checkpoint_into(f3, str(tmp_path / 'extra.pth'), strict=True)


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference checkout not available')
def test_init_detector_and_inference_detector_host_logic(monkeypatch, tmp_path):
    """init_detector(config file) + a trained-checkpoint load + inference_detector on an ndarray and on a file path,
    with every native op replaced by the torch stand-ins (tiny 'image': the pipeline still pads it to 1024 x 1024)."""
    import torch_ops_mock as mock
    import rsprompter_amd.anchor_heads as ah
    import rsprompter_amd.apis as apis
    import rsprompter_amd.detectors as det
    import rsprompter_amd.necks as necks
    import rsprompter_amd.sam_decoder as sd
    import rsprompter_amd.sam_encoder as se
    from rsprompter_amd.checkpoint import save_checkpoint
    from rsprompter_amd.synth import synth_state_dict
    for m in (ah, necks, sd, se, det, apis):
        monkeypatch.setattr(m, 'ops', mock)
    cfg_path = os.path.join(REF, 'configs/rsprompter/rsprompter_anchor-nwpu.py')
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        donor = apis.init_detector(cfg_path, device='cpu')
        donor.load_state_dict(synth_state_dict(donor, seed=0), strict=True)
        save_checkpoint(donor, str(tmp_path / 'trained.pth'))
        model = apis.init_detector(cfg_path, str(tmp_path / 'trained.pth'), device='cpu')
    assert model.cfg.model.type == 'RSPrompterAnchor'
    assert torch.equal(model.state_dict()['rpn_head.rpn_conv.weight'], donor.state_dict()['rpn_head.rpn_conv.weight'])
    g = np.random.default_rng(0)
    img = g.integers(0, 256, size=(96, 128, 3), dtype=np.uint8)
    r = apis.inference_detector(model, img)
    assert tuple(r.pred_instances.masks.shape[1:]) == (96, 128)        # masks come back at the image's ori_shape
    assert r.metainfo['scale_factor'] == (8.0, 8.0) and r.pred_instances.bboxes.shape[1] == 4
    assert float(r.pred_instances.bboxes.max()) <= 128.0 + 1e-3         # boxes rescaled to the original image
    rs = apis.inference_detector(model, [img])
    assert isinstance(rs, list) and len(rs) == 1


class _Evil:
    """a class the restricted loader must keep refusing"""

    def __reduce__(self):
        return (print, ('code ran while unpickling',))


def test_mmengine_style_checkpoint_with_numpy_meta_loads_without_full_unpickling(tmp_path):
    """Reference-trained mmengine `.pth` files keep numpy scalars / arrays and OrderedDicts next to the state_dict
    (meta, message_hub: mmengine/runner/checkpoint.py save_checkpoint).  The restricted loader alone refuses those; the
    allow-list of harmless reconstructors lets the upstream format load out of the box, and a file that pickles anything
    else is still refused unless the caller opts into the full unpickler."""
    import collections
    import numpy as np
    import pytest
    from rsprompter_amd import checkpoint as ck
    sd = collections.OrderedDict(a=torch.randn(3, 4), b=torch.arange(5))
    good = dict(meta=dict(epoch=12, iter=np.int64(3456), seed=np.array([1, 2, 3]), lr=np.float64(1e-4),
                          time='2023-10-01', cfg='model = dict(...)'),
                message_hub=dict(log_scalars=collections.OrderedDict(loss=np.float32(0.25)), runtime_info=dict(iter=7)),
                state_dict=sd)
    torch.save(good, str(tmp_path / 'mm.pth'))
    got = ck.read_state_dict(str(tmp_path / 'mm.pth'))
    assert set(got) == {'a', 'b'} and torch.equal(got['a'], sd['a'])
    torch.save(dict(meta=dict(hook=_Evil()), state_dict=sd), str(tmp_path / 'evil.pth'))
    with pytest.raises(RuntimeError, match='refused'):
        ck.read_state_dict(str(tmp_path / 'evil.pth'))
