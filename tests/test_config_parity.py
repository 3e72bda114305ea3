"""CPU: the reference's own config files build, unchanged, through our registry/config loader
(only where /root/reference is present, i.e. in the build container)."""
import os
import warnings

import pytest

REF = '/root/reference/configs/rsprompter'


def _norm(x):
    if isinstance(x, dict):
        return {k: _norm(v) for k, v in x.items()}
    if isinstance(x, (list, tuple, range)):
        return [_norm(v) for v in x]
    return x


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference checkout not available')
@pytest.mark.parametrize('fname,arch,nc,ps', [('rsprompter_anchor-nwpu.py', 'base', 10, (70, 5)),
                                              ('rsprompter_anchor-ssdd.py', 'base', 1, (30, 5)),
                                              ('rsprompter_anchor-whu.py', 'base', 1, (100, 5))])
def test_reference_anchor_configs_build(fname, arch, nc, ps):
    import rsprompter_amd as ra
    from rsprompter_amd.default_configs import rsprompter_anchor
    cfg = ra.Config.fromfile(os.path.join(REF, fname))
    ours = rsprompter_anchor(arch, nc, ps)
    assert _norm(cfg.model) == _norm(ours)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m = ra.build_model(cfg)
    assert type(m).__name__ == 'RSPrompterAnchor'
    assert m.roi_head.bbox_head.num_classes == nc
    # parameters follow the reference state_dict layout
    keys = set(m.state_dict())
    for k in ['backbone.vision_encoder.layers.0.attn.qkv.weight',
              'shared_image_embedding.shared_image_embedding.positional_embedding',
              'neck.feature_aggregator.downconvs.0.0.weight', 'neck.feature_spliter.fpn1.0.weight',
              'neck.feature_spliter.lateral_convs.0.norm_layer.weight', 'rpn_head.rpn_conv.weight',
              'roi_head.bbox_head.shared_fcs.0.weight', 'roi_head.mask_head.no_mask_embed.weight',
              'roi_head.mask_head.mask_decoder.mask_decoder.transformer.layers.0.self_attn.q_proj.weight',
              'roi_head.mask_head.point_emb.8.weight']:
        assert k in keys, k


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference checkout not available')
def test_reference_peft512_config_builds_with_peft_key_layout():
    import rsprompter_amd as ra
    from rsprompter_amd.default_configs import rsprompter_anchor_peft512
    cfg = ra.Config.fromfile(os.path.join(REF, 'rsprompter_anchor-nwpu-peft-512.py'))
    assert _norm(cfg.model) == _norm(rsprompter_anchor_peft512('base', 10, (70, 5)))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m = ra.build_model(cfg)
    keys = set(m.state_dict())
    p = 'backbone.vision_encoder.base_model.model.'          # peft 0.8.2 wrapper layout
    for k in [p + 'layers.0.attn.qkv.base_layer.weight', p + 'layers.0.attn.qkv.lora_A.default.weight',
              p + 'layers.3.attn.qkv.lora_B.default.weight', p + 'layers.0.ffn.layers.0.0.weight',
              p + 'channel_reduction.3.bias', p + 'pos_embed', 'neck.feature_aggregator.channel_fusion.4.weight']:
        assert k in keys, k
    assert tuple(m.state_dict()[p + 'pos_embed'].shape) == (1, 32, 32, 768)
    assert tuple(m.state_dict()[p + 'layers.2.attn.rel_pos_h'].shape) == (63, 64)      # global layer at 512 px
    # load-time interpolation of a 1024-px checkpoint (vit_sam.py:612-662)
    import torch
    sd = {p + 'pos_embed': torch.randn(1, 64, 64, 768), p + 'layers.2.attn.rel_pos_h': torch.randn(127, 64)}
    m.backbone.load_state_dict({k[len('backbone.'):]: v for k, v in sd.items()}, strict=False)
    assert tuple(m.backbone.vision_encoder.pos_embed.shape) == (1, 32, 32, 768)


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference checkout not available')
@pytest.mark.parametrize('fname,nc,ps,mpi', [('rsprompter_query-whu.py', 1, (100, 5), 100),
                                             ('rsprompter_query-nwpu.py', 10, (70, 5), 70),
                                             ('rsprompter_query-ssdd.py', 1, (30, 5), 30)])
def test_reference_query_configs_build(fname, nc, ps, mpi):
    import rsprompter_amd as ra
    from rsprompter_amd.default_configs import rsprompter_query
    cfg = ra.Config.fromfile(os.path.join(REF, fname))
    assert _norm(cfg.model) == _norm(rsprompter_query('base', nc, ps, max_per_image=mpi))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m = ra.build_model(cfg)
    assert type(m).__name__ == 'RSPrompterQuery' and m.panoptic_head.num_queries == ps[0]
    keys = set(m.state_dict())
    for k in ['panoptic_head.pixel_decoder.encoder.layers.0.self_attn.sampling_offsets.weight',
              'panoptic_head.pixel_decoder.input_convs.0.gn.weight', 'panoptic_head.pixel_decoder.mask_feature.bias',
              'panoptic_head.transformer_decoder.layers.5.cross_attn.attn.in_proj_weight',
              'panoptic_head.transformer_decoder.post_norm.weight', 'panoptic_head.query_feat.weight',
              'panoptic_head.cls_embed.2.weight', 'panoptic_head.mask_embed.4.bias', 'panoptic_head.point_emb.4.weight',
              'panoptic_head.sam_mask_embed.conv2.weight', 'panoptic_head.sam_mask_embed.layer_norm2.bias',
              'panoptic_head.mask_decoder.mask_decoder.iou_token.weight']:
        assert k in keys, k


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference checkout not available')
def test_config_loader_base_and_delete():
    import rsprompter_amd as ra
    cfg = ra.Config.fromfile(os.path.join(REF, 'rsprompter_anchor-nwpu-peft-512.py'))
    assert cfg.model.backbone.type == 'MMPretrainSamVisionEncoder'      # `_delete_=True` replaced the base dict
    assert 'extra_config' not in cfg.model.backbone
    assert cfg.model.neck.feature_aggregator.type == 'PseudoFeatureAggregator'
    assert cfg.model.rpn_head.anchor_generator.strides == [4, 8, 16, 32, 64]   # inherited from _base_


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference checkout not available')
@pytest.mark.parametrize('fname,nc', [('samseg-maskrcnn-nwpu.py', 10), ('samseg-maskrcnn-ssdd.py', 1), ('samseg-maskrcnn-whu.py', 1)])
def test_reference_samseg_maskrcnn_configs_build(fname, nc):
    """the SAMSegMaskRCNN sibling model builds from the reference's config files unchanged."""
    import rsprompter_amd as ra
    from rsprompter_amd.default_configs import samseg_maskrcnn
    cfg = ra.Config.fromfile(os.path.join(REF, fname))
    assert _norm(cfg.model) == _norm(samseg_maskrcnn('base', nc))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m = ra.build_model(cfg)
    assert type(m).__name__ == 'SAMSegMaskRCNN' and type(m.roi_head).__name__ == 'StandardRoIHead'
    assert m.rpn_head.num_base_priors == 3 and m.roi_head.mask_head.num_classes == nc
    keys = set(m.state_dict())
    for k in ['roi_head.mask_head.convs.3.conv.weight', 'roi_head.mask_head.upsample.weight',
              'roi_head.mask_head.conv_logits.bias', 'roi_head.bbox_head.fc_reg.weight', 'rpn_head.rpn_cls.weight']:
        assert k in keys, k
    assert not any('mask_decoder' in k or 'shared_image_embedding' in k for k in keys)


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference checkout not available')
@pytest.mark.parametrize('fname,nc,nq', [('samseg-mask2former-nwpu.py', 10, 70), ('samseg-mask2former-ssdd.py', 1, None),
                                         ('samseg-mask2former-whu.py', 1, None)])
def test_reference_samseg_mask2former_configs_build(fname, nc, nq):
    """the SAMSegMask2Former sibling model builds from the reference's config files unchanged."""
    import rsprompter_amd as ra
    from rsprompter_amd.default_configs import samseg_mask2former
    cfg = ra.Config.fromfile(os.path.join(REF, fname))
    nq = cfg.model.panoptic_head.num_queries
    assert _norm(cfg.model) == _norm(samseg_mask2former('base', nc, nq))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m = ra.build_model(cfg)
    assert type(m).__name__ == 'SAMSegMask2Former' and type(m.panoptic_head).__name__ == 'Mask2FormerHead'
    assert m.panoptic_head.num_transformer_decoder_layers == 9 and m.panoptic_head.pixel_decoder.feat == 256


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference checkout not available')
@pytest.mark.parametrize('fname,nc', [('samdet-nwpu.py', 10), ('samdet-ssdd.py', 1), ('samdet-whu.py', 1)])
def test_reference_samdet_configs_build(fname, nc):
    """SAMDet (Faster R-CNN R50-FPN + HF SamModel) builds from the reference's config files unchanged."""
    import rsprompter_amd as ra
    from rsprompter_amd.default_configs import samdet
    cfg = ra.Config.fromfile(os.path.join(REF, fname))
    assert _norm(cfg.model) == _norm(samdet('base', nc))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m = ra.build_model(cfg)
    assert type(m).__name__ == 'SAMDet' and type(m.detector).__name__ == 'FasterRCNN'
    assert type(m.detector.backbone).__name__ == 'ResNet' and type(m.detector.neck).__name__ == 'FPN'
    assert m.detector.rpn_head.num_base_priors == 3 and m.detector.roi_head.mask_head is None
    assert m.detector.roi_head.bbox_head.num_classes == nc and m.test_cfg is None
    keys = set(m.state_dict())
    for k in ['detector.backbone.layer4.2.bn3.running_var', 'detector.backbone.layer1.0.downsample.0.weight',
              'detector.neck.fpn_convs.3.conv.bias', 'detector.roi_head.bbox_head.fc_reg.weight',
              'segmentor.sam_model.prompt_encoder.point_embed.3.weight',
              'segmentor.sam_model.prompt_encoder.shared_embedding.positional_embedding',
              'segmentor.sam_model.shared_image_embedding.positional_embedding',
              'segmentor.sam_model.vision_encoder.layers.11.attn.rel_pos_w',
              'segmentor.sam_model.mask_decoder.output_hypernetworks_mlps.3.proj_out.bias']:
        assert k in keys, k
