"""Programmatic `model = dict(...)` trees equal to the ones the reference's config files produce
(configs/rsprompter/_base_/rsprompter_anchor.py:57-200 merged with rsprompter_anchor-<dataset>.py).
Used by bench.py / tests on machines where /root/reference is absent; tests/test_config_parity.py
checks them field by field against the real config files when those are present.
"""

import copy
SELECT_LAYERS = {'base': range(1, 13, 2), 'large': range(1, 25, 2), 'huge': range(1, 33, 2)}
MEAN = [0.485 * 255, 0.456 * 255, 0.406 * 255]
STD = [0.229 * 255, 0.224 * 255, 0.225 * 255]


def rsprompter_anchor(arch='base', num_classes=10, prompt_shape=(70, 5), pretrain_name=None, ckpt=None):
    name = pretrain_name or f'work_dirs/sam_cache/sam_vit_{arch}'
    init = dict(type='Pretrained', checkpoint=ckpt or f'{name}/pytorch_model.bin')
    crop = (1024, 1024)
    pre = dict(type='DetDataPreprocessor', mean=MEAN, std=STD, bgr_to_rgb=True, pad_mask=True,
               pad_size_divisor=32,
               batch_augments=[dict(type='BatchFixedSizePad', size=crop, img_pad_value=0, pad_mask=True,
                                    mask_pad_value=0, pad_seg=False)])
    return dict(
        type='RSPrompterAnchor', data_preprocessor=pre, decoder_freeze=False,
        shared_image_embedding=dict(type='RSSamPositionalEmbedding', hf_pretrain_name=name, init_cfg=init),
        backbone=dict(type='RSSamVisionEncoder', hf_pretrain_name=name,
                      extra_config=dict(output_hidden_states=True), init_cfg=init),
        neck=dict(type='RSFPN',
                  feature_aggregator=dict(type='RSFeatureAggregator', in_channels=name, out_channels=256,
                                          hidden_channels=32, select_layers=SELECT_LAYERS[arch]),
                  feature_spliter=dict(type='RSSimpleFPN', backbone_channel=256, in_channels=[64, 128, 256, 256],
                                       out_channels=256, num_outs=5,
                                       norm_cfg=dict(type='LN2d', requires_grad=True))),
        rpn_head=dict(type='RPNHead', in_channels=256, feat_channels=256,
                      anchor_generator=dict(type='AnchorGenerator', scales=[4, 8], ratios=[0.5, 1.0, 2.0],
                                            strides=[4, 8, 16, 32, 64]),
                      bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[.0, .0, .0, .0],
                                      target_stds=[1.0, 1.0, 1.0, 1.0]),
                      loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                      loss_bbox=dict(type='SmoothL1Loss', loss_weight=1.0)),
        roi_head=dict(
            type='RSPrompterAnchorRoIPromptHead', with_extra_pe=True,
            bbox_roi_extractor=dict(type='SingleRoIExtractor',
                                    roi_layer=dict(type='RoIAlign', output_size=7, sampling_ratio=0),
                                    out_channels=256, featmap_strides=[4, 8, 16, 32]),
            bbox_head=dict(type='Shared2FCBBoxHead', in_channels=256, fc_out_channels=1024, roi_feat_size=7,
                           num_classes=num_classes,
                           bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[0., 0., 0., 0.],
                                           target_stds=[0.1, 0.1, 0.2, 0.2]),
                           reg_class_agnostic=False,
                           loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
                           loss_bbox=dict(type='SmoothL1Loss', loss_weight=1.0)),
            mask_roi_extractor=dict(type='SingleRoIExtractor',
                                    roi_layer=dict(type='RoIAlign', output_size=14, sampling_ratio=0),
                                    out_channels=256, featmap_strides=[4, 8, 16, 32]),
            mask_head=dict(type='RSPrompterAnchorMaskHead',
                           mask_decoder=dict(type='RSSamMaskDecoder', hf_pretrain_name=name, init_cfg=init),
                           in_channels=256, roi_feat_size=14, per_pointset_point=prompt_shape[1],
                           with_sincos=True, multimask_output=False, class_agnostic=True,
                           loss_mask=dict(type='CrossEntropyLoss', use_mask=True, loss_weight=1.0))),
        train_cfg=dict(
            rpn=dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3,
                                   match_low_quality=True, ignore_iof_thr=-1),
                     sampler=dict(type='RandomSampler', num=256, pos_fraction=0.5, neg_pos_ub=-1,
                                  add_gt_as_proposals=False),
                     allowed_border=-1, pos_weight=-1, debug=False),
            rpn_proposal=dict(nms_pre=2000, max_per_img=1000, nms=dict(type='nms', iou_threshold=0.7),
                              min_bbox_size=0),
            rcnn=dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5,
                                    match_low_quality=True, ignore_iof_thr=-1),
                      sampler=dict(type='RandomSampler', num=256, pos_fraction=0.25, neg_pos_ub=-1,
                                   add_gt_as_proposals=True),
                      mask_size=crop, pos_weight=-1, debug=False)),
        test_cfg=dict(rpn=dict(nms_pre=1000, max_per_img=1000, nms=dict(type='nms', iou_threshold=0.7),
                               min_bbox_size=0),
                      rcnn=dict(score_thr=0.05, nms=dict(type='nms', iou_threshold=0.5), max_per_img=100,
                                mask_thr_binary=0.5)))


def rsprompter_anchor_peft512(arch='base', num_classes=10, prompt_shape=(60, 5), pretrain_name=None, ckpt=None):
    """configs/rsprompter/rsprompter_anchor-nwpu-peft-512.py: ViTSAM at 512 px + LoRA(qkv) + PseudoFeatureAggregator."""
    m = rsprompter_anchor(arch, num_classes, prompt_shape, pretrain_name, ckpt)
    name = pretrain_name or f'work_dirs/sam_cache/sam_vit_{arch}'
    init = dict(type='Pretrained', checkpoint=ckpt or f'{name}/pytorch_model.bin')
    crop = (512, 512)
    m['data_preprocessor']['batch_augments'][0]['size'] = crop
    m['backbone'] = dict(type='MMPretrainSamVisionEncoder', hf_pretrain_name=name, img_size=crop[0],
                         init_cfg=init,
                         peft_config=dict(peft_type='LORA', r=16, target_modules=['qkv'], lora_alpha=32,
                                          lora_dropout=0.05, bias='none'))
    m['neck']['feature_aggregator'] = dict(type='PseudoFeatureAggregator', in_channels=256, hidden_channels=512,
                                           out_channels=256)
    m['train_cfg']['rcnn']['mask_size'] = crop
    return m


def rsprompter_query(arch='base', num_classes=1, prompt_shape=(100, 5), pretrain_name=None, ckpt=None,
                     max_per_image=None):
    """configs/rsprompter/_base_/rsprompter_query.py:58-202 merged with rsprompter_query-<dataset>.py."""
    name = pretrain_name or f'work_dirs/sam_cache/sam_vit_{arch}'
    init = dict(type='Pretrained', checkpoint=ckpt or f'{name}/pytorch_model.bin')
    a = rsprompter_anchor(arch, num_classes, prompt_shape, pretrain_name, ckpt)
    return dict(
        type='RSPrompterQuery', data_preprocessor=a['data_preprocessor'], decoder_freeze=False,
        shared_image_embedding=a['shared_image_embedding'], backbone=a['backbone'], neck=a['neck'],
        panoptic_head=dict(
            type='RSMask2FormerHead', decoder_plus=True,
            mask_decoder=dict(type='RSSamMaskDecoder', hf_pretrain_name=name, init_cfg=init),
            per_pointset_point=prompt_shape[1], with_sincos=True, multimask_output=False,
            in_channels=[256, 256, 256, 256, 256], feat_channels=128, out_channels=256,
            num_things_classes=num_classes, num_stuff_classes=0, num_queries=prompt_shape[0],
            num_transformer_feat_level=3,
            pixel_decoder=dict(
                type='MSDeformAttnPixelDecoder', strides=[4, 8, 16, 32, 64], num_outs=3,
                norm_cfg=dict(type='GN', num_groups=32), act_cfg=dict(type='ReLU'),
                encoder=dict(num_layers=3, layer_cfg=dict(
                    self_attn_cfg=dict(embed_dims=128, num_heads=8, num_levels=3, num_points=4, dropout=0.0,
                                       batch_first=True),
                    ffn_cfg=dict(embed_dims=128, feedforward_channels=512, num_fcs=2, ffn_drop=0.0,
                                 act_cfg=dict(type='ReLU', inplace=True)))),
                positional_encoding=dict(num_feats=64, normalize=True)),
            enforce_decoder_input_project=False, positional_encoding=dict(num_feats=64, normalize=True),
            transformer_decoder=dict(
                return_intermediate=True, num_layers=6,
                layer_cfg=dict(
                    self_attn_cfg=dict(embed_dims=128, num_heads=8, dropout=0.0, batch_first=True),
                    cross_attn_cfg=dict(embed_dims=128, num_heads=8, dropout=0.0, batch_first=True),
                    ffn_cfg=dict(embed_dims=128, feedforward_channels=512, num_fcs=2, ffn_drop=0.0,
                                 act_cfg=dict(type='ReLU', inplace=True))),
                init_cfg=None),
            loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=2.0, reduction='mean',
                          class_weight=[1.0] * num_classes + [0.1]),
            loss_mask=dict(type='CrossEntropyLoss', use_sigmoid=True, reduction='mean', loss_weight=5.0),
            loss_dice=dict(type='DiceLoss', use_sigmoid=True, activate=True, reduction='mean', naive_dice=True,
                           eps=1.0, loss_weight=5.0)),
        panoptic_fusion_head=dict(type='RSMaskFormerFusionHead', num_things_classes=num_classes,
                                  num_stuff_classes=0, loss_panoptic=None, init_cfg=None),
        train_cfg=dict(
            num_points=12544, oversample_ratio=3.0, importance_sample_ratio=0.75,
            assigner=dict(type='HungarianAssigner', match_costs=[
                dict(type='ClassificationCost', weight=2.0),
                dict(type='CrossEntropyLossCost', weight=5.0, use_sigmoid=True),
                dict(type='DiceCost', weight=5.0, pred_act=True, eps=1.0)]),
            sampler=dict(type='MaskPseudoSampler')),
        test_cfg=dict(panoptic_on=False, semantic_on=False, instance_on=True,
                      max_per_image=prompt_shape[0] if max_per_image is None else max_per_image, iou_thr=0.8,
                      filter_low_score=True))


LORA_QKV = dict(peft_type='LORA', r=16, target_modules=['qkv'], lora_alpha=32, lora_dropout=0.05, bias='none')


def rsprompter_query_lora(arch='huge', num_classes=1, prompt_shape=(100, 5), pretrain_name=None, ckpt=None,
                          max_per_image=None):
    """BASELINE.json configs[4]: the query tree with LoRA(qkv, r16, alpha32) on the 1024-px HF encoder
    (RSSamVisionEncoder's `peft_config`, models.py:776-797)."""
    m = rsprompter_query(arch, num_classes, prompt_shape, pretrain_name, ckpt, max_per_image)
    m['backbone'] = dict(m['backbone'], peft_config=dict(LORA_QKV))
    return m


def rsprompter_query_peft512(arch='base', num_classes=10, prompt_shape=(70, 5), pretrain_name=None, ckpt=None):
    """configs/rsprompter/rsprompter_query-nwpu-peft-512.py: ViTSAM at 512 px + LoRA(qkv) + PseudoFeatureAggregator."""
    m = rsprompter_query(arch, num_classes, prompt_shape, pretrain_name, ckpt)
    name = pretrain_name or f'work_dirs/sam_cache/sam_vit_{arch}'
    init = dict(type='Pretrained', checkpoint=ckpt or f'{name}/pytorch_model.bin')
    crop = (512, 512)
    m['data_preprocessor'] = dict(m['data_preprocessor'])
    m['data_preprocessor']['batch_augments'] = [dict(m['data_preprocessor']['batch_augments'][0], size=crop)]
    m['backbone'] = dict(type='MMPretrainSamVisionEncoder', hf_pretrain_name=name, img_size=crop[0], init_cfg=init,
                         peft_config=dict(LORA_QKV))
    m['neck'] = dict(m['neck'], feature_aggregator=dict(type='PseudoFeatureAggregator', in_channels=256,
                                                        hidden_channels=512, out_channels=256))
    return m


def samseg_maskrcnn(arch='huge', num_classes=10, pretrain_name=None, ckpt=None):
    """configs/rsprompter/_base_/samseg-maskrcnn.py:57-170 merged with samseg-maskrcnn-<dataset>.py."""
    a = rsprompter_anchor(arch, num_classes, (100, 5), pretrain_name, ckpt)
    rpn = copy.deepcopy(a['rpn_head'])
    rpn['anchor_generator'] = dict(type='AnchorGenerator', scales=[8], ratios=[0.5, 1.0, 2.0], strides=[4, 8, 16, 32, 64])
    roi = a['roi_head']
    train_cfg = copy.deepcopy(a['train_cfg'])
    train_cfg['rcnn']['sampler']['num'] = 512
    train_cfg['rcnn']['mask_size'] = 28
    return dict(
        type='SAMSegMaskRCNN', data_preprocessor=a['data_preprocessor'], backbone=a['backbone'], neck=a['neck'], rpn_head=rpn,
        roi_head=dict(type='StandardRoIHead', bbox_roi_extractor=roi['bbox_roi_extractor'], bbox_head=roi['bbox_head'],
                      mask_roi_extractor=roi['mask_roi_extractor'],
                      mask_head=dict(type='FCNMaskHead', num_convs=4, in_channels=256, conv_out_channels=256,
                                     num_classes=num_classes,
                                     loss_mask=dict(type='CrossEntropyLoss', use_mask=True, loss_weight=1.0))),
        train_cfg=train_cfg, test_cfg=a['test_cfg'])


def samseg_mask2former(arch='base', num_classes=10, num_queries=70, pretrain_name=None, ckpt=None):
    """configs/rsprompter/_base_/samseg-mask2former.py:58-178 merged with samseg-mask2former-<dataset>.py:
    the query tree with the STANDARD Mask2FormerHead (feat 256, 9 decoder layers) and no SAM prompt / mask decoder."""
    q = rsprompter_query(arch, num_classes, (num_queries, 5), pretrain_name, ckpt)
    ph = q['panoptic_head']

    def widen(layer_cfg, ffn):
        lc = copy.deepcopy(layer_cfg)
        for k in ('self_attn_cfg', 'cross_attn_cfg'):
            if k in lc:
                lc[k]['embed_dims'] = 256
        lc['ffn_cfg'].update(embed_dims=256, feedforward_channels=ffn)
        return lc
    pd = copy.deepcopy(ph['pixel_decoder'])
    pd['encoder']['layer_cfg'] = widen(pd['encoder']['layer_cfg'], 1024)
    pd['positional_encoding'] = dict(num_feats=128, normalize=True)
    td = copy.deepcopy(ph['transformer_decoder'])
    td.update(num_layers=9, layer_cfg=widen(td['layer_cfg'], 2048))
    head = dict(type='Mask2FormerHead', in_channels=[256, 256, 256, 256, 256], feat_channels=256, out_channels=256,
                num_things_classes=num_classes, num_stuff_classes=0, num_queries=num_queries,
                num_transformer_feat_level=3, pixel_decoder=pd, enforce_decoder_input_project=False,
                positional_encoding=dict(num_feats=128, normalize=True), transformer_decoder=td,
                loss_cls=ph['loss_cls'], loss_mask=ph['loss_mask'], loss_dice=ph['loss_dice'])
    return dict(type='SAMSegMask2Former', data_preprocessor=q['data_preprocessor'], backbone=q['backbone'], neck=q['neck'],
                panoptic_head=head,
                panoptic_fusion_head=dict(type='MaskFormerFusionHead', num_things_classes=num_classes, num_stuff_classes=0,
                                          loss_panoptic=None, init_cfg=None),
                train_cfg=q['train_cfg'], test_cfg=q['test_cfg'])


def samdet(arch='base', num_classes=10, pretrain_name=None, ckpt=None):
    """configs/rsprompter/_base_/samdet.py:38-178 merged with samdet-<dataset>.py: Faster R-CNN R50-FPN
    detector + the HF SamModel prompted with its boxes."""
    name = pretrain_name or f'work_dirs/sam_cache/sam_vit_{arch}'
    a = rsprompter_anchor(arch, num_classes, (100, 5), pretrain_name, ckpt)
    rpn = copy.deepcopy(a['rpn_head'])
    rpn['anchor_generator'] = dict(type='AnchorGenerator', scales=[8], ratios=[0.5, 1.0, 2.0], strides=[4, 8, 16, 32, 64])
    train_cfg = copy.deepcopy(a['train_cfg'])
    train_cfg['rcnn']['assigner']['match_low_quality'] = False
    train_cfg['rcnn']['sampler']['num'] = 512
    del train_cfg['rcnn']['mask_size']
    test_cfg = copy.deepcopy(a['test_cfg'])
    del test_cfg['rcnn']['mask_thr_binary']
    detector = dict(
        type='FasterRCNN',
        backbone=dict(type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                      norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, style='pytorch',
                      init_cfg=dict(type='Pretrained', checkpoint='torchvision://resnet50')),
        neck=dict(type='FPN', in_channels=[256, 512, 1024, 2048], out_channels=256, num_outs=5),
        rpn_head=rpn,
        roi_head=dict(type='StandardRoIHead', bbox_roi_extractor=a['roi_head']['bbox_roi_extractor'],
                      bbox_head=a['roi_head']['bbox_head']),
        train_cfg=train_cfg, test_cfg=test_cfg)
    return dict(type='SAMDet', data_preprocessor=a['data_preprocessor'], detector=detector,
                segmentor=dict(type='RSSamModel', hf_pretrain_name=name,
                               init_cfg=dict(type='Pretrained', checkpoint=ckpt or f'{name}/pytorch_model.bin')))
