"""Golden vectors for the FORWARDS the reference wires out of mmcv building blocks, produced by executing the REAL
reference classes (`/root/reference`) on top of torch stand-ins for the mmcv / mmengine leaves (mmcv_standins.py) and
thin adapters for the transformers 4.38.1 -> 5.15 API drift:

  RSSimpleFPN.{__init__,forward}                  mmdet/rsprompter/models.py:1278-1363
  PseudoFeatureAggregator.{__init__,forward}      models.py:943-984   (with the real mmpretrain LayerNorm2d, utils/norm.py:52-90)
  RSFPN.forward                                   models.py:917-940
  RSPrompterAnchorMaskHead.{__init__,forward}     models.py:1597-1698 around HF SamMaskDecoder / SamPromptEncoder
  RSMask2FormerHead.{__init__,_forward_head,forward}   models.py:274-463 (on Mask2FormerHead.__init__ mask2former_head.py:63-155)
  MSDeformAttnPixelDecoder.{__init__,forward}     mmdet/models/layers/msdeformattn_pixel_decoder.py:45-246
     Mask2FormerTransformerEncoder / DeformableDetrTransformerEncoderLayer   mask2former_layers.py:10-53, deformable_detr_layers.py:237-249
     MlvlPointGenerator.single_level_grid_priors  point_generator.py (real), SinePositionalEncoding (real)
  Mask2FormerTransformerDecoder / ...DecoderLayer mask2former_layers.py:56-135, detr_layers.py:241-372
  ViTSAM.{__init__,forward} (+ TransformerEncoderLayer, Attention)   mmpretrain/models/backbones/vit_sam.py:160-602
     with the real mmpretrain build_norm_layer / resize_pos_embed (utils/norm.py:93-135, utils/embed.py:16-59)

Weights are NOT stored: both sides draw them from rsprompter_amd.synth.synth_state_dict (a pure function of seed, key
name and shape), so the replay in tests/test_oracle_golden.py also proves that the oracle's `state_dict` KEY LAYOUT
equals the real classes' (the key/shape lists are stored).  Run in the build container:
  python tests/golden/make_golden_forwards.py   ->   tests/golden/reference_vectors_forwards.pt
"""
import os
import sys
import types

import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import make_golden as mg  # noqa: E402
from mmcv_standins import Standins  # noqa: E402

OUT = os.path.join(HERE, 'reference_vectors_forwards.pt')


class Registry:
    """mmengine Registry surface the reference uses: register_module / get / build (+ scope)."""
    scope = 'golden'

    def __init__(self):
        self.table = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self.table[name or cls.__name__] = cls
            return cls
        return deco(module) if module is not None else deco

    def get(self, name):
        return self.table.get(str(name).split('.')[-1])

    def build(self, cfg, *a, **k):
        cfg = dict(cfg)
        t = cfg.pop('type')
        cls = self.get(t) if isinstance(t, str) else t
        if cls is None:
            if any(s in str(t) for s in ('Loss', 'Cost', 'Assigner', 'Sampler')):
                return nn.Identity()            # training-only members
            raise KeyError(t)
        return cls(**cfg)


class CD(dict):
    """ConfigDict stand-in (attribute access, recursive)."""

    def __init__(self, d=()):
        super().__init__()
        for k, v in dict(d).items():
            self[k] = CD(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __deepcopy__(self, memo):
        import copy
        return CD({k: copy.deepcopy(v, memo) for k, v in self.items()})


def install():
    mg._install_stubs()
    reg = Registry()
    reg.register_module('LN', module=nn.LayerNorm)
    st = Standins(reg)
    made = st.make()
    mods = sys.modules
    for n in ('mmdet.registry', 'mmpretrain.registry'):
        mods[n].MODELS = reg
        mods[n].TASK_UTILS = reg
    for name in ('mmcv.cnn',):
        for k in ('ConvModule', 'Conv2d', 'build_norm_layer', 'build_activation_layer'):
            setattr(mods[name], k, made[k])
    tr = mods['mmcv.cnn.bricks.transformer']
    tr.FFN, tr.MultiheadAttention, tr.PatchEmbed = made['FFN'], made['MultiheadAttention'], made['PatchEmbed']
    tr.MultiScaleDeformableAttention = made['MultiScaleDeformableAttention']
    mods['mmcv.ops'].MultiScaleDeformableAttention = made['MultiScaleDeformableAttention']
    mm = mods['mmengine.model']
    mm.ModuleList, mm.Sequential = nn.ModuleList, nn.Sequential
    for fn in ('caffe2_xavier_init', 'normal_init', 'xavier_init', 'constant_init'):
        setattr(mm, fn, lambda *a, **k: None)
    mods['mmengine'].ConfigDict = CD
    mods['mmengine.config'].ConfigDict = CD
    mods['mmengine.model.weight_init'].trunc_normal_ = lambda *a, **k: None
    mods['mmdet.utils'].reduce_mean = lambda x: x
    return reg, st


# ---- transformers 4.38.1 -> 5.15 drift adapters (the reference was written against 4.38.1, README.md:137) ----------
def sam_adapters(models):
    from transformers.models.sam import modeling_sam as hf
    from transformers.models.sam.configuration_sam import SamConfig

    class _SamConfig:
        """`SamConfig.from_pretrained(name)` without a hub / local config dir: default SAM configs (the three public
        checkpoints differ only in the vision tower, which is not built here), eager attention like 4.38.1."""
        @staticmethod
        def from_pretrained(name):
            cfg = SamConfig()
            for c in (cfg, cfg.vision_config, cfg.prompt_encoder_config, cfg.mask_decoder_config):
                c._attn_implementation = 'eager'
            cfg._full = cfg
            cfg.prompt_encoder_config._full = cfg
            return cfg

    class _MaskDecoder(hf.SamMaskDecoder):
        """4.38.1 signature: accepts output_attentions / attention_similarity / target_embedding, returns a 3-tuple."""

        def forward(self, image_embeddings, image_positional_embeddings, sparse_prompt_embeddings,
                    dense_prompt_embeddings, multimask_output, output_attentions=None, attention_similarity=None,
                    target_embedding=None):
            assert attention_similarity is None and target_embedding is None
            masks, iou = super().forward(image_embeddings=image_embeddings,
                                         image_positional_embeddings=image_positional_embeddings,
                                         sparse_prompt_embeddings=sparse_prompt_embeddings,
                                         dense_prompt_embeddings=dense_prompt_embeddings,
                                         multimask_output=multimask_output)[:2]
            return masks, iou, None

    def _prompt_encoder(prompt_cfg, shared_patch_embedding=None):
        """4.38.1: SamPromptEncoder(prompt_encoder_config, shared_patch_embedding); 5.15: SamPromptEncoder(SamConfig)."""
        m = hf.SamPromptEncoder(prompt_cfg._full)
        m.init_weights = lambda: None
        return m

    models.SamConfig = _SamConfig
    models.SamMaskDecoder = _MaskDecoder
    models.SamPromptEncoder = _prompt_encoder
    # the RS* wrappers subclass the HF classes only nominally (they call BaseModule.__init__ and hold the HF module as
    # a member), so their bases need no patching; `prompt_encoder.init_weights()` (models.py:303,1634) is mmengine's.
    models.RSSamPromptEncoder.init_weights = lambda self: None


def rnd(seed, *shape):
    """inputs are NOT stored either: a pure function of (seed, shape), regenerated by the replay test."""
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def keyshapes(m):
    return [(k, tuple(v.shape)) for k, v in m.state_dict().items()]


def seeded(m, seed):
    from rsprompter_amd.synth import synth_state_dict
    m.load_state_dict(synth_state_dict(m, seed), strict=True)
    return m.eval()


def load_sources():
    """execute the real reference sources the module forwards are made of; returns (m2h, vs, models) modules"""
    L = mg._load
    mods = sys.modules
    # ------------------------------------------------------------------ real leaf sources that ARE in the reference
    pe_mod = L('mmdet/models/layers/positional_encoding.py', '')
    pg = L('mmdet/models/task_modules/prior_generators/point_generator.py', '')
    mods['mmdet.models.task_modules.prior_generators'].MlvlPointGenerator = pg.MlvlPointGenerator
    mods['mmdet.models.layers'].SinePositionalEncoding = pe_mod.SinePositionalEncoding
    mods['mmdet.models'].SinePositionalEncoding = pe_mod.SinePositionalEncoding
    lu = types.ModuleType('mmdet.models.layers.transformer.utils')
    lu.inverse_sigmoid = lambda x, eps=1e-5: torch.log(x.clamp(eps, 1 - eps) / (1 - x).clamp(eps, 1))
    mods['mmdet.models.layers.transformer.utils'] = lu
    detr = L('mmdet/models/layers/transformer/detr_layers.py', '')
    L('mmdet/models/layers/transformer/deformable_detr_layers.py', '')
    m2f = L('mmdet/models/layers/transformer/mask2former_layers.py', '')
    mods['mmdet.models.layers.transformer'].Mask2FormerTransformerEncoder = m2f.Mask2FormerTransformerEncoder
    mods['mmdet.models.layers'].Mask2FormerTransformerDecoder = m2f.Mask2FormerTransformerDecoder
    mods['mmdet.models.layers'].DetrTransformerDecoder = detr.DetrTransformerDecoder
    pd = L('mmdet/models/layers/msdeformattn_pixel_decoder.py', '')
    L('mmdet/models/dense_heads/base_dense_head.py', '')
    L('mmdet/models/dense_heads/anchor_free_head.py', '')
    L('mmdet/models/dense_heads/maskformer_head.py', '')
    m2h = L('mmdet/models/dense_heads/mask2former_head.py', '')
    mods['mmdet.models'].Mask2FormerHead = m2h.Mask2FormerHead
    nrm = L('mmpretrain/models/utils/norm.py', '')
    emb = L('mmpretrain/models/utils/embed.py', '')
    mods['mmpretrain.models'].LayerNorm2d = nrm.LayerNorm2d
    pu = mods['mmpretrain.models.utils']
    pu.LayerNorm2d, pu.build_norm_layer, pu.resize_pos_embed = nrm.LayerNorm2d, nrm.build_norm_layer, emb.resize_pos_embed
    pu.to_2tuple = lambda v: (v, v) if not isinstance(v, (tuple, list)) else tuple(v)
    mods['mmpretrain.models.backbones.base_backbone'].BaseBackbone = mods['mmengine.model'].BaseModule
    vs = L('mmpretrain/models/backbones/vit_sam.py', '')
    models = L('mmdet/rsprompter/models.py', '')
    sam_adapters(models)
    return m2h, vs, models


@torch.no_grad()
def main():
    reg, st = install()
    m2h, vs, models = load_sources()
    out = {}

    # ------------------------------------------------------------------ RSSimpleFPN (+ RSFPN wiring) -- config values
    # of configs/rsprompter/_base_/rsprompter_anchor.py:82-89
    import rsprompter_amd as ra
    cfg = ra.Config.fromfile('/root/reference/configs/rsprompter/rsprompter_anchor-nwpu.py').model
    fpn = seeded(models.RSSimpleFPN(**{k: v for k, v in cfg.neck.feature_spliter.items() if k != 'type'}), 11)
    x = rnd(101, 2, 256, 8, 8)
    out['simple_fpn'] = dict(keys=keyshapes(fpn), seed=11, x=(101, (2, 256, 8, 8)), outs=[o[:, ::8].clone() for o in fpn(x)],
                             eps=[fpn.fpn1[1].eps, fpn.lateral_convs[0].norm_layer.eps, fpn.fpn_convs[3].norm_layer.eps])

    # ------------------------------------------------------------------ PseudoFeatureAggregator (real LayerNorm2d) + RSFPN
    cfg5 = ra.Config.fromfile('/root/reference/configs/rsprompter/rsprompter_anchor-nwpu-peft-512.py').model
    neck = seeded(models.RSFPN(feature_aggregator=dict(cfg5.neck.feature_aggregator),
                               feature_spliter=dict(cfg5.neck.feature_spliter)), 12)
    x = rnd(102, 1, 256, 8, 8)
    out['pseudo_neck'] = dict(keys=keyshapes(neck), seed=12, x=(102, (1, 256, 8, 8)),
                              agg=neck.feature_aggregator((x,)).clone(), outs=[o[:, ::8].clone() for o in neck((x,))])

    # ------------------------------------------------------------------ RSPrompterAnchorMaskHead
    mh_cfg = {k: v for k, v in cfg.roi_head.mask_head.items() if k != 'type'}
    mh_cfg['mask_decoder'] = dict(mh_cfg['mask_decoder'], init_cfg=None)
    mh = seeded(models.RSPrompterAnchorMaskHead(**mh_cfg), 13)
    R, B, hw = 5, 3, 8
    feats, emb_i = rnd(103, R, 256, 14, 14), rnd(104, B, 256, hw, hw)
    pe_i = rnd(105, 1, 256, hw, hw).repeat(B, 1, 1, 1)
    roi_img = torch.tensor([0., 0., 2., 2., 2.])            # image 1 has no RoI (models.py:1677-1679)
    low, iou = mh(feats, emb_i, pe_i, roi_img)
    out['anchor_mask_head'] = dict(keys=keyshapes(mh), seed=13, feats=(103, (R, 256, 14, 14)), emb=(104, (B, 256, hw, hw)),
                                   pe=(105, (1, 256, hw, hw)), roi_img=roi_img,
                                   low_res_masks=low.clone(), iou=iou.clone())

    # ------------------------------------------------------------------ RSMask2FormerHead (query path) on the reference's
    # own panoptic_head config (configs/rsprompter/rsprompter_query-nwpu.py merged over _base_/rsprompter_query.py)
    qcfg = ra.Config.fromfile('/root/reference/configs/rsprompter/rsprompter_query-nwpu.py').model
    ph = CD({k: v for k, v in qcfg.panoptic_head.items() if k != 'type'})
    NQ, NC = 12, 3
    ph.update(num_queries=NQ, num_things_classes=NC, train_cfg=None, test_cfg=None)
    ph['mask_decoder'] = CD(dict(ph['mask_decoder'], init_cfg=None))
    ph['loss_cls'] = CD(dict(ph['loss_cls'], class_weight=[1.0] * NC + [0.1]))
    head = seeded(models.RSMask2FormerHead(**ph), 14)
    Bq = 2
    xs = [rnd(110 + i, Bq, 256, s, s) for i, s in enumerate((64, 32, 16, 8, 4))]
    emb_q = rnd(120, Bq, 256, 16, 16)
    pe_q = rnd(121, 1, 256, 16, 16).repeat(Bq, 1, 1, 1)
    mask_features, memories = head.pixel_decoder(xs)
    cls_l, mask_l, mpp_l = head(xs, None, emb_q, pe_q)
    # one decoder layer on its own (mask2former_layers.py:73-135), with a mask that blocks whole rows of keys
    lay = head.transformer_decoder.layers[2]
    q, kv = rnd(130, Bq, NQ, 128), rnd(131, Bq, 64, 128)
    qpos, kpos = rnd(132, Bq, NQ, 128), rnd(133, Bq, 64, 128)
    am = rnd(134, Bq * 8, NQ, 64) < 0.0
    am[:, :, 0] = False
    out['query_head'] = dict(
        keys=keyshapes(head), seed=14, num_queries=NQ, num_classes=NC, batch=Bq,
        xs=[(110 + i, (Bq, 256, s_, s_)) for i, s_ in enumerate((64, 32, 16, 8, 4))], emb=(120, (Bq, 256, 16, 16)),
        pe=(121, (1, 256, 16, 16)),
        mask_features=mask_features[:, ::4, ::4, ::4].clone(), memories=[m[:, ::2].clone() for m in memories],
        cls_pred_all=[c.clone() for c in cls_l], mask_pred_plus_all=[m[:, :, ::4, ::4].clone() for m in mpp_l],
        cls_pred=cls_l[-1].clone(), mask_pred=mask_l[-1][:, :, ::2, ::2].clone(),
        mask_pred_first=mask_l[0][:, :, ::4, ::4].clone(), mask_pred_plus=mpp_l[-1][:, :, ::2, ::2].clone(),
        dec_layer=dict(index=2, q=(130, (Bq, NQ, 128)), kv=(131, (Bq, 64, 128)), qpos=(132, (Bq, NQ, 128)),
                       kpos=(133, (Bq, 64, 128)), mask=(134, (Bq * 8, NQ, 64)),
                       out=lay(query=q, key=kv, value=kv, query_pos=qpos, key_pos=kpos, cross_attn_mask=am,
                               query_key_padding_mask=None, key_padding_mask=None).clone()))

    # ------------------------------------------------------------------ ViTSAM (the encoder of the *-peft-512 configs),
    # built exactly like MMPretrainSamVisionEncoder builds it (models.py:822-832) at a small img_size
    vit = seeded(vs.ViTSAM(arch='base', img_size=256, patch_size=16, out_channels=256, use_abs_pos=True,
                           use_rel_pos=True, window_size=14), 15)
    img = rnd(140, 1, 3, 256, 256)
    y = vit(img)
    assert isinstance(y, tuple) and len(y) == 1
    out['vitsam'] = dict(keys=keyshapes(vit), seed=15, img_size=256, x=(140, (1, 3, 256, 256)), out=y[0][:, ::2].clone())

    torch.save(out, OUT)
    print('wrote', OUT, os.path.getsize(OUT) // 1024, 'KiB')
    for k, v in out.items():
        print(' ', k, len(v['keys']), 'state_dict keys')


if __name__ == '__main__':
    main()
