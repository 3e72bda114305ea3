"""Query prompter of RSPrompter on HIP kernels: MSDeformAttnPixelDecoder, RSMask2FormerHead,
RSMaskFormerFusionHead and the RSPrompterQuery detector (registry names / ctor kwargs of
configs/rsprompter/_base_/rsprompter_query.py:58-202).

Reference code mirrored here:
  RSPrompterQuery               mmdet/rsprompter/models.py:173-272  (Mask2Former -> MaskFormer maskformer.py:18-170)
  RSMask2FormerHead             models.py:274-463, 633-658           (Mask2FormerHead mask2former_head.py:62-156)
  MSDeformAttnPixelDecoder      mmdet/models/layers/msdeformattn_pixel_decoder.py:21-246
  Mask2Former decoder layer     mmdet/models/layers/transformer/mask2former_layers.py:73-135
  RSMaskFormerFusionHead        models.py:661-715 + maskformer_fusion_head.py:126-182 + structures/mask/utils.py:56-77

Inference shortcut: with decoder_plus=True the cross-attention masks depend only on
`mask_pred_plus`, and predict() keeps only the LAST SAM-decoder output (models.py:644-646), so the SAM mask
decoder runs once instead of seven times -- result identical.
"""
import copy
import math

import torch

from . import debug, ops
from .anchor_heads import _metas_of, _sine_pe
from .detectors import BaseDetectorHIP
from .necks import conv3x3_weight
from .nnutil import HIPModule, add_param, nchw_view, nhwc_view
from .registry import MODELS
from .structures import InstanceData


def _g(root, dotted):
    for p in dotted.split('.'):
        root = getattr(root, p)
    return root


def _pw(mod, bias=True):
    w = mod.weight
    return ops.PackedWeight(w.reshape(w.shape[0], -1), mod.bias if bias and hasattr(mod, 'bias') else None)


def _add_linear(root, name, cout, cin):
    add_param(root, name + '.weight', (cout, cin))
    add_param(root, name + '.bias', (cout,))


def _add_ln(root, name, c):
    add_param(root, name + '.weight', (c,), 1.0)
    add_param(root, name + '.bias', (c,))


@MODELS.register_module()
class MSDeformAttnPixelDecoder(HIPModule):
    def __init__(self, in_channels=(256, 256, 256, 256, 256), strides=(4, 8, 16, 32, 64), feat_channels=256,
                 out_channels=256, num_outs=3, norm_cfg=None, act_cfg=None, encoder=None, positional_encoding=None,
                 init_cfg=None):
        super().__init__()
        lc = encoder['layer_cfg']
        sa = lc['self_attn_cfg']
        if (sa['num_heads'], sa['num_points']) != (8, 4) or sa['embed_dims'] not in (128, 256) or \
                not 1 <= sa['num_levels'] <= min(5, len(in_channels)):
            raise NotImplementedError('the MSDeformAttn kernel covers 8 heads x (16 | 32), 1-5 levels, 4 points: the '
                                      'RSPrompter (128) and samseg-mask2former (256) configurations use 3 levels')
        if feat_channels != sa['embed_dims'] or (norm_cfg or {}).get('num_groups', 32) != 32:
            raise NotImplementedError('feat_channels must equal the encoder width; GroupNorm(32) only')
        self.in_channels, self.strides = list(in_channels), list(strides)
        self.n_in, self.n_enc, self.num_layers = len(in_channels), sa['num_levels'], encoder['num_layers']
        self.feat, self.out_channels, self.num_outs = feat_channels, out_channels, num_outs
        self.ffn_dim = lc['ffn_cfg']['feedforward_channels']
        self.pe_feats = (positional_encoding or {}).get('num_feats', 128)
        f = feat_channels
        for i in range(self.n_enc):
            add_param(self, f'input_convs.{i}.conv.weight', (f, in_channels[self.n_in - 1 - i], 1, 1))
            add_param(self, f'input_convs.{i}.conv.bias', (f,))
            _add_ln(self, f'input_convs.{i}.gn', f)
        for n in range(self.num_layers):
            p = f'encoder.layers.{n}'
            _add_linear(self, p + '.self_attn.sampling_offsets', 8 * self.n_enc * 4 * 2, f)
            _add_linear(self, p + '.self_attn.attention_weights', 8 * self.n_enc * 4, f)
            _add_linear(self, p + '.self_attn.value_proj', f, f)
            _add_linear(self, p + '.self_attn.output_proj', f, f)
            _add_linear(self, p + '.ffn.layers.0.0', self.ffn_dim, f)
            _add_linear(self, p + '.ffn.layers.1', f, self.ffn_dim)
            _add_ln(self, p + '.norms.0', f)
            _add_ln(self, p + '.norms.1', f)
        add_param(self, 'level_encoding.weight', (self.n_enc, f))
        for i in range(self.n_in - self.n_enc):
            add_param(self, f'lateral_convs.{i}.conv.weight', (f, in_channels[i], 1, 1))
            _add_ln(self, f'lateral_convs.{i}.gn', f)
            add_param(self, f'output_convs.{i}.conv.weight', (f, f, 3, 3))
            _add_ln(self, f'output_convs.{i}.gn', f)
        add_param(self, 'mask_feature.weight', (out_channels, f, 1, 1))
        add_param(self, 'mask_feature.bias', (out_channels,))
        self._const = {}

    def _apply(self, fn, *a, **kw):
        self._const = {}
        return super()._apply(fn, *a, **kw)

    def _pack(self):
        P = dict(inp=[_pw(_g(self, f'input_convs.{i}.conv')) for i in range(self.n_enc)], layers=[])
        for n in range(self.num_layers):
            L = _g(self, f'encoder.layers.{n}')
            sa = L.self_attn
            # offsets (192 at 3 levels) and attention logits (96) come out of ONE GEMM on (query + pos)
            w = torch.cat([sa.sampling_offsets.weight.detach(), sa.attention_weights.weight.detach()], 0)
            b = torch.cat([sa.sampling_offsets.bias.detach(), sa.attention_weights.bias.detach()], 0)
            P['layers'].append(dict(ow=ops.PackedWeight(w, b), value=_pw(sa.value_proj), out=_pw(sa.output_proj),
                                    f0=_pw(_g(L, 'ffn.layers.0.0')), f1=_pw(_g(L, 'ffn.layers.1'))))
        P['lat'] = [_pw(_g(self, f'lateral_convs.{i}.conv'), bias=False) for i in range(self.n_in - self.n_enc)]
        P['outc'] = [ops.PackedWeight(conv3x3_weight(_g(self, f'output_convs.{i}.conv').weight.detach()))
                     for i in range(self.n_in - self.n_enc)]
        P['mask_feature'] = _pw(self.mask_feature)
        self._packed = P

    def _constants(self, shapes, dev):
        """input-independent tables: level_encoding + sine PE (msdeformattn_pixel_decoder.py:171-173) and the
        normalised reference points ((x+.5)/W, (y+.5)/H) (:175-182), concatenated over the encoder levels."""
        key = (tuple(shapes), str(dev))
        if key not in self._const:
            pos, ref = [], []
            for i, (h, w) in enumerate(shapes):
                pe = _sine_pe(h, w, self.pe_feats)[0].permute(1, 2, 0).reshape(h * w, -1)
                pos.append(self.level_encoding.weight.detach().cpu()[i].view(1, -1) + pe)
                s = self.strides[self.n_in - 1 - i]
                sx = (torch.arange(0, w) + 0.5) * s
                sy = (torch.arange(0, h) + 0.5) * s
                xx = sx.repeat(h)
                yy = sy.view(-1, 1).repeat(1, w).view(-1)
                ref.append(torch.stack([xx, yy], -1) / (torch.tensor([w, h], dtype=torch.float32) * s))
            self._const = {key: (torch.cat(pos, 0).contiguous().to(dev), torch.cat(ref, 0).contiguous().to(dev))}
        return self._const[key]

    def forward(self, feats):
        """feats: 5 logical-NCHW channels-last levels -> (mask_feature [B,out,H0,W0], [3 memories low->high res])."""
        if self._packed is None:
            self._pack()
        P = self._packed
        x = [nhwc_view(f) for f in feats]
        B = x[0].shape[0]
        f = self.feat
        toks, shapes = [], []
        for i in range(self.n_enc):
            xi = x[self.n_in - 1 - i]
            _, h, w, c = xi.shape
            gn = _g(self, f'input_convs.{i}.gn')
            t = ops.gemm(xi.view(B * h * w, c), P['inp'][i])
            toks.append(ops.groupnorm(t.view(B, h * w, f), gn.weight, gn.bias, 32))
            shapes.append((h, w))
        q = torch.cat(toks, 1).contiguous()                       # [B, Ntok, 128]
        Ntok = q.shape[1]
        q = q.view(B * Ntok, f)
        pos, ref = self._constants(shapes, q.device)
        for n in range(self.num_layers):
            L, W = _g(self, f'encoder.layers.{n}'), P['layers'][n]
            qp = ops.add_rows(q, pos, vmod=Ntok)
            value = ops.gemm(q, W['value'])
            ow = ops.gemm(qp, W['ow'])
            samp = ops.msdeform_attn(value, ow, ref, B, Ntok, shapes, head_dim=f // 8)
            q1 = ops.gemm(samp, W['out'], res=q)                   # output_proj + identity
            q1 = ops.layernorm(q1, _g(L, 'norms.0').weight, _g(L, 'norms.0').bias, 1e-5)
            hmid = ops.gemm(q1, W['f0'], act=ops.ACT_RELU)
            q2 = ops.gemm(hmid, W['f1'], res=q1)
            q = ops.layernorm(q2, _g(L, 'norms.1').weight, _g(L, 'norms.1').bias, 1e-5)
        mem = q.view(B, Ntok, f)
        outs, start = [], 0
        for (h, w) in shapes:
            outs.append(mem[:, start:start + h * w].contiguous().view(B, h, w, f))
            start += h * w
        for i in range(self.n_in - self.n_enc - 1, -1, -1):       # module index == feature index (:232-242)
            xi = x[i]
            _, h, w, c = xi.shape
            lg, og = _g(self, f'lateral_convs.{i}.gn'), _g(self, f'output_convs.{i}.gn')
            cur = ops.gemm(xi.view(B * h * w, c), P['lat'][i], bias=None)
            up = ops.resize_bilinear(outs[-1], (h, w))
            y = ops.groupnorm(cur.view(B, h * w, f), lg.weight, lg.bias, 32, add=up.view(B, h * w, f))
            y = ops.gemm(y.view(B, h, w, f), P['outc'][i], bias=None, conv=(3, 1, 1))
            outs.append(ops.groupnorm(y.view(B, h * w, f), og.weight, og.bias, 32, relu=True).view(B, h, w, f))
        last = outs[-1]
        _, h, w, _ = last.shape
        mf = ops.gemm(last.view(B * h * w, f), P['mask_feature'])
        return nchw_view(mf.view(B, h, w, self.out_channels)), [nchw_view(o) for o in outs[:self.num_outs]]


class LazyUpsampledMasks:
    """`F.interpolate(mask_pred_results, batch_input_shape)` of models.py:652-656 kept symbolic: the fusion head
    samples the low-res logits directly (the two interpolations are evaluated per output pixel in
    rsp_query_mask_post), so the [B, Nq, 1024, 1024] fp32 tensor (400 MiB / image at Nq=100) is never written."""

    def __init__(self, low_res, size):
        self.low_res, self.size = low_res, tuple(size)
        self.shape = tuple(low_res.shape[:2]) + self.size

    def materialize(self):
        B, Nq, h, w = self.low_res.shape
        x = self.low_res.reshape(B * Nq, h, w, 1).expand(-1, -1, -1, 4).contiguous()
        y = ops.resize_bilinear(x, self.size)
        return y[..., 0].reshape(B, Nq, *self.size)

    def __len__(self):
        return self.low_res.shape[0]

    def __iter__(self):
        return iter(self.low_res)


class _Mask2FormerCore(HIPModule):
    """What Mask2FormerHead (mask2former_head.py:62-156) and RSMask2FormerHead (models.py:274-338) share: the pixel
    decoder, the masked-attention transformer decoder (mask2former_layers.py:73-135), the query / level embeddings and
    the mask-embedding MLP.  `_decode` runs `forward` up to the last `_forward_head` (mask2former_head.py:403-456)."""

    def _init_core(self, in_channels, feat_channels, out_channels, num_things_classes, num_stuff_classes, num_queries,
                   num_transformer_feat_level, pixel_decoder, enforce_decoder_input_project, transformer_decoder,
                   positional_encoding, train_cfg, test_cfg):
        self.num_things_classes, self.num_stuff_classes = num_things_classes, num_stuff_classes
        self.num_classes = num_things_classes + num_stuff_classes
        self.num_queries, self.num_transformer_feat_level = num_queries, num_transformer_feat_level
        td = transformer_decoder
        self.num_heads = td['layer_cfg']['cross_attn_cfg']['num_heads']
        self.num_transformer_decoder_layers = td['num_layers']
        self.feat_channels, self.out_channels = feat_channels, out_channels
        self.ffn_dim = td['layer_cfg']['ffn_cfg']['feedforward_channels']
        # mask2former_head.py:93-100: Conv2d(feat, feat, 1) per level when the widths differ or it is enforced
        self.input_proj = (td['layer_cfg']['cross_attn_cfg']['embed_dims'] != feat_channels
                           or bool(enforce_decoder_input_project))
        if td['layer_cfg']['cross_attn_cfg']['embed_dims'] != feat_channels:
            raise NotImplementedError('decoder width != feat_channels (pixel decoder memories are feat_channels wide)')
        # mask2former_head.py:106-107
        assert pixel_decoder['encoder']['layer_cfg']['self_attn_cfg']['num_levels'] == num_transformer_feat_level
        if pixel_decoder.get('num_outs', 3) < num_transformer_feat_level:
            raise ValueError(f'pixel_decoder.num_outs={pixel_decoder.get("num_outs", 3)} memories for '
                             f'{num_transformer_feat_level} transformer feature levels (mask2former_head.py:409 indexes '
                             'multi_scale_memorys[i] for every level)')
        pd = copy.deepcopy(dict(pixel_decoder))
        pd.update(in_channels=in_channels, feat_channels=feat_channels, out_channels=out_channels)
        self.pixel_decoder = MODELS.build(pd)
        self.pe_feats = (positional_encoding or {}).get('num_feats', 128)
        f = feat_channels
        for n in range(self.num_transformer_decoder_layers):
            p = f'transformer_decoder.layers.{n}'
            for a in ('self_attn', 'cross_attn'):
                add_param(self, f'{p}.{a}.attn.in_proj_weight', (3 * f, f))
                add_param(self, f'{p}.{a}.attn.in_proj_bias', (3 * f,))
                _add_linear(self, f'{p}.{a}.attn.out_proj', f, f)
            _add_linear(self, p + '.ffn.layers.0.0', self.ffn_dim, f)
            _add_linear(self, p + '.ffn.layers.1', f, self.ffn_dim)
            for j in range(3):
                _add_ln(self, f'{p}.norms.{j}', f)
        _add_ln(self, 'transformer_decoder.post_norm', f)
        add_param(self, 'query_embed.weight', (num_queries, f))
        add_param(self, 'query_feat.weight', (num_queries, f))
        add_param(self, 'level_embed.weight', (num_transformer_feat_level, f))
        if self.input_proj:
            for i in range(num_transformer_feat_level):
                add_param(self, f'decoder_input_projs.{i}.weight', (f, f, 1, 1))
                add_param(self, f'decoder_input_projs.{i}.bias', (f,))
        self.test_cfg, self.train_cfg = test_cfg, train_cfg
        self._const = {}

    def _add_mask_embed(self):
        f = self.feat_channels
        _add_linear(self, 'mask_embed.0', f, f)
        _add_linear(self, 'mask_embed.2', f, f)
        _add_linear(self, 'mask_embed.4', self.out_channels, f)

    def _apply(self, fn, *a, **kw):
        self._const = {}
        return super()._apply(fn, *a, **kw)

    def _pack_core(self, linears):
        f = self.feat_channels
        P = dict(layers=[])
        for n in range(self.num_transformer_decoder_layers):
            L = _g(self, f'transformer_decoder.layers.{n}')
            d = {}
            for a in ('self_attn', 'cross_attn'):
                at = _g(L, a + '.attn')
                w, b = at.in_proj_weight.detach(), at.in_proj_bias.detach()
                for j, nm in enumerate('qkv'):
                    d[f'{a}.{nm}'] = ops.PackedWeight(w[j * f:(j + 1) * f], b[j * f:(j + 1) * f])
                d[f'{a}.o'] = _pw(at.out_proj)
            d['f0'], d['f1'] = _pw(_g(L, 'ffn.layers.0.0')), _pw(_g(L, 'ffn.layers.1'))
            P['layers'].append(d)
        for nm in tuple(linears):
            P[nm] = _pw(_g(self, nm))
        if self.input_proj:
            for i in range(self.num_transformer_feat_level):
                m = _g(self, f'decoder_input_projs.{i}')
                P[f'in_proj.{i}'] = ops.PackedWeight(m.weight.detach().reshape(f, f), m.bias)
        return P

    def _pos_tables(self, shapes, dev):
        key = (tuple(shapes), str(dev))
        if key not in self._const:
            self._const = {key: [_sine_pe(h, w, self.pe_feats)[0].permute(1, 2, 0).reshape(h * w, -1).contiguous().to(dev)
                                 for (h, w) in shapes]}
        return self._const[key]

    def _mlp(self, x, names, last_act=False):
        P = self._packed
        for i, nm in enumerate(names):
            act = ops.ACT_RELU if (i + 1 < len(names) or last_act) else ops.ACT_NONE
            x = ops.gemm(x, P[nm], act=act)
        return x

    def _mha(self, W, pfx, q_in, k_in, v_in, identity, B, Tq, Tk, mask=None):
        """mmcv MultiheadAttention(batch_first) over nn.MultiheadAttention: returns identity + out_proj(attn)."""
        f, nh = self.feat_channels, self.num_heads
        dh = f // nh
        q = ops.gemm(q_in, W[pfx + '.q'])
        k = ops.gemm(k_in, W[pfx + '.k'])
        v = ops.gemm(v_in, W[pfx + '.v'])
        o = torch.empty_like(q)
        ops.attention(q, k, v, o, B=B, nh=nh, dh=dh, Tq=Tq, Tk=Tk, scale=dh ** -0.5,
                      q_strides=(Tq * f, f, dh), k_strides=(Tk * f, f, dh), v_strides=(Tk * f, f, dh),
                      o_strides=(Tq * f, f, dh), mask=mask)
        return ops.gemm(o, W[pfx + '.o'], res=identity)

    def _head_light(self, qf, mf_planes, B, HW0):
        """the per-layer part of `_forward_head` (mask2former_head.py:361-367 / models.py:340-357): post_norm and the
        mask logits einsum('bqc,bchw->bqhw') that the next layer's attention mask is thresholded from."""
        pn = _g(self, 'transformer_decoder.post_norm')
        dn = ops.layernorm(qf, pn.weight, pn.bias, 1e-5)
        me = self._mlp(dn, ('mask_embed.0', 'mask_embed.2', 'mask_embed.4'))
        Nq = self.num_queries
        mpp = torch.empty((B, Nq, HW0), dtype=torch.float32, device=qf.device)
        for b in range(B):   # mask_feature (as fp16 planes) is the GEMM "weight"
            ops.gemm(me[b * Nq:(b + 1) * Nq], ops.PlaneWeight(mf_planes, b * HW0, HW0), out=mpp[b], bias=None)
        return dn, mpp

    def _decode(self, x, stage_masks=None):
        """-> (dn [B*Nq, f] post-normed last query features, mask logits [B, Nq, H0, W0], trace).
        stage_masks(dn) -> [B, Nq, Hm, Wm] replaces the mask-embedding logits as the source of the next layer's attention
        mask (RSMask2FormerHead with decoder_plus=False: the SAM decoder's own masks, models.py:380-385)."""
        P = self._packed
        B = x[0].shape[0]
        f, Nq = self.feat_channels, self.num_queries
        mask_features, mem = self.pixel_decoder(x)
        mf = nhwc_view(mask_features)
        H0, W0 = mf.shape[1], mf.shape[2]
        mf_planes = ops.to_planes(mf.reshape(B * H0 * W0, self.out_channels))
        shapes = [tuple(m.shape[-2:]) for m in mem]
        pos_tabs = self._pos_tables(shapes, mf.device)
        dec_in, dec_kin = [], []
        for i in range(self.num_transformer_feat_level):
            m = nhwc_view(mem[i]).reshape(-1, f)
            if self.input_proj:
                m = ops.gemm(m, P[f'in_proj.{i}'])                                           # Conv2d 1x1 (:404-405)
            d = ops.add_rows(m, self.level_embed.weight[i:i + 1].contiguous(), vmod=1)       # + level_embed (:409-410)
            dec_in.append(d)
            dec_kin.append(ops.add_rows(d, pos_tabs[i], vmod=pos_tabs[i].shape[0]))          # key + key_pos
        qf = self.query_feat.weight.detach().unsqueeze(0).expand(B, -1, -1).reshape(B * Nq, f).contiguous()
        qe = self.query_embed.weight.detach()
        trace = dict(attn_masks=[], query_feats=[], mask_pred_plus_all=[], mask_features=mask_features, memory=mem)

        def head(qf_):
            if stage_masks is None:
                dn_, mpp_ = self._head_light(qf_, mf_planes, B, H0 * W0)
                return dn_, mpp_.view(B, Nq, H0, W0)
            pn = _g(self, 'transformer_decoder.post_norm')
            dn_ = ops.layernorm(qf_, pn.weight, pn.bias, 1e-5)
            return dn_, stage_masks(dn_)
        dn, mpp = head(qf)
        trace['mask_pred_plus_all'].append(mpp)
        for i in range(self.num_transformer_decoder_layers):
            lvl = i % self.num_transformer_feat_level
            h, w = shapes[lvl]
            attn_mask = ops.query_attn_mask(mpp.contiguous(), (h, w))                        # :386-391 + :439-442
            trace['attn_masks'].append(attn_mask)
            W, L = P['layers'][i], _g(self, f'transformer_decoder.layers.{i}')
            qp = ops.add_rows(qf, qe, vmod=Nq)
            qf = self._mha(W, 'cross_attn', qp, dec_kin[lvl], dec_in[lvl], qf, B, Nq, h * w, mask=attn_mask)
            qf = ops.layernorm(qf, _g(L, 'norms.0').weight, _g(L, 'norms.0').bias, 1e-5)
            qp = ops.add_rows(qf, qe, vmod=Nq)
            qf = self._mha(W, 'self_attn', qp, qp, qf, qf, B, Nq, Nq)
            qf = ops.layernorm(qf, _g(L, 'norms.1').weight, _g(L, 'norms.1').bias, 1e-5)
            hmid = ops.gemm(qf, W['f0'], act=ops.ACT_RELU)
            qf = ops.gemm(hmid, W['f1'], res=qf)
            qf = ops.layernorm(qf, _g(L, 'norms.2').weight, _g(L, 'norms.2').bias, 1e-5)
            trace['query_feats'].append(qf)
            dn, mpp = head(qf)
            trace['mask_pred_plus_all'].append(mpp)
        return dn, mpp, trace


@MODELS.register_module()
class Mask2FormerHead(_Mask2FormerCore):
    """The standard mmdet head of the `samseg-mask2former` configs (mask2former_head.py:62-156, 340-460; predict:
    maskformer_head.py:569-604): single-Linear class head, mask logits = mask_embed . mask_feature, no SAM decoder."""

    def __init__(self, in_channels=None, feat_channels=256, out_channels=256, num_things_classes=80,
                 num_stuff_classes=53, num_queries=100, num_transformer_feat_level=3, pixel_decoder=None,
                 enforce_decoder_input_project=False, transformer_decoder=None, positional_encoding=None, loss_cls=None,
                 loss_mask=None, loss_dice=None, train_cfg=None, test_cfg=None, init_cfg=None, **kwargs):
        super().__init__()
        self._init_core(in_channels, feat_channels, out_channels, num_things_classes, num_stuff_classes, num_queries,
                        num_transformer_feat_level, pixel_decoder, enforce_decoder_input_project, transformer_decoder,
                        positional_encoding, train_cfg, test_cfg)
        _add_linear(self, 'cls_embed', self.num_classes + 1, feat_channels)
        self._add_mask_embed()

    def _pack(self):
        self._packed = self._pack_core(('cls_embed', 'mask_embed.0', 'mask_embed.2', 'mask_embed.4'))

    def forward(self, x, batch_data_samples=None):
        """-> (cls [B,Nq,nc+1], mask logits [B,Nq,H0,W0], trace): the LAST decoder stage, which is all predict reads"""
        if self._packed is None:
            self._pack()
        dn, mask_pred, trace = self._decode(x)
        B = x[0].shape[0]
        cls = ops.gemm(dn, self._packed['cls_embed']).view(B, self.num_queries, self.num_classes + 1)
        return cls, mask_pred, trace

    def predict(self, x, batch_data_samples):
        """maskformer_head.py:569-604; the bilinear up-sampling to batch_input_shape stays symbolic (LazyUpsampledMasks)."""
        metas = _metas_of(batch_data_samples)
        cls, mask_pred, trace = self(x, batch_data_samples)
        self._last_trace = debug.keep(trace)
        size = metas[0].get('batch_input_shape', metas[0].get('pad_shape'))
        return cls, LazyUpsampledMasks(mask_pred, size[:2])


@MODELS.register_module()
class RSMask2FormerHead(_Mask2FormerCore):
    def __init__(self, mask_decoder, decoder_plus, with_sincos=True, per_pointset_point=1, multimask_output=False,
                 attention_similarity=None, target_embedding=None, output_attentions=None, in_channels=None,
                 feat_channels=128, out_channels=256, num_things_classes=80, num_stuff_classes=0, num_queries=100,
                 num_transformer_feat_level=3, pixel_decoder=None, enforce_decoder_input_project=False,
                 transformer_decoder=None, positional_encoding=None, loss_cls=None, loss_mask=None, loss_dice=None,
                 train_cfg=None, test_cfg=None, init_cfg=None, **kwargs):
        super().__init__()
        if multimask_output and not decoder_plus:
            # models.py:369-385: `mask_pred.reshape(img_bs, -1, h, w)` folds the three masks into the query axis; without
            # decoder_plus those [B, 3 Nq, h, w] masks are the attention-mask source of Nq queries and the reference fails in
            # its first decoder layer (nn.MultiheadAttention: "The shape of the 3D attn_mask is ..."; recorded from the real
            # class in tests/golden/reference_vectors_query_options.pt)
            raise ValueError('RSMask2FormerHead(multimask_output=True) needs decoder_plus=True: the reference itself fails '
                             'with the folded masks as its attention mask')
        self.multimask_output = bool(multimask_output)
        self.per_pointset_point, self.decoder_plus, self.with_sincos = per_pointset_point, bool(decoder_plus), bool(with_sincos)
        self._init_core(in_channels, feat_channels, out_channels, num_things_classes, num_stuff_classes, num_queries,
                        num_transformer_feat_level, pixel_decoder, enforce_decoder_input_project, transformer_decoder,
                        positional_encoding, train_cfg, test_cfg)
        f = feat_channels
        _add_linear(self, 'cls_embed.0', f, f)
        _add_linear(self, 'cls_embed.2', self.num_classes + 1, f)
        _add_linear(self, 'point_emb.0', f // 2, f)
        _add_linear(self, 'point_emb.2', f // 2, f // 2)
        _add_linear(self, 'point_emb.4', out_channels * (2 if with_sincos else 1) * per_pointset_point, f // 2)
        self.mask_decoder = MODELS.build(mask_decoder)
        if not self.decoder_plus:
            # models.py:303-307: no mask-embedding MLP, the prompt encoder's no_mask_embed as the dense prompt
            add_param(self, 'no_mask_embed.weight', (1, out_channels))
            return
        self._add_mask_embed()
        # the reference keeps prompt_encoder.mask_embed as `sam_mask_embed` (models.py:297-305)
        add_param(self, 'sam_mask_embed.conv1.weight', (4, 1, 2, 2))
        add_param(self, 'sam_mask_embed.conv1.bias', (4,))
        add_param(self, 'sam_mask_embed.conv2.weight', (16, 4, 2, 2))
        add_param(self, 'sam_mask_embed.conv2.bias', (16,))
        add_param(self, 'sam_mask_embed.conv3.weight', (out_channels, 16, 1, 1))
        add_param(self, 'sam_mask_embed.conv3.bias', (out_channels,))
        _add_ln(self, 'sam_mask_embed.layer_norm1', 4)
        _add_ln(self, 'sam_mask_embed.layer_norm2', 16)

    def _pack(self):
        lin = ('cls_embed.0', 'cls_embed.2', 'point_emb.0', 'point_emb.2', 'point_emb.4')
        if not self.decoder_plus:
            self._packed = self._pack_core(lin)
            return
        P = self._pack_core(lin + ('mask_embed.0', 'mask_embed.2', 'mask_embed.4'))
        sm = self.sam_mask_embed
        P['sam_embed'] = dict(conv1_w=sm.conv1.weight.detach().contiguous(), conv1_b=sm.conv1.bias.detach(),
                              ln1_w=sm.layer_norm1.weight.detach(), ln1_b=sm.layer_norm1.bias.detach(),
                              conv2_w=sm.conv2.weight.detach().contiguous(), conv2_b=sm.conv2.bias.detach(),
                              ln2_w=sm.layer_norm2.weight.detach(), ln2_b=sm.layer_norm2.bias.detach(),
                              conv3_w=sm.conv3.weight.detach().reshape(self.out_channels, 16).contiguous(),
                              conv3_b=sm.conv3.bias.detach())
        self._packed = P

    def forward(self, x, batch_data_samples=None, image_embeddings=None, image_positional_embeddings=None):
        """models.py:395-463 (inference schedule).  Returns (cls [B,Nq,nc+1], SAM low-res masks [B,Nq,4h,4w], trace);
        multimask_output=True: masks [B, 3 Nq, 4h, 4w], entry 3 q + j = mask token j + 1 of prompt set q (models.py:369-380:
        the reference's reshape folds the three masks into the query axis; the fusion head then reads mask `query index`)."""
        if self._packed is None:
            self._pack()
        P = self._packed
        B = x[0].shape[0]
        Nq = self.num_queries
        emb = nhwc_view(image_embeddings)
        he, we = emb.shape[1], emb.shape[2]
        roi_img = torch.arange(B, dtype=torch.int32, device=emb.device).repeat_interleave(Nq).contiguous()

        def prompts(dn_):
            pe_ = self._mlp(dn_, ('point_emb.0', 'point_emb.2', 'point_emb.4'))
            if self.with_sincos:
                pe_ = ops.sincos_pairs(pe_)                   # sin(x[..., ::2]) + x[..., 1::2] (models.py:346-347)
            return pe_.view(B * Nq, self.per_pointset_point, self.out_channels)

        if not self.decoder_plus:
            # models.py:361-385: the SAM decoder runs in EVERY stage with the no_mask dense prompt and its masks drive the
            # next layer's attention mask (the reference expands the dense prompt to img_bs rows, which only broadcasts
            # for one image per batch; the evident intent -- the same vector for every prompt set -- is what runs here)
            def sam_stage(dn_):
                m_, _ = self.mask_decoder.mask_decoder.decode(image_embeddings, image_positional_embeddings, prompts(dn_),
                                                              self.no_mask_embed.weight.reshape(-1), roi_img, want_iou=False)
                return m_.view(B, Nq, m_.shape[-2], m_.shape[-1])
            dn, mask_pred, trace = self._decode(x, stage_masks=sam_stage)
            cls = self._mlp(dn, ('cls_embed.0', 'cls_embed.2')).view(B, Nq, self.num_classes + 1)
            trace.update(mask_pred_plus=None, sparse_embeddings=prompts(dn))
            return cls, mask_pred, trace
        dn, mpp, trace = self._decode(x)
        H0, W0 = mpp.shape[-2:]
        # ---- the last `_forward_head` in full: class logits, prompts, dense prompt, ONE SAM decoder call ----
        cls = self._mlp(dn, ('cls_embed.0', 'cls_embed.2')).view(B, Nq, self.num_classes + 1)
        sparse = prompts(dn)
        src = ops.sam_mask_embed(mpp.reshape(B * Nq, H0, W0), emb.reshape(B * he * we, -1), roi_img, P['sam_embed'], he, we)
        ident = torch.arange(B * Nq, dtype=torch.int32, device=emb.device)
        masks, _ = self.mask_decoder.mask_decoder.decode(None, image_positional_embeddings, sparse, None, ident,
                                                         want_iou=False, src_rows=src, hw=(he, we), src_is_identity=True,
                                                         multimask_output=self.multimask_output)
        mask_pred = masks.view(B, -1, masks.shape[-2], masks.shape[-1])
        trace.update(mask_pred_plus=mpp, sparse_embeddings=sparse)
        return cls, mask_pred, trace

    def predict(self, x, batch_data_samples, image_embeddings=None, image_positional_embeddings=None):
        """models.py:633-658."""
        metas = _metas_of(batch_data_samples)
        cls, mask_pred, trace = self(x, batch_data_samples, image_embeddings, image_positional_embeddings)
        self._last_trace = debug.keep(trace)
        size = metas[0].get('batch_input_shape', metas[0].get('pad_shape'))
        return cls, LazyUpsampledMasks(mask_pred, size[:2])


@MODELS.register_module()
class RSMaskFormerFusionHead(HIPModule):
    def __init__(self, num_things_classes=80, num_stuff_classes=53, test_cfg=None, loss_panoptic=None, init_cfg=None,
                 **kwargs):
        super().__init__()
        self.num_things_classes, self.num_stuff_classes = num_things_classes, num_stuff_classes
        self.num_classes = num_things_classes + num_stuff_classes
        self.test_cfg = test_cfg or {}

    def predict(self, mask_cls_results, mask_pred_results, batch_data_samples, rescale=False, **kwargs):
        """models.py:662-715 + instance_postprocess (maskformer_fusion_head.py:126-182)."""
        metas = _metas_of(batch_data_samples)
        cfg = self.test_cfg
        if cfg.get('panoptic_on', True) or cfg.get('semantic_on', False):
            raise NotImplementedError('only instance_on (the RSPrompter test_cfg) is implemented')
        if not isinstance(mask_pred_results, LazyUpsampledMasks):
            raise TypeError('expected the LazyUpsampledMasks returned by RSMask2FormerHead.predict')
        low = mask_pred_results.low_res
        B, Nq = mask_cls_results.shape[:2]                  # maskformer_fusion_head.py:143 `num_queries = mask_cls.shape[0]`
        nc = self.num_classes
        k = min(cfg.get('max_per_image', 100), Nq * nc)
        scores, flat = ops.query_topk(mask_cls_results.contiguous(), k)
        results = []
        for b, meta in enumerate(metas):
            oh, ow = meta['ori_shape'][:2]
            sf = meta['scale_factor']
            Hb, Wb = mask_pred_results.size
            crop = (min(int(oh * sf[1]), Hb), min(int(ow * sf[0]), Wb))
            out_hw = (oh, ow) if rescale else crop
            labels = torch.remainder(flat[b], nc)            # index arithmetic on 100 ints (plumbing)
            qidx = torch.div(flat[b], nc, rounding_mode='floor').to(torch.int32).contiguous()
            masks, det, boxes = ops.query_mask_post(low[b].contiguous(), qidx, scores[b].contiguous(), (Hb, Wb), crop, out_hw)
            r = InstanceData(bboxes=boxes, labels=labels.to(torch.long), scores=det, masks=masks)
            r.query_indices = qidx
            results.append(dict(ins_results=r))
        return results


@MODELS.register_module()
class MaskFormerFusionHead(RSMaskFormerFusionHead):
    """The reference's (edited) mmdet MaskFormerFusionHead.predict (maskformer_fusion_head.py:184-270) has the same body
    as RSMaskFormerFusionHead.predict (models.py:662-715): crop by int(ori * scale_factor), resize, instance_postprocess."""


@MODELS.register_module()
class RSPrompterQuery(BaseDetectorHIP):
    def __init__(self, shared_image_embedding, decoder_freeze=True, backbone=None, neck=None, panoptic_head=None,
                 panoptic_fusion_head=None, train_cfg=None, test_cfg=None, data_preprocessor=None, init_cfg=None):
        super().__init__()
        self.data_preprocessor = MODELS.build(data_preprocessor or dict(type='DetDataPreprocessor'))
        self.backbone = MODELS.build(backbone)
        self.neck = MODELS.build(neck) if neck is not None else None
        ph = copy.deepcopy(dict(panoptic_head))             # maskformer.py:33-37
        ph.update(train_cfg=train_cfg, test_cfg=test_cfg)
        self.panoptic_head = MODELS.build(ph)
        pf = copy.deepcopy(dict(panoptic_fusion_head))
        pf.update(test_cfg=test_cfg)
        self.panoptic_fusion_head = MODELS.build(pf)
        self.num_things_classes = self.panoptic_head.num_things_classes
        self.num_stuff_classes = self.panoptic_head.num_stuff_classes
        self.num_classes = self.panoptic_head.num_classes
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.shared_image_embedding = MODELS.build(shared_image_embedding)
        self.decoder_freeze = decoder_freeze
        self.eval()

    def get_image_wide_positional_embeddings(self, size):
        return self.shared_image_embedding.image_wide(size)

    def extract_feat(self, batch_inputs):
        """models.py:217-234."""
        vo = self.backbone(batch_inputs)
        if hasattr(vo, 'hidden_states') and vo.hidden_states is not None:
            emb, hs = vo[0], vo[1]
        elif isinstance(vo, tuple):
            emb, hs = vo[0], vo
        else:
            raise NotImplementedError
        pe = self.get_image_wide_positional_embeddings(size=emb.shape[-1]).expand(emb.shape[0], -1, -1, -1)
        return self.neck(hs), emb, pe

    @torch.no_grad()
    def predict(self, batch_inputs, batch_data_samples, rescale=True):
        """models.py:249-272."""
        x, emb, pe = self.extract_feat(batch_inputs)
        self._last_embeddings = debug.keep(emb)            # parity tests / bench canary only (rsprompter_amd/debug.py)
        cls, masks = self.panoptic_head.predict(x, batch_data_samples, image_embeddings=emb,
                                                image_positional_embeddings=pe)
        self._last_head_out = debug.keep((cls, masks))     # parity tests only (rsprompter_amd/debug.py)
        results = self.panoptic_fusion_head.predict(cls, masks, batch_data_samples, rescale=rescale)
        for s, r in zip(batch_data_samples, results):       # maskformer.py:112-152
            if 'ins_results' in r:
                s.pred_instances = r['ins_results']
        return batch_data_samples

    @torch.no_grad()
    def _forward(self, batch_inputs, batch_data_samples=None):
        """`mode='tensor'`: MaskFormer._forward (maskformer.py:153-170) -> panoptic_head.forward: raw head outputs
        `(cls_pred_list, mask_pred_list)` without post-processing.  The lists hold the LAST decoder stage only: the
        reference's forward also returns the 7 intermediate SAM-decoder results (models.py:432,456-463), which only
        the training losses read and which the single-call inference schedule does not compute.
        Reference quirk (not reproduced): the inherited `_forward` passes extract_feat's 3-tuple to the head as the
        feature pyramid and no image embeddings, so the reference's own tensor mode raises."""
        x, emb, pe = self.extract_feat(batch_inputs)
        cls, mask_pred, trace = self.panoptic_head(x, batch_data_samples, emb, pe)
        return [cls], [mask_pred], [trace['mask_pred_plus']]


@MODELS.register_module()
class SAMSegMask2Former(BaseDetectorHIP):
    """models.py:1247-1274 over Mask2Former / MaskFormer (maskformer.py:18-170): SAM encoder -> RSFPN -> the standard
    Mask2FormerHead -> MaskFormerFusionHead.  No prompt encoder, no SAM mask decoder."""

    def __init__(self, backbone=None, neck=None, panoptic_head=None, panoptic_fusion_head=None, train_cfg=None,
                 test_cfg=None, data_preprocessor=None, init_cfg=None):
        super().__init__()
        self.data_preprocessor = MODELS.build(data_preprocessor or dict(type='DetDataPreprocessor'))
        self.backbone = MODELS.build(backbone)
        self.neck = MODELS.build(neck) if neck is not None else None
        ph = copy.deepcopy(dict(panoptic_head))             # maskformer.py:33-37
        ph.update(train_cfg=train_cfg, test_cfg=test_cfg)
        self.panoptic_head = MODELS.build(ph)
        pf = copy.deepcopy(dict(panoptic_fusion_head))
        pf.update(test_cfg=test_cfg)
        self.panoptic_fusion_head = MODELS.build(pf)
        self.num_things_classes = self.panoptic_head.num_things_classes
        self.num_stuff_classes = self.panoptic_head.num_stuff_classes
        self.num_classes = self.panoptic_head.num_classes
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.eval()

    def extract_feat(self, batch_inputs):
        """models.py:1262-1274: the neck output only."""
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
        """maskformer.py:83-151."""
        x = self.extract_feat(batch_inputs)
        cls, masks = self.panoptic_head.predict(x, batch_data_samples)
        self._last_head_out = debug.keep((cls, masks))
        results = self.panoptic_fusion_head.predict(cls, masks, batch_data_samples, rescale=rescale)
        for s, r in zip(batch_data_samples, results):
            if 'ins_results' in r:
                s.pred_instances = r['ins_results']
        return batch_data_samples

    @torch.no_grad()
    def _forward(self, batch_inputs, batch_data_samples=None):
        """`mode='tensor'` (maskformer.py:153-170): raw `(cls_pred_list, mask_pred_list)`, last decoder stage only."""
        cls, mask_pred, _ = self.panoptic_head(self.extract_feat(batch_inputs), batch_data_samples)
        return [cls], [mask_pred]
