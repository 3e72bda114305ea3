"""Torch stand-ins for the mmcv / mmengine building blocks the reference's forwards are wired from (TEST INFRASTRUCTURE,
used only by tests/golden/make_golden_forwards.py in the build container).

mmcv / mmengine are not installable here and their sources are not under /root/reference, so the golden generator
injects these stand-ins and then runs the REAL reference classes (`RSSimpleFPN`, `RSPrompterAnchorMaskHead`,
`RSMask2FormerHead`, `MSDeformAttnPixelDecoder`, `Mask2FormerTransformerDecoderLayer`, `ViTSAM`, ...) on top of them.
What that pins is everything the reference itself wrote: module wiring, `state_dict` key layout, tensor permutes,
level ordering, residual / norm order, the attention-mask rule.  The stand-ins follow mmcv 2.x's documented behaviour
(SURVEY.md App. B); the two genuine ops -- `MultiScaleDeformableAttention` and `MultiheadAttention` -- are thin
adapters giving the ORACLE's leaf classes (oracle/query.py) mmcv's constructor / forward signatures, so the leaf
arithmetic is not restated a second time (it is checked by the known-answer tests in tests/test_oracle_golden.py).
"""
import torch
from torch import nn


# ----------------------------------------------------------------------------- mmcv.cnn
def infer_abbr(cls):
    """mmcv/cnn/bricks/norm.py::infer_abbr: `_abbr_`, else by base class, else by class name, else 'norm_layer'."""
    if hasattr(cls, '_abbr_'):
        return cls._abbr_
    if issubclass(cls, nn.modules.instancenorm._InstanceNorm):
        return 'in'
    if issubclass(cls, nn.modules.batchnorm._BatchNorm):
        return 'bn'
    if issubclass(cls, nn.GroupNorm):
        return 'gn'
    if issubclass(cls, nn.LayerNorm):
        return 'ln'
    name = cls.__name__.lower()
    for key, ab in (('batch', 'bn'), ('group', 'gn'), ('layer', 'ln'), ('instance', 'in')):
        if key in name:
            return ab
    return 'norm_layer'


class Standins:
    """factory bound to a registry (norm / activation types are looked up there, like mmcv does in MODELS)."""

    def __init__(self, registry):
        self.registry = registry
        self.norms = {'BN': nn.BatchNorm2d, 'BN2d': nn.BatchNorm2d, 'GN': nn.GroupNorm, 'LN': nn.LayerNorm}
        self.acts = {'ReLU': nn.ReLU, 'GELU': nn.GELU}

    def build_norm_layer(self, cfg, num_features, postfix=''):
        """mmcv.cnn.build_norm_layer -> (name, layer)."""
        cfg_ = dict(cfg)
        t = cfg_.pop('type')
        cls = self.norms.get(t) or self.registry.get(t)
        assert cls is not None, t
        name = infer_abbr(cls) + str(postfix)
        requires_grad = cfg_.pop('requires_grad', True)
        cfg_.setdefault('eps', 1e-5)
        if cls is nn.GroupNorm:
            layer = cls(num_channels=num_features, **cfg_)
        else:
            layer = cls(num_features, **cfg_)
        for p in layer.parameters():
            p.requires_grad = requires_grad
        return name, layer

    def build_activation_layer(self, cfg):
        cfg_ = dict(cfg)
        return self.acts[cfg_.pop('type')](**cfg_)

    def make(self):
        S = self

        class ConvModule(nn.Module):
            """mmcv.cnn.ConvModule: conv -> norm -> act; bias='auto' = no bias iff a norm follows; norm registered under
            its abbreviation (`bn`, `gn`, `ln`, or `norm_layer` for classes mmcv cannot abbreviate, e.g. LN2d)."""

            def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                         bias='auto', conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'), inplace=True,
                         with_spectral_norm=False, padding_mode='zeros', order=('conv', 'norm', 'act')):
                super().__init__()
                assert conv_cfg is None and order == ('conv', 'norm', 'act') and not with_spectral_norm
                self.with_norm, self.with_activation = norm_cfg is not None, act_cfg is not None
                if bias == 'auto':
                    bias = not self.with_norm
                self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
                if self.with_norm:
                    self.norm_name, norm = S.build_norm_layer(norm_cfg, out_channels)
                    self.add_module(self.norm_name, norm)
                if self.with_activation:
                    a = dict(act_cfg)
                    if a['type'] not in ('Tanh', 'PReLU', 'Sigmoid', 'HSigmoid', 'Swish', 'GELU'):
                        a.setdefault('inplace', inplace)
                    self.activate = S.build_activation_layer(a)

            @property
            def norm(self):
                return getattr(self, self.norm_name) if self.with_norm else None

            def forward(self, x, activate=True, norm=True):
                x = self.conv(x)
                if norm and self.with_norm:
                    x = self.norm(x)
                if activate and self.with_activation:
                    x = self.activate(x)
                return x

        class FFN(nn.Module):
            """mmcv.cnn.bricks.transformer.FFN: layers = Sequential(Sequential(Linear, act, Dropout) x (num_fcs-1),
            Linear, Dropout); forward(x, identity=None) = (identity or x) + dropout_layer(layers(x))."""

            def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2,
                         act_cfg=dict(type='ReLU', inplace=True), ffn_drop=0., dropout_layer=None, add_identity=True,
                         init_cfg=None, layer_scale_init_value=0.):
                super().__init__()
                assert num_fcs >= 2 and layer_scale_init_value == 0.
                self.embed_dims, self.feedforward_channels, self.num_fcs = embed_dims, feedforward_channels, num_fcs
                layers, cin = [], embed_dims
                for _ in range(num_fcs - 1):
                    layers.append(nn.Sequential(nn.Linear(cin, feedforward_channels), S.build_activation_layer(act_cfg),
                                                nn.Dropout(ffn_drop)))
                    cin = feedforward_channels
                layers.append(nn.Linear(feedforward_channels, embed_dims))
                layers.append(nn.Dropout(ffn_drop))
                self.layers = nn.Sequential(*layers)
                self.add_identity = add_identity

            def forward(self, x, identity=None):
                out = self.layers(x)
                if not self.add_identity:
                    return out
                return (x if identity is None else identity) + out

        class PatchEmbed(nn.Module):
            """mmcv.cnn.bricks.transformer.PatchEmbed (conv_type='Conv2d', padding='corner' is a no-op when the input is
            a multiple of the patch): projection conv -> flatten(2).transpose(1, 2); returns (x [B, N, C], (H, W))."""

            def __init__(self, in_channels=3, embed_dims=768, conv_type='Conv2d', kernel_size=16, stride=16,
                         padding='corner', dilation=1, bias=True, norm_cfg=None, input_size=None, init_cfg=None):
                super().__init__()
                assert norm_cfg is None
                self.embed_dims = embed_dims
                self.projection = nn.Conv2d(in_channels, embed_dims, kernel_size, stride, 0, dilation, bias=bias)
                self.norm = None
                if input_size is not None:
                    s = (input_size, input_size) if isinstance(input_size, int) else tuple(input_size)
                    self.init_input_size = s
                    self.init_out_size = ((s[0] - kernel_size) // stride + 1, (s[1] - kernel_size) // stride + 1)

            def forward(self, x):
                assert x.shape[-2] % self.projection.stride[0] == 0 and x.shape[-1] % self.projection.stride[1] == 0
                x = self.projection(x)
                out_size = (x.shape[2], x.shape[3])
                return x.flatten(2).transpose(1, 2), out_size

        from oracle.query import MHA, MSDeformAttn

        class MultiheadAttention(MHA):
            """mmcv MultiheadAttention(batch_first=True), signature adapter over oracle.query.MHA."""

            def __init__(self, embed_dims, num_heads, attn_drop=0., proj_drop=0., dropout_layer=None, init_cfg=None,
                         batch_first=False, dropout=None, **kwargs):
                super().__init__(embed_dims, num_heads)
                assert batch_first and not attn_drop and not proj_drop and not dropout
                self.embed_dims, self.num_heads, self.batch_first = embed_dims, num_heads, batch_first

            def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_pos=None, attn_mask=None,
                        key_padding_mask=None, **kwargs):
                if key is None:
                    key = query
                if value is None:
                    value = key
                if key_pos is None and query_pos is not None and query_pos.shape == key.shape:
                    key_pos = query_pos
                assert identity is None and key_padding_mask is None
                zq = torch.zeros_like(query) if query_pos is None else query_pos
                zk = torch.zeros_like(key) if key_pos is None else key_pos
                return super().forward(query, key, value, zq, zk, attn_mask)

        class MultiScaleDeformableAttention(MSDeformAttn):
            """mmcv.ops MultiScaleDeformableAttention(batch_first=True), signature adapter over oracle.query.MSDeformAttn."""

            def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, im2col_step=64, dropout=0.1,
                         batch_first=False, norm_cfg=None, init_cfg=None, value_proj_ratio=1.0):
                super().__init__(embed_dims, num_heads, num_levels, num_points)
                assert batch_first and not dropout and value_proj_ratio == 1.0
                self.embed_dims, self.batch_first = embed_dims, batch_first

            def init_weights(self):
                pass

            def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_padding_mask=None,
                        reference_points=None, spatial_shapes=None, level_start_index=None, **kwargs):
                assert (value is None or value is query) and identity is None      # mmcv: value defaults to query
                assert key_padding_mask is None or not bool(key_padding_mask.any())
                zq = torch.zeros_like(query) if query_pos is None else query_pos
                return super().forward(query, zq, reference_points, spatial_shapes)

        return dict(ConvModule=ConvModule, FFN=FFN, PatchEmbed=PatchEmbed, MultiheadAttention=MultiheadAttention,
                    MultiScaleDeformableAttention=MultiScaleDeformableAttention, Conv2d=nn.Conv2d,
                    build_norm_layer=self.build_norm_layer, build_activation_layer=self.build_activation_layer)
