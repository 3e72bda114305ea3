"""Neck of RSPrompter on HIP kernels: RSFPN = feature aggregator + RSSimpleFPN.

Reference: mmdet/rsprompter/models.py  RSFPN :917-940, PseudoFeatureAggregator :943-984,
RSFeatureAggregator :987-1057, RSSimpleFPN :1277-1363, LN2d :32-50.  Same registry names,
ctor kwargs and `state_dict` keys.  Everything runs channels-last: a feature map is a
[B*H*W, C] matrix, 1x1 convs are GEMMs, 3x3 convs are implicit GEMMs, LN2d is a row
LayerNorm, eval-mode BatchNorm is folded into the conv weights at pack time.
"""
import torch

from . import ops
from .nnutil import HIPModule, add_param, infer_sam_arch, nchw_view, nhwc_view
from .registry import MODELS


@MODELS.register_module(force=True)
class LN2d(HIPModule):
    """models.py:32-50; on channels-last data this is `rsp_layernorm` over the last dim."""

    def __init__(self, normalized_shape, eps=1e-6, requires_grad=True):
        super().__init__()
        add_param(self, 'weight', (normalized_shape,), 1.0)
        add_param(self, 'bias', (normalized_shape,))
        self.eps = eps

    def forward(self, x):
        y = ops.layernorm(nhwc_view(x), self.weight, self.bias, self.eps)
        return nchw_view(y)


def fold_bn(w, b, bn, eps=1e-5):
    """eval-mode BatchNorm2d folded into the preceding conv (models.py:1013-1018, 1026-1028)."""
    scale = bn.weight.detach() / torch.sqrt(bn.running_var.detach() + eps)
    w2 = w.detach() * scale.view(-1, *([1] * (w.dim() - 1)))
    b0 = b.detach() if b is not None else torch.zeros_like(scale)
    b2 = (b0 - bn.running_mean.detach()) * scale + bn.bias.detach()
    return w2, b2


def conv3x3_weight(w):
    """[O, I, 3, 3] -> [O, (ky, kx, I)] (K order of the NHWC implicit GEMM)."""
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)


def _add_conv(root, name, cout, cin, k, bias=True):
    add_param(root, name + '.weight', (cout, cin, k, k))
    if bias:
        add_param(root, name + '.bias', (cout,))


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
class RSFeatureAggregator(HIPModule):
    in_channels_dict = {'base': [768] * 13, 'large': [1024] * 25, 'huge': [1280] * 33}

    def __init__(self, in_channels, hidden_channels=64, out_channels=256, select_layers=range(1, 12, 2),
                 init_cfg=None):
        super().__init__()
        assert isinstance(in_channels, str)
        self.in_channels = self.in_channels_dict[infer_sam_arch(in_channels)]
        self.select_layers = list(select_layers)
        self.hidden_channels, self.out_channels = hidden_channels, out_channels
        h = hidden_channels
        for i, l in enumerate(self.select_layers):
            _add_conv(self, f'downconvs.{i}.0', h, self.in_channels[l], 1)
            _add_bn(self, f'downconvs.{i}.1', h)
            _add_conv(self, f'downconvs.{i}.3', h, h, 3)
            _add_bn(self, f'downconvs.{i}.4', h)
            _add_conv(self, f'hidden_convs.{i}.0', h, h, 3)
            _add_bn(self, f'hidden_convs.{i}.1', h)
        _add_conv(self, 'fusion_conv.0', out_channels, h, 1)
        _add_bn(self, 'fusion_conv.1', out_channels)
        _add_conv(self, 'fusion_conv.3', out_channels, out_channels, 3)
        _add_bn(self, 'fusion_conv.4', out_channels)
        _add_conv(self, 'fusion_conv.6', out_channels, out_channels, 3)

    def _pack(self):
        P = dict(down=[], hid=[])
        for i in range(len(self.select_layers)):
            c0, b0 = _g(self, f'downconvs.{i}.0'), _g(self, f'downconvs.{i}.1')
            c3, b3 = _g(self, f'downconvs.{i}.3'), _g(self, f'downconvs.{i}.4')
            ch, bh = _g(self, f'hidden_convs.{i}.0'), _g(self, f'hidden_convs.{i}.1')
            w, b = fold_bn(c0.weight, c0.bias, b0)
            d0 = ops.PackedWeight(w.reshape(w.shape[0], -1), b)
            w, b = fold_bn(c3.weight, c3.bias, b3)
            d3 = ops.PackedWeight(conv3x3_weight(w), b)
            w, b = fold_bn(ch.weight, ch.bias, bh)
            P['down'].append((d0, d3))
            P['hid'].append(ops.PackedWeight(conv3x3_weight(w), b))
        f0, f1, f3, f4, f6 = (_g(self, f'fusion_conv.{j}') for j in (0, 1, 3, 4, 6))
        w, b = fold_bn(f0.weight, f0.bias, f1)
        P['f0'] = ops.PackedWeight(w.reshape(w.shape[0], -1), b)
        w, b = fold_bn(f3.weight, f3.bias, f4)
        P['f3'] = ops.PackedWeight(conv3x3_weight(w), b)
        P['f6'] = ops.PackedWeight(conv3x3_weight(f6.weight.detach()), f6.bias)
        self._packed = P

    def forward(self, inputs):
        assert len(inputs) == len(self.in_channels)
        if self._packed is None:
            self._pack()
        P = self._packed
        B, H, W, _ = inputs[0].shape
        hc = self.hidden_channels
        x = None
        for i, l in enumerate(self.select_layers):
            hs = inputs[l]
            if not hs.is_contiguous():
                hs = hs.contiguous()
            d0, d3 = P['down'][i]
            f = ops.gemm(hs.view(B * H * W, -1), d0, act=ops.ACT_RELU)
            # hidden_state = x + relu(bn(conv3x3(f)))  (models.py:1050-1053)
            hsx = ops.gemm(f.view(B, H, W, hc), d3, act=ops.ACT_RELU, conv=(3, 1, 1), res=x)
            # x = hidden_state + relu(bn(conv3x3(hidden_state)))  (models.py:1054-1055)
            x = ops.gemm(hsx.view(B, H, W, hc), P['hid'][i], act=ops.ACT_RELU, conv=(3, 1, 1), res=hsx)
        y = ops.gemm(x, P['f0'], act=ops.ACT_RELU)
        y = ops.gemm(y.view(B, H, W, self.out_channels), P['f3'], act=ops.ACT_RELU, conv=(3, 1, 1))
        y = ops.gemm(y.view(B, H, W, self.out_channels), P['f6'], conv=(3, 1, 1))
        return nchw_view(y.view(B, H, W, self.out_channels))


@MODELS.register_module()
class PseudoFeatureAggregator(HIPModule):
    """models.py:943-984 (512-px / LoRA configs): 1x1 -> LN -> 3x3 -> LN -> 3x3 -> LN, bias-free."""

    def __init__(self, in_channels, hidden_channels=64, out_channels=256, init_cfg=None):
        super().__init__()
        self.c = (in_channels, hidden_channels, out_channels)
        _add_conv(self, 'channel_fusion.0', hidden_channels, in_channels, 1, bias=False)
        add_param(self, 'channel_fusion.1.weight', (hidden_channels,), 1.0)
        add_param(self, 'channel_fusion.1.bias', (hidden_channels,))
        _add_conv(self, 'channel_fusion.2', hidden_channels, hidden_channels, 3, bias=False)
        add_param(self, 'channel_fusion.3.weight', (hidden_channels,), 1.0)
        add_param(self, 'channel_fusion.3.bias', (hidden_channels,))
        _add_conv(self, 'channel_fusion.4', out_channels, hidden_channels, 3, bias=False)
        add_param(self, 'channel_fusion.5.weight', (out_channels,), 1.0)
        add_param(self, 'channel_fusion.5.bias', (out_channels,))

    def _pack(self):
        cf = self.channel_fusion
        w0 = getattr(cf, '0').weight.detach()
        self._packed = dict(
            c0=ops.PackedWeight(w0.reshape(w0.shape[0], -1)),
            c2=ops.PackedWeight(conv3x3_weight(getattr(cf, '2').weight.detach())),
            c4=ops.PackedWeight(conv3x3_weight(getattr(cf, '4').weight.detach())))

    def forward(self, inputs):
        assert len(inputs) == 1
        if self._packed is None:
            self._pack()
        P, cf = self._packed, self.channel_fusion
        x = nhwc_view(inputs[0])
        B, H, W, _ = x.shape
        _, ch, co = self.c
        y = ops.gemm(x.view(B * H * W, -1), P['c0'], bias=None)
        y = ops.layernorm(y, getattr(cf, '1').weight, getattr(cf, '1').bias, 1e-6)
        y = ops.gemm(y.view(B, H, W, ch), P['c2'], bias=None, conv=(3, 1, 1))
        y = ops.layernorm(y, getattr(cf, '3').weight, getattr(cf, '3').bias, 1e-6)
        y = ops.gemm(y.view(B, H, W, ch), P['c4'], bias=None, conv=(3, 1, 1))
        y = ops.layernorm(y, getattr(cf, '5').weight, getattr(cf, '5').bias, 1e-6)
        return nchw_view(y.view(B, H, W, co))


def convt_weights(w, b):
    """ConvTranspose2d(k2,s2) weight [Cin, Cout, 2, 2] -> per-dy GEMM weights [(dx, co), Cin]."""
    wt = w.detach().permute(2, 3, 1, 0)  # [dy, dx, co, ci]
    cin = w.shape[0]
    packed = tuple(ops.PackedWeight(wt[dy].reshape(-1, cin)) for dy in (0, 1))
    bias2 = None if b is None else b.detach().repeat(2).contiguous()
    return packed, bias2


def convt_weights4(w, b):
    """ConvTranspose2d(k2,s2) weight [Cin, Cout, 2, 2] -> ONE GEMM weight [(dy, dx, co), Cin] + bias tiled x4."""
    cin = w.shape[0]
    packed = ops.PackedWeight(w.detach().permute(2, 3, 1, 0).reshape(-1, cin))
    bias4 = None if b is None else b.detach().repeat(4).contiguous()
    return packed, bias4


@MODELS.register_module()
class RSSimpleFPN(HIPModule):
    def __init__(self, backbone_channel, in_channels, out_channels, num_outs, conv_cfg=None, norm_cfg=None,
                 act_cfg=None, init_cfg=None):
        super().__init__()
        assert isinstance(in_channels, (list, tuple))
        c = backbone_channel
        # every norm here is built by mmcv `build_norm_layer(norm_cfg, ...)` (models.py:1299, ConvModule :1311-1327), which
        # fills in `eps=1e-5` when the cfg has none (mmcv/cnn/bricks/norm.py: `cfg_.setdefault('eps', 1e-5)`) -- so these
        # LN2d layers do NOT run with LN2d's own default of 1e-6 (models.py:38).  SURVEY App. D-style quirk: reproduced.
        self.norm_eps = float((norm_cfg or {}).get('eps', 1e-5))
        self.backbone_channel, self.in_channels = c, list(in_channels)
        self.out_channels, self.num_ins, self.num_outs = out_channels, len(in_channels), num_outs
        add_param(self, 'fpn1.0.weight', (c, c // 2, 2, 2))
        add_param(self, 'fpn1.0.bias', (c // 2,))
        add_param(self, 'fpn1.1.weight', (c // 2,), 1.0)
        add_param(self, 'fpn1.1.bias', (c // 2,))
        add_param(self, 'fpn1.3.weight', (c // 2, c // 4, 2, 2))
        add_param(self, 'fpn1.3.bias', (c // 4,))
        add_param(self, 'fpn2.0.weight', (c, c // 2, 2, 2))
        add_param(self, 'fpn2.0.bias', (c // 2,))
        for i, ci in enumerate(self.in_channels):
            add_param(self, f'lateral_convs.{i}.conv.weight', (out_channels, ci, 1, 1))
            add_param(self, f'lateral_convs.{i}.norm_layer.weight', (out_channels,), 1.0)
            add_param(self, f'lateral_convs.{i}.norm_layer.bias', (out_channels,))
            add_param(self, f'fpn_convs.{i}.conv.weight', (out_channels, out_channels, 3, 3))
            add_param(self, f'fpn_convs.{i}.norm_layer.weight', (out_channels,), 1.0)
            add_param(self, f'fpn_convs.{i}.norm_layer.bias', (out_channels,))

    def _pack(self):
        P = {}
        P['t1a'] = convt_weights(_g(self, 'fpn1.0').weight, _g(self, 'fpn1.0').bias)
        P['t1b'] = convt_weights(_g(self, 'fpn1.3').weight, _g(self, 'fpn1.3').bias)
        P['t2'] = convt_weights(_g(self, 'fpn2.0').weight, _g(self, 'fpn2.0').bias)
        P['lat'], P['out'] = [], []
        for i in range(self.num_ins):
            lw = _g(self, f'lateral_convs.{i}.conv').weight.detach()
            P['lat'].append(ops.PackedWeight(lw.reshape(lw.shape[0], -1)))
            P['out'].append(ops.PackedWeight(conv3x3_weight(_g(self, f'fpn_convs.{i}.conv').weight.detach())))
        self._packed = P

    def forward(self, input):
        if self._packed is None:
            self._pack()
        P = self._packed
        x = nhwc_view(input)
        B, H, W, C = x.shape
        ln = _g(self, 'fpn1.1')
        # intermediates that only feed GEMMs travel as fp16 planes (no fp32 copy, no separate split pass)
        xp = ops.to_planes(x.contiguous())
        t = ops.conv_transpose2x2(xp, *P['t1a'])
        t = ops.layernorm(t, ln.weight, ln.bias, self.norm_eps, act=ops.ACT_GELU, planes=True, f32=False)   # models.py:1299-1300
        f1 = ops.conv_transpose2x2(t, *P['t1b'], out_planes=True)
        f2 = ops.conv_transpose2x2(xp, *P['t2'], out_planes=True)
        f4 = ops.pool2(x, 0)
        ins = [f1, f2, xp, f4]
        outs = []
        for i in range(self.num_ins):
            xi = ins[i]
            b, h, w, ci = xi.shape
            ll, lo = _g(self, f'lateral_convs.{i}.norm_layer'), _g(self, f'fpn_convs.{i}.norm_layer')
            y = ops.gemm(xi.view(b * h * w, ci), P['lat'][i], bias=None)
            y = ops.layernorm(y, ll.weight, ll.bias, self.norm_eps, planes=True, f32=False)
            y = ops.gemm(y.view(b, h, w, self.out_channels), P['out'][i], bias=None, conv=(3, 1, 1))
            y = ops.layernorm(y, lo.weight, lo.bias, self.norm_eps)
            outs.append(y.view(b, h, w, self.out_channels))
        for _ in range(self.num_outs - self.num_ins):
            outs.append(ops.pool2(outs[-1], 1))   # F.max_pool2d(x, 1, stride=2) == subsampling (models.py:1362)
        return tuple(nchw_view(o) for o in outs)


@MODELS.register_module()
class RSFPN(HIPModule):
    def __init__(self, feature_aggregator=None, feature_spliter=None, init_cfg=None):
        super().__init__()
        if feature_aggregator is not None:
            self.feature_aggregator = MODELS.build(feature_aggregator)
        if feature_spliter is not None:
            self.feature_spliter = MODELS.build(feature_spliter)

    def forward(self, inputs):
        x = self.feature_aggregator(inputs) if hasattr(self, 'feature_aggregator') else inputs
        x = self.feature_spliter(x) if hasattr(self, 'feature_spliter') else (x,)
        return x
