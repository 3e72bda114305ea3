"""CPU oracle of the 512-px / LoRA encoder variant (tests only): the in-repo ViTSAM
(mmpretrain/models/backbones/vit_sam.py) restated in plain torch with the SAME attribute names
(so `state_dict` keys equal the reference's), wrapped the way peft 0.8.2 wraps it
(`base_model.model.<path>.qkv.{base_layer,lora_A.default,lora_B.default}`).

window_partition / window_unpartition / get_rel_pos / add_decomposed_rel_pos follow
vit_sam.py:17-157 and are pinned against the real file by tests/test_oracle_golden.py.
"""
import torch
import torch.nn.functional as F
from torch import nn

from .hf_sam import ARCH


def window_partition(x, ws):                       # vit_sam.py:17-43
    B, H, W, C = x.shape
    ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
    if ph or pw:
        x = F.pad(x, (0, 0, 0, pw, 0, ph))
    Hp, Wp = H + ph, W + pw
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C), (Hp, Wp)


def window_unpartition(w, ws, pad_hw, hw):         # vit_sam.py:46-75
    Hp, Wp = pad_hw
    H, W = hw
    B = w.shape[0] // (Hp * Wp // ws // ws)
    x = w.view(B, Hp // ws, Wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).contiguous().view(B, Hp, Wp, -1)
    return x[:, :H, :W, :].contiguous()


def get_rel_pos(q_size, k_size, rel_pos):          # vit_sam.py:78-114
    max_rel_dist = int(2 * max(q_size, k_size) - 1)
    if rel_pos.shape[0] != max_rel_dist:
        r = F.interpolate(rel_pos.reshape(1, rel_pos.shape[0], -1).permute(0, 2, 1), size=max_rel_dist, mode='linear')
        r = r.reshape(-1, max_rel_dist).permute(1, 0)
    else:
        r = rel_pos
    q_coords = torch.arange(q_size)[:, None] * max(k_size / q_size, 1.0)
    k_coords = torch.arange(k_size)[None, :] * max(q_size / k_size, 1.0)
    rel = (q_coords - k_coords) + (k_size - 1) * max(q_size / k_size, 1.0)
    return r[rel.long()]


class LoraLinear(nn.Module):
    """peft LoRA Linear in eval mode: base(x) + lora_B(lora_A(x)) * (alpha / r)."""

    def __init__(self, cin, cout, r=16, alpha=32):
        super().__init__()
        self.base_layer = nn.Linear(cin, cout)
        self.lora_A = nn.ModuleDict(dict(default=nn.Linear(cin, r, bias=False)))
        self.lora_B = nn.ModuleDict(dict(default=nn.Linear(r, cout, bias=False)))
        self.scaling = alpha / r

    def forward(self, x):
        return self.base_layer(x) + self.lora_B['default'](self.lora_A['default'](x)) * self.scaling


class Attention(nn.Module):                        # vit_sam.py:160-221
    def __init__(self, dim, heads, input_size, lora):
        super().__init__()
        self.num_heads, self.scale = heads, (dim // heads) ** -0.5
        self.qkv = LoraLinear(dim, 3 * dim) if lora else nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)
        self.rel_pos_h = nn.Parameter(torch.zeros(2 * input_size - 1, dim // heads))
        self.rel_pos_w = nn.Parameter(torch.zeros(2 * input_size - 1, dim // heads))

    def forward(self, x):
        B, H, W, _ = x.shape
        qkv = self.qkv(x).reshape(B, H * W, 3, self.num_heads, -1).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.reshape(3, B * self.num_heads, H * W, -1).unbind(0)
        attn = (q * self.scale) @ k.transpose(-2, -1)
        Rh, Rw = get_rel_pos(H, H, self.rel_pos_h), get_rel_pos(W, W, self.rel_pos_w)
        r_q = q.reshape(B * self.num_heads, H, W, -1)
        rel_h = torch.einsum('bhwc,hkc->bhwk', r_q, Rh)
        rel_w = torch.einsum('bhwc,wkc->bhwk', r_q, Rw)
        attn = (attn.view(-1, H, W, H, W) + rel_h[:, :, :, :, None] + rel_w[:, :, :, None, :]).view(-1, H * W, H * W)
        attn = attn.softmax(dim=-1)
        x = (attn @ v).view(B, self.num_heads, H, W, -1).permute(0, 2, 3, 1, 4).reshape(B, H, W, -1)
        return self.proj(x)


class _FFN(nn.Module):                             # mmcv FFN(num_fcs=2, GELU): x + Linear(act(Linear(x)))
    def __init__(self, dim, hidden):
        super().__init__()
        self.layers = nn.Sequential(nn.Sequential(nn.Linear(dim, hidden), nn.GELU()), nn.Linear(hidden, dim))

    def forward(self, x, identity):
        return identity + self.layers(x)


class Layer(nn.Module):                            # vit_sam.py:224-313
    def __init__(self, dim, heads, mlp, window, grid, lora):
        super().__init__()
        self.window_size = window
        self.ln1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, heads, window if window > 0 else grid, lora)
        self.ln2 = nn.LayerNorm(dim, eps=1e-6)
        self.ffn = _FFN(dim, mlp)

    def forward(self, x):
        shortcut = x
        x = self.ln1(x)
        if self.window_size > 0:
            H, W = x.shape[1], x.shape[2]
            x, pad_hw = window_partition(x, self.window_size)
        x = self.attn(x)
        if self.window_size > 0:
            x = window_unpartition(x, self.window_size, pad_hw, (H, W))
        x = shortcut + x
        return self.ffn(self.ln2(x), identity=x)


class _LN2d(nn.LayerNorm):                         # mmpretrain/models/utils/norm.py:52-90
    def forward(self, x):
        return F.layer_norm(x.permute(0, 2, 3, 1), self.normalized_shape, self.weight, self.bias, self.eps).permute(0, 3, 1, 2)


class _PatchEmbed(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.projection = nn.Conv2d(3, dim, 16, 16)


class ViTSAM(nn.Module):                           # vit_sam.py:317-602 (out_indices=-1, out_channels=256)
    def __init__(self, arch='base', img_size=512, lora=False):
        super().__init__()
        a = ARCH[arch]
        dim, depth, heads = a['hidden_size'], a['num_hidden_layers'], a['num_attention_heads']
        self.grid = img_size // 16
        self.patch_embed = _PatchEmbed(dim)
        self.pos_embed = nn.Parameter(torch.zeros(1, self.grid, self.grid, dim))
        self.layers = nn.ModuleList([
            Layer(dim, heads, dim * 4, 14 if i not in a['global_attn_indexes'] else 0, self.grid, lora)
            for i in range(depth)])
        self.channel_reduction = nn.Sequential(nn.Conv2d(dim, 256, 1, bias=False), _LN2d(256, eps=1e-6),
                                               nn.Conv2d(256, 256, 3, padding=1, bias=False), _LN2d(256, eps=1e-6))

    def forward(self, x):
        x = self.patch_embed.projection(x).permute(0, 2, 3, 1) + self.pos_embed
        for layer in self.layers:
            x = layer(x)
        return (self.channel_reduction(x.permute(0, 3, 1, 2)),)


class PeftWrapped(nn.Module):
    """`get_peft_model(m, cfg)` naming: PeftModel.base_model (LoraModel).model == m."""

    def __init__(self, m):
        super().__init__()
        self.base_model = nn.Module()
        self.base_model.model = m

    def forward(self, x):
        return self.base_model.model(x)
