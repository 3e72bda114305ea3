"""Seeded synthetic weights and inputs: there is no SAM checkpoint
and no dataset in the build/bench environment, so parity and throughput runs use
deterministic random tensors.  Each tensor's values depend only on (seed, key name,
shape), so the oracle and the HIP modules get identical weights through the ordinary
`state_dict` API irrespective of module construction order.
"""
import hashlib
import math

import torch


def _gen(seed, name):
    h = hashlib.sha256(f'{seed}:{name}'.encode()).digest()
    g = torch.Generator(device='cpu')
    g.manual_seed(int.from_bytes(h[:8], 'little') & 0x7fffffffffffffff)
    return g


def synth_tensor(name, ref, seed=0):
    g = _gen(seed, name)
    shape = tuple(ref.shape)
    leaf = name.split('.')[-1]
    if leaf == 'num_batches_tracked':
        return torch.zeros(shape, dtype=ref.dtype)
    if leaf == 'running_mean':
        return torch.randn(shape, generator=g) * 0.1
    if leaf == 'running_var':
        return torch.rand(shape, generator=g) + 0.5
    if leaf in ('rel_pos_h', 'rel_pos_w'):
        return torch.randn(shape, generator=g) * 0.05
    if leaf == 'pos_embed':
        return torch.randn(shape, generator=g) * 0.02
    if leaf == 'positional_embedding':
        return torch.randn(shape, generator=g)
    if 'lora_A' in name or 'lora_B' in name:
        return torch.randn(shape, generator=g) * 0.02
    if leaf == 'bias':
        return torch.randn(shape, generator=g) * 0.05
    if leaf == 'weight' and len(shape) == 1:      # LayerNorm / BatchNorm / LN2d scale
        return 1.0 + torch.randn(shape, generator=g) * 0.02
    parent = name.split('.')[-2] if '.' in name else ''
    if parent in ('iou_token', 'mask_tokens', 'no_mask_embed', 'query_embed', 'query_feat', 'level_embed'):
        return torch.randn(shape, generator=g) * 0.5      # nn.Embedding tables
    if len(shape) >= 2:                            # Linear / Conv / ConvTranspose
        if 'upscale_conv' in name or '.fpn1.' in name or '.fpn2.' in name:
            fan_in = shape[0]                      # ConvTranspose2d weight is [Cin, Cout, k, k]
        else:
            fan_in = int(math.prod(shape[1:]))
        return torch.randn(shape, generator=g) / math.sqrt(max(fan_in, 1))
    return torch.randn(shape, generator=g) * 0.02


def synth_state_dict(module, seed=0):
    sd = {}
    for k, v in module.state_dict().items():
        sd[k] = synth_tensor(k, v, seed).to(v.dtype)
    return sd


def synth_images(n, size=(1024, 1024), seed=1234):
    """uint8 [3,H,W] tiles, the recipe of mmdet/testing/_utils.py:135 (seeded randint)."""
    out = []
    for i in range(n):
        g = torch.Generator(device='cpu')
        g.manual_seed(seed + i)
        out.append(torch.randint(0, 256, (3, size[0], size[1]), generator=g, dtype=torch.uint8))
    return out


def synth_metas(n, size=(1024, 1024), ori_shape=None, scale_factor=(1.0, 1.0)):
    ori = tuple(ori_shape) if ori_shape is not None else tuple(size)
    return [dict(img_shape=tuple(size), ori_shape=ori, pad_shape=tuple(size),
                 batch_input_shape=tuple(size), scale_factor=tuple(scale_factor), img_id=i)
            for i in range(n)]
