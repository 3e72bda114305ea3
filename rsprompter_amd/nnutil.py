"""Small helpers shared by the facade modules (parameter trees in the
reference's `state_dict` key layout, channels-last plumbing, checkpoints)."""
import os
import re
import warnings

import torch
from torch import nn

SAM_ARCH = {
    # SamVisionConfig values of facebook/sam-vit-{base,large,huge} (Appendix A of SURVEY.md)
    'base': dict(hidden=768, depth=12, heads=12, global_idx=(2, 5, 8, 11), mlp=3072),
    'large': dict(hidden=1024, depth=24, heads=16, global_idx=(5, 11, 17, 23), mlp=4096),
    'huge': dict(hidden=1280, depth=32, heads=16, global_idx=(7, 15, 23, 31), mlp=5120),
}


def infer_sam_arch(name):
    """Same rule as the reference (models.py:1005 / :824): substring of the hub id / path."""
    name = str(name)
    return 'base' if 'base' in name else 'large' if 'large' in name else 'huge'


def add_param(root, dotted, shape=None, init=0.0, buffer=False, tensor=None):
    """Register a parameter/buffer at `a.b.c` creating plain nn.Module containers on the way,
    so that `root.state_dict()` uses exactly the reference's key names."""
    parts = dotted.split('.')
    mod = root
    for p in parts[:-1]:
        if not hasattr(mod, p):
            mod.add_module(p, nn.Module())
        mod = getattr(mod, p)
    if tensor is None:
        tensor = torch.full(tuple(shape), float(init), dtype=torch.float32)
    if buffer:
        mod.register_buffer(parts[-1], tensor)
    else:
        mod.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))
    return tensor


def get_by_name(root, dotted):
    obj = root
    for p in dotted.split('.'):
        obj = getattr(obj, p)
    return obj


def nhwc_view(x):
    """[B, C, H, W] (any strides) -> contiguous [B, H, W, C] tensor; free if the input is
    already channels-last (which is what every module of this package produces)."""
    y = x.permute(0, 2, 3, 1)
    if not y.is_contiguous():
        y = y.contiguous()  # foreign NCHW input: one device copy (plumbing)
    return y


def nchw_view(x_nhwc):
    """contiguous [B, H, W, C] -> logical [B, C, H, W] view (channels_last strides, no copy)."""
    return x_nhwc.permute(0, 3, 1, 2)


def load_checkpoint_into(module, path, revise_keys=(), strict=False, prefix=None):
    """see rsprompter_amd/checkpoint.py (formats: HF .bin / .safetensors (+ sharded index), mmengine .pth, DeepSpeed
    zero_to_fp32 output; ConvModule norm-name aliases)."""
    from .checkpoint import load_checkpoint_into as _load
    return _load(module, path, revise_keys=revise_keys, strict=strict, prefix=prefix)


class HIPModule(nn.Module):
    """Base of the facade modules: tracks when weights must be (re)packed for the kernels."""

    def __init__(self):
        super().__init__()
        self._packed = None

    def _apply(self, fn, *a, **kw):
        self._packed = None
        return super()._apply(fn, *a, **kw)

    def load_state_dict(self, *a, **kw):
        self._packed = None
        for m in self.modules():
            if isinstance(m, HIPModule):
                m._packed = None
        return super().load_state_dict(*a, **kw)

    def _load_from_state_dict(self, *a, **kw):
        self._packed = None
        return super()._load_from_state_dict(*a, **kw)

    def init_weights(self):
        pass
