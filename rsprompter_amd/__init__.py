"""rsprompter_amd -- MI355X-native (gfx950) implementation of RSPrompter's inference hot path:
SAM ViT image encoder -> anchor/query prompt generator -> SAM mask decoder, behind the
reference's own `MODELS` registry + `forward(inputs, data_samples, mode)` API.

Importing the package registers every module (the stand-in for
`custom_imports = dict(imports=['mmdet.rsprompter'])`, configs/rsprompter/_base_/rsprompter_anchor.py:3).
All arithmetic runs in librsp_hip.so (hand-written HIP, C ABI in include/rsp_hip.h); there is no
CPU or PyTorch-eager fallback.
"""
from .registry import MODELS, TASK_UTILS  # noqa: F401
from .config import Config, ConfigDict  # noqa: F401
from .structures import DetDataSample, InstanceData  # noqa: F401
from . import sam_encoder, sam_decoder, necks, anchor_heads, detectors, query_heads, samdet  # noqa: F401  (registration)

__version__ = '0.1.0'


def build_model(cfg):
    """cfg: a Config, a dict with a `model` key, or the model dict itself."""
    model_cfg = cfg['model'] if 'model' in cfg and 'type' not in cfg else cfg
    return MODELS.build(model_cfg)
