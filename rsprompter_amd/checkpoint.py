"""Checkpoint formats the inference path has to accept.

Reference call sites: the `init_cfg=dict(type='Pretrained', checkpoint=...)` loads of the SAM pieces through
`mmengine.runner.checkpoint.load_checkpoint(..., revise_keys=...)` (mmdet/rsprompter/models.py:777-783, 836-852),
`tools/test.py <config> <checkpoint>` for a trained RSPrompter `.pth`, and README.md:330-335 (DeepSpeed runs are
converted with `zero_to_fp32.py` before evaluation).  What arrives on disk:

  * HuggingFace SAM weights: `pytorch_model.bin` (pickle) or `model.safetensors`, a flat state_dict whose vision keys
    carry the `vision_encoder.` prefix (stripped by the reference's revise_keys), possibly sharded with an
    `*.index.json` (`weight_map`: key -> shard file);
  * mmengine checkpoints: dict(meta=..., state_dict=..., [optimizer, message_hub, ...]);
  * `zero_to_fp32.py` output: a flat fp32 state_dict, keys possibly prefixed `module.` (DeepSpeed engine wrapper) --
    the same rewrite mmengine applies by default (`revise_keys=[(r'^module\\.', '')]`);
  * a directory holding one of the above.
Norm sub-modules of mmcv `ConvModule`s are named after the norm class (`norm_layer` for the reference's own LN2d,
`ln` / `gn` / `bn` for torch norms); a checkpoint written with the other spelling of the SAME layer is accepted
(SURVEY.md App. C caveat): `...{lateral,fpn}_convs.N.ln.*` <-> `...norm_layer.*`.
"""
import json
import os
import re
import warnings

import torch

_WEIGHT_FILES = ('model.safetensors', 'pytorch_model.bin', 'model.safetensors.index.json',
                 'pytorch_model.bin.index.json')


def _trusted_default():
    return os.environ.get('RSP_TRUSTED_CHECKPOINTS', '0') == '1'


def _harmless_globals():
    """Reconstructors that reference-trained mmengine `.pth` files carry in their `meta` / `message_hub` blocks and that
    build plain data only (numpy scalars / arrays / dtypes, OrderedDict, mmengine's HistoryBuffer when mmengine is
    installed): allow-listed for the restricted unpickler, so the upstream checkpoint format loads without the full
    unpickler.  Anything else in a file still needs the explicit opt-in."""
    import collections
    allow = [collections.OrderedDict, collections.defaultdict]
    try:
        import numpy as np
        allow += [np.dtype, np.ndarray, np.float64, np.float32, np.int64, np.int32, np.bool_]
        for modname in ('numpy.core.multiarray', 'numpy._core.multiarray'):
            try:
                mod = __import__(modname, fromlist=['scalar'])
                allow += [mod.scalar, mod._reconstruct]
                break
            except Exception:
                continue
        allow += [type(np.dtype('float32')), type(np.dtype('float64')), type(np.dtype('int64')), type(np.dtype('int32'))]
    except Exception:
        pass
    try:
        from mmengine.logging.history_buffer import HistoryBuffer
        allow.append(HistoryBuffer)
    except Exception:
        pass
    return allow


def _read_file(path, trusted=None):
    """One weight file -> the object stored in it.  Pickle files are read with torch's restricted unpickler
    (`weights_only=True`: tensors, containers, numbers, strings), retried once with an allow-list of harmless numpy /
    collections / mmengine reconstructors (`_harmless_globals`) -- the objects reference-trained mmengine `.pth` files keep
    next to their state_dict.  A file that pickles anything else is refused; the full unpickler -- which executes code
    chosen by whoever wrote the file -- is used only on an explicit opt-in (`trusted=True` or RSP_TRUSTED_CHECKPOINTS=1).
    I/O and corruption errors are never retried."""
    import pickle
    if path.endswith('.safetensors'):
        from safetensors.torch import load_file
        return load_file(path, device='cpu')
    try:
        return torch.load(path, map_location='cpu', weights_only=True)
    except pickle.UnpicklingError as e0:
        try:
            with torch.serialization.safe_globals(_harmless_globals()):
                return torch.load(path, map_location='cpu', weights_only=True)
        except pickle.UnpicklingError:
            pass
        e = e0
        if trusted is None:
            trusted = _trusted_default()
        if not trusted:
            raise RuntimeError(
                f'{path}: the restricted (weights-only) loader refused this checkpoint ({str(e).splitlines()[0]}). '
                'It pickles objects other than tensors / plain containers; loading it runs code from the file. If you '
                'trust its origin pass trusted=True (read_state_dict / load_checkpoint) or set RSP_TRUSTED_CHECKPOINTS=1; '
                'prefer re-saving it as safetensors.') from e
        return torch.load(path, map_location='cpu', weights_only=False)


def read_state_dict(path, trusted=None):
    """file / index json / directory -> flat {key: tensor} (wrappers and DeepSpeed's `module.` prefix removed).
    `trusted`: see _read_file (None = the RSP_TRUSTED_CHECKPOINTS environment switch, default off)."""
    path = os.path.expanduser(str(path))
    if os.path.isdir(path):
        for name in _WEIGHT_FILES:
            if os.path.exists(os.path.join(path, name)):
                path = os.path.join(path, name)
                break
        else:
            raise FileNotFoundError(f'no weight file ({", ".join(_WEIGHT_FILES)}) under {path}')
    if path.endswith('.index.json'):
        with open(path) as f:
            shards = sorted(set(json.load(f)['weight_map'].values()))
        sd = {}
        for s in shards:
            sd.update(_read_file(os.path.join(os.path.dirname(path), s), trusted))
    else:
        sd = _read_file(path, trusted)
    for key in ('state_dict', 'model', 'module'):      # mmengine / lightning / DeepSpeed engine wrappers
        if isinstance(sd, dict) and key in sd and isinstance(sd[key], dict) and \
                any(isinstance(v, torch.Tensor) for v in sd[key].values()):
            sd = sd[key]
            break
    sd = {k: v for k, v in sd.items() if isinstance(v, torch.Tensor)}
    return {re.sub(r'^module\.', '', k): v for k, v in sd.items()}


_NORM_ALIASES = ('norm_layer', 'ln', 'gn', 'bn')


def _alias_candidates(key):
    """other spellings of a ConvModule norm key: a.b.<norm>.weight with <norm> in the alias set."""
    parts = key.split('.')
    if len(parts) >= 2 and parts[-2] in _NORM_ALIASES:
        for alt in _NORM_ALIASES:
            if alt != parts[-2]:
                yield '.'.join(parts[:-2] + [alt, parts[-1]])


def remap_to(own_keys, sd):
    """keep the entries the module owns; resolve ConvModule norm-name aliases for the ones it spells differently."""
    own = set(own_keys)
    out, unused = {}, []
    for k, v in sd.items():
        if k in own:
            out[k] = v
            continue
        hit = next((a for a in _alias_candidates(k) if a in own and a not in sd), None)
        if hit is not None:
            out[hit] = v
        else:
            unused.append(k)
    return out, unused


def load_checkpoint_into(module, path, revise_keys=(), strict=False, prefix=None, trusted=None):
    """mmengine.load_checkpoint stand-in: read any of the formats above, apply the regex key rewrites, keep what the
    module owns (after alias resolution) and load it.  Returns False (with a warning) when the file is missing, like
    the rest of the loaders: the synthetic-weight tests and benches run without the SAM files on disk."""
    if path is None:
        return False
    path = os.path.expanduser(str(path))
    if not os.path.exists(path):
        warnings.warn(f'checkpoint {path} not found; keeping current weights')
        return False
    sd = read_state_dict(path, trusted)
    out = {}
    for k, v in sd.items():
        for pat, rep in revise_keys:
            k = re.sub(pat, rep, k)
        out[k] = v
    if prefix is not None:
        out = {k[len(prefix):]: v for k, v in out.items() if k.startswith(prefix)}
    filtered, unused = remap_to(module.state_dict().keys(), out)
    res = module.load_state_dict(filtered, strict=False)
    if strict and (res.missing_keys or unused):
        raise RuntimeError(f'checkpoint {path}: missing {res.missing_keys[:8]}..., unexpected {unused[:8]}...')
    module._last_load_report = dict(loaded=len(filtered), missing=list(res.missing_keys), unused=unused)
    return True


def save_checkpoint(module, path, fmt=None, meta=None):
    """state_dict -> `.safetensors`, HF-style `.bin` (flat) or mmengine-style `.pth` (dict(meta, state_dict))."""
    sd = {k: v.detach().cpu().contiguous() for k, v in module.state_dict().items()}
    fmt = fmt or ('safetensors' if path.endswith('.safetensors') else 'mmengine' if path.endswith('.pth') else 'flat')
    if fmt == 'safetensors':
        from safetensors.torch import save_file
        save_file(sd, path)
    elif fmt == 'mmengine':
        torch.save(dict(meta=dict(meta or {}), state_dict=sd), path)
    else:
        torch.save(sd, path)
    return path
