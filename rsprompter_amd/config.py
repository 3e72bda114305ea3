"""Minimal loader for the reference's python config files
(mmengine.Config semantics used by configs/rsprompter/*.py: `_base_`
inheritance, recursive dict merge, `_delete_=True`, attribute access;
SURVEY.md §5.6).  mmengine itself is not available in the target image.
"""
import copy
import os


class ConfigDict(dict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e

    def __setattr__(self, name, value):
        self[name] = value

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _to_cd(x):
    if isinstance(x, dict):
        return ConfigDict({k: _to_cd(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_to_cd(v) for v in x]
    if isinstance(x, tuple):
        return tuple(_to_cd(v) for v in x)
    return x


def _merge(base, child):
    """child overrides base; dicts merge recursively unless child has _delete_=True."""
    out = dict(base)
    for k, v in child.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            if v.get('_delete_', False):
                v = {kk: vv for kk, vv in v.items() if kk != '_delete_'}
                out[k] = v
            else:
                out[k] = _merge(out[k], v)
        else:
            if isinstance(v, dict) and v.get('_delete_', False):
                v = {kk: vv for kk, vv in v.items() if kk != '_delete_'}
            out[k] = v
    return out


def _load_py(path):
    path = os.path.abspath(path)
    with open(path) as f:
        src = f.read()
    ns = {'__file__': path}
    exec(compile(src, path, 'exec'), ns)
    cfg = {k: v for k, v in ns.items()
           if not k.startswith('__') and not callable(v) and not isinstance(v, type(os))}
    bases = cfg.pop('_base_', [])
    if isinstance(bases, str):
        bases = [bases]
    merged = {}
    for b in bases:
        merged = _merge(merged, _load_py(os.path.join(os.path.dirname(path), b)))
    return _merge(merged, cfg)


class Config(ConfigDict):
    @staticmethod
    def fromfile(path):
        return Config(_to_cd(_load_py(path)))

    def merge_from_dict(self, options):
        """`--cfg-options a.b.c=1` style overrides."""
        for key, v in options.items():
            d = self
            parts = key.split('.')
            for p in parts[:-1]:
                d = d.setdefault(p, ConfigDict())
            d[parts[-1]] = _to_cd(v)
