"""`MODELS` registry shim with the mmengine surface the RSPrompter configs rely on
(reference: mmdet/registry.py:62, `@MODELS.register_module()` in
mmdet/rsprompter/models.py, `MODELS.build(cfg)` e.g. models.py:63,927,1626).

mmengine is not installed in the target image, so this is a ~60-line stand-in:
type strings -> classes, `build(cfg_dict)` -> instance, `force=True` overrides.
Scope prefixes such as 'mmdet.' / 'mmpretrain.' are accepted and stripped.
"""
import copy


class Registry:
    def __init__(self, name):
        self.name = name
        self._module_dict = {}

    def __contains__(self, key):
        return self.get(key) is not None

    def get(self, key):
        if not isinstance(key, str):
            return key
        if key in self._module_dict:
            return self._module_dict[key]
        short = key.split('.')[-1]
        return self._module_dict.get(short)

    def _register(self, cls, name=None, force=False):
        names = [name or cls.__name__] if not isinstance(name, (list, tuple)) else list(name)
        for n in names:
            if n in self._module_dict and not force and self._module_dict[n] is not cls:
                raise KeyError(f'{n} is already registered in {self.name}')
            self._module_dict[n] = cls

    def register_module(self, name=None, force=False, module=None):
        if module is not None:
            self._register(module, name, force)
            return module

        def deco(cls):
            self._register(cls, name, force)
            return cls
        return deco

    def build(self, cfg, *args, **default_args):
        if cfg is None:
            return None
        if not isinstance(cfg, dict):
            return cfg  # already built
        cfg = copy.copy(dict(cfg))
        for k, v in default_args.items():
            cfg.setdefault(k, v)
        if 'type' not in cfg:
            raise KeyError(f'`cfg` must contain the key "type", got {list(cfg)}')
        typ = cfg.pop('type')
        cls = self.get(typ)
        if cls is None:
            raise KeyError(f'{typ} is not in the {self.name} registry')
        return cls(*args, **cfg)


MODELS = Registry('model')
TASK_UTILS = Registry('task util')
