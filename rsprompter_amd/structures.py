"""Stand-ins for mmengine.structures.InstanceData / mmdet DetDataSample:
just enough for the predict path's input/output contract:
`data_sample.metainfo[...]`, `data_sample.pred_instances.{bboxes,scores,labels,masks}`.
"""
import torch


class InstanceData:
    def __init__(self, **kwargs):
        self._fields = {}
        for k, v in kwargs.items():
            setattr(self, k, v)

    def __setattr__(self, k, v):
        if k.startswith('_'):
            object.__setattr__(self, k, v)
        else:
            self._fields[k] = v

    def __getattr__(self, k):
        f = object.__getattribute__(self, '_fields')
        if k in f:
            return f[k]
        raise AttributeError(k)

    def __delattr__(self, k):
        del self._fields[k]

    def __contains__(self, k):
        return k in self._fields

    def keys(self):
        return self._fields.keys()

    def get(self, k, default=None):
        return self._fields.get(k, default)

    def pop(self, k, *a):
        return self._fields.pop(k, *a)

    def __len__(self):
        for v in self._fields.values():
            return len(v)
        return 0

    def __getitem__(self, idx):
        out = InstanceData()
        for k, v in self._fields.items():
            out._fields[k] = v[idx]
        return out

    def to(self, *a, **kw):
        out = InstanceData()
        for k, v in self._fields.items():
            out._fields[k] = v.to(*a, **kw) if isinstance(v, torch.Tensor) else v
        return out

    def __repr__(self):
        return 'InstanceData(' + ', '.join(
            f'{k}={tuple(v.shape) if hasattr(v, "shape") else v}' for k, v in self._fields.items()) + ')'


class DetDataSample:
    def __init__(self, metainfo=None):
        self._metainfo = dict(metainfo or {})
        self._data = {}

    @property
    def metainfo(self):
        return self._metainfo

    def set_metainfo(self, m):
        self._metainfo.update(m)

    def get(self, k, default=None):
        if k in self._data:
            return self._data[k]
        return self._metainfo.get(k, default)

    def __getattr__(self, k):
        if k.startswith('_'):
            raise AttributeError(k)
        d = object.__getattribute__(self, '_data')
        if k in d:
            return d[k]
        m = object.__getattribute__(self, '_metainfo')
        if k in m:
            return m[k]
        raise AttributeError(k)

    def __setattr__(self, k, v):
        if k.startswith('_'):
            object.__setattr__(self, k, v)
        else:
            self._data[k] = v
