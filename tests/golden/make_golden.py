"""Generate golden vectors by executing the REAL reference source files.

Run in the build container (where /root/reference exists):  python tests/golden/make_golden.py
The reference package cannot be imported as a whole (mmcv / mmengine / peft / torchvision are not
installed), so the individual source files are loaded with permissive stub modules standing in for
the missing third-party imports.  Only functions / classes whose bodies are pure torch (+einops,
numpy) are executed; every tensor they return is stored in tests/golden/reference_vectors.pt together
with the inputs, and tests/test_oracle_golden.py replays them against oracle/ on any machine.
"""
import importlib.machinery
import importlib.util
import os
import sys
import types

import torch
from torch import nn

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'reference_vectors.pt')


class _Anything(type):
    """A class usable as base class / decorator / callable for whatever the reference imports."""
    def __getattr__(cls, name):
        return _make_stub(name)


def _make_stub(name):
    return _Anything(name, (nn.Module,), {'__init__': lambda self, *a, **k: nn.Module.__init__(self)})


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return _make_stub(name)


class _Registry:
    def register_module(self, *a, **k):
        return lambda cls: cls

    def build(self, cfg, *a, **k):
        raise RuntimeError('registry build is not available in the golden generator')


class _StubFinder:
    """meta-path finder: any not-yet-loaded module below these roots resolves to a permissive stub."""
    ROOTS = ('mmcv', 'mmengine', 'mmdet', 'mmpretrain', 'peft')

    def find_spec(self, fullname, path=None, target=None):
        if fullname.split('.')[0] in self.ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def _install_stubs():
    sys.meta_path.append(_StubFinder())
    names = ['mmcv', 'mmcv.cnn', 'mmcv.cnn.bricks', 'mmcv.cnn.bricks.transformer', 'mmcv.ops', 'mmengine',
             'mmengine.dist', 'mmengine.model', 'mmengine.structures', 'mmengine.config', 'mmengine.utils',
             'mmengine.registry', 'mmengine.model.weight_init', 'peft', 'mmdet', 'mmdet.models',
             'mmdet.models.task_modules', 'mmdet.models.utils', 'mmdet.structures', 'mmdet.structures.bbox',
             'mmdet.utils', 'mmpretrain', 'mmpretrain.models', 'mmpretrain.registry', 'mmdet.registry',
             'mmpretrain.models.utils', 'mmpretrain.models.backbones.base_backbone']
    for n in names:
        m = _StubModule(n)
        m.__path__ = []
        sys.modules[n] = m
    reg = _Registry()
    sys.modules['mmdet.registry'].MODELS = reg
    sys.modules['mmdet.registry'].TASK_UTILS = reg
    sys.modules['mmpretrain.registry'].MODELS = reg
    sys.modules['mmengine.model'].BaseModule = type('BaseModule', (nn.Module,), {
        '__init__': lambda self, init_cfg=None: nn.Module.__init__(self)})
    sys.modules['mmengine'].ConfigDict = dict
    sys.modules['mmengine.dist'].is_main_process = lambda: True
    sys.modules['mmengine.utils'].to_2tuple = lambda v: (v, v) if not isinstance(v, (tuple, list)) else tuple(v)

    class LayerNorm2d(nn.LayerNorm):      # mmpretrain/models/utils/norm.py:52-90 restated for PseudoFeatureAggregator
        def forward(self, x):
            return torch.nn.functional.layer_norm(x.permute(0, 2, 3, 1), self.normalized_shape, self.weight,
                                                  self.bias, self.eps).permute(0, 3, 1, 2)
    sys.modules['mmpretrain.models'].LayerNorm2d = LayerNorm2d
    sys.modules['mmdet.utils'].ConfigType = dict
    sys.modules['mmdet.utils'].OptConfigType = dict
    sys.modules['mmdet.utils'].MultiConfig = dict
    sys.modules['mmdet.utils'].OptMultiConfig = dict
    sys.modules['mmdet.utils'].InstanceList = list
    sys.modules['mmdet.structures'].SampleList = list
    sys.modules['mmdet.structures'].OptSampleList = list
    bbox = sys.modules['mmdet.structures.bbox']
    bbox.get_box_tensor = lambda b: b
    bbox.HorizontalBoxes = lambda b: b
    bbox.BaseBoxes = type('BaseBoxes', (), {})


def _load(rel_path, name):
    name = rel_path[:-3].replace('/', '.')        # real dotted name so that relative imports resolve (to stubs)
    for i in range(1, len(name.split('.'))):
        importlib.import_module('.'.join(name.split('.')[:i]))
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel_path))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    _install_stubs()
    torch.manual_seed(0)
    out = {}

    # ---- delta2bbox (delta_xywh_bbox_coder.py:264-361), incl. the reference's own known-answer test
    coder = _load('mmdet/models/task_modules/coders/delta_xywh_bbox_coder.py', '_ref_coder')
    rois = torch.Tensor([[0., 0., 1., 1.], [0., 0., 1., 1.], [0., 0., 1., 1.], [5., 5., 5., 5.]])
    deltas = torch.Tensor([[0., 0., 0., 0.], [1., 1., 1., 1.], [0., 0., 2., -1.], [0.7, -1.9, -0.5, 0.3]])
    out['delta2bbox_kat'] = dict(rois=rois, deltas=deltas, max_shape=(32, 32),
                                 out=coder.delta2bbox(rois, deltas, max_shape=(32, 32)))
    g = torch.Generator().manual_seed(1)
    xy = torch.rand(200, 2, generator=g) * 900
    r2 = torch.cat([xy, xy + torch.rand(200, 2, generator=g) * 300 + 1], 1)
    d2 = torch.randn(200, 40, generator=g) * 2
    out['delta2bbox_rand'] = dict(rois=r2, deltas=d2, stds=(0.1, 0.1, 0.2, 0.2), max_shape=(1024, 1024),
                                  out=coder.delta2bbox(r2, d2, stds=(0.1, 0.1, 0.2, 0.2), max_shape=(1024, 1024)))
    d3 = torch.randn(200, 4, generator=g) * 3
    out['delta2bbox_rpn'] = dict(rois=r2, deltas=d3, max_shape=(1024, 1000),
                                 out=coder.delta2bbox(r2, d3, max_shape=(1024, 1000)))

    # ---- AnchorGenerator (anchor_generator.py:69-301)
    ag = _load('mmdet/models/task_modules/prior_generators/anchor_generator.py', '_ref_anchor')
    gen = ag.AnchorGenerator(strides=[4, 8, 16, 32, 64], ratios=[0.5, 1.0, 2.0], scales=[4, 8])
    sizes = [(256, 256), (128, 128), (64, 64), (32, 32), (16, 16)]
    pri = gen.grid_priors(sizes, device='cpu')
    out['anchors'] = dict(base=[b.clone() for b in gen.base_anchors], sizes=sizes,
                          sample_idx=[torch.arange(0, p.shape[0], max(1, p.shape[0] // 997)) for p in pri])
    out['anchors']['samples'] = [p[i] for p, i in zip(pri, out['anchors']['sample_idx'])]
    out['anchors']['counts'] = [p.shape[0] for p in pri]
    # the reference's own known-answer test (tests/.../test_anchor_generator.py:290-309)
    gen2 = ag.AnchorGenerator(strides=[4, 8], ratios=[1.], scales=[1.], base_sizes=[4, 8])
    out['anchors_kat'] = [a.clone() for a in gen2.grid_priors([(2, 2), (1, 1)], device='cpu')]

    # ---- SinePositionalEncoding (positional_encoding.py:15-110)
    pe_mod = _load('mmdet/models/layers/positional_encoding.py', '_ref_pe')
    spe = pe_mod.SinePositionalEncoding(num_feats=128, normalize=True)
    out['sine_pe'] = spe(torch.zeros((1, 24, 40), dtype=torch.bool))

    # ---- vit_sam.py helpers (window partition / rel-pos; vit_sam.py:17-157)
    import torch.nn.functional as F  # noqa: F401
    vs = _load('mmpretrain/models/backbones/vit_sam.py', '_ref_vitsam')
    x = torch.randn(2, 20, 20, 8, generator=g)
    win, pad_hw = vs.window_partition(x, 14)
    out['window'] = dict(x=x, windows=win, pad_hw=pad_hw, back=vs.window_unpartition(win, 14, pad_hw, (20, 20)))
    rp = torch.randn(27, 16, generator=g)
    out['rel_pos'] = dict(rel_pos=rp, same=vs.get_rel_pos(14, 14, rp), resized=vs.get_rel_pos(20, 20, rp))
    q = torch.randn(3, 14 * 14, 16, generator=g)
    attn = torch.randn(3, 196, 196, generator=g)
    rph, rpw = torch.randn(27, 16, generator=g), torch.randn(27, 16, generator=g)
    out['decomposed_rel_pos'] = dict(q=q, attn=attn, rph=rph, rpw=rpw,
                                     out=vs.add_decomposed_rel_pos(attn.clone(), q, rph, rpw, (14, 14), (14, 14)))

    # ---- mmdet/rsprompter/models.py pure-torch pieces
    import transformers  # noqa: F401  (real)
    models = _load('mmdet/rsprompter/models.py', '_ref_models')
    ln = models.LN2d(8)
    with torch.no_grad():
        ln.weight.copy_(torch.randn(8, generator=g)); ln.bias.copy_(torch.randn(8, generator=g))
    xin = torch.randn(2, 8, 5, 7, generator=g)
    out['ln2d'] = dict(w=ln.weight.detach().clone(), b=ln.bias.detach().clone(), x=xin, out=ln(xin).detach())

    agg = models.RSFeatureAggregator('sam_vit_base', hidden_channels=16, out_channels=64,
                                     select_layers=range(1, 13, 2)).eval()
    sd = {k: torch.randn(v.shape, generator=g) * (0.05 if v.dim() > 1 else 0.5) + (1.0 if 'running_var' in k else 0.0)
          if v.dtype.is_floating_point else v for k, v in agg.state_dict().items()}
    for k in sd:
        if 'running_var' in k:
            sd[k] = sd[k].abs() + 0.5
    agg.load_state_dict(sd)
    hs = tuple(torch.randn(1, 5, 5, 768, generator=g) for _ in range(13))
    with torch.no_grad():
        out['aggregator'] = dict(state=sd, inputs=hs, out=agg(hs))

    # mask post-processing (models.py:1746-1784) as an unbound method on a dummy self
    low = torch.randn(3, 1, 64, 64, generator=g) * 3
    for tag, meta in [('ident', dict(ori_shape=(256, 256), scale_factor=(1.0, 1.0), batch_input_shape=(256, 256))),
                      ('rescale', dict(ori_shape=(128, 128), scale_factor=(2.0, 2.0), batch_input_shape=(256, 256))),
                      ('odd', dict(ori_shape=(150, 100), scale_factor=(1.5, 1.5), batch_input_shape=(256, 256)))]:
        boxes = torch.rand(3, 4, generator=g) * 100
        b_in = boxes.clone()
        m = models.RSPrompterAnchorMaskHead._predict_by_feat_single(
            None, low.clone(), b_in, None, meta, dict(mask_thr_binary=0.5) if False else types.SimpleNamespace(mask_thr_binary=0.5),
            rescale=True)
        out[f'mask_post_{tag}'] = dict(low=low, boxes=boxes, meta=meta, masks=m, boxes_out=b_in)

    # image-wide positional embedding (models.py:85-95) with HF SamPositionalEmbedding
    from transformers.models.sam.configuration_sam import SamVisionConfig
    from transformers.models.sam.modeling_sam import SamPositionalEmbedding
    pe = SamPositionalEmbedding(SamVisionConfig())
    with torch.no_grad():
        pe.positional_embedding.copy_(torch.randn(2, 128, generator=g))
    holder = types.SimpleNamespace(shared_image_embedding=types.SimpleNamespace(shared_image_embedding=pe))
    holder.shared_image_embedding.__call__ = None
    fake = types.SimpleNamespace()
    fake.shared_image_embedding = lambda coords: pe(coords)
    fake.shared_image_embedding.shared_image_embedding = pe
    with torch.no_grad():
        out['image_pe'] = dict(G=pe.positional_embedding.detach().clone(),
                               out=models.RSPrompterAnchor.get_image_wide_positional_embeddings(fake, 16))

    torch.save(out, OUT)
    print('wrote', OUT, {k: (list(v.keys()) if isinstance(v, dict) else type(v).__name__) for k, v in out.items()})


if __name__ == '__main__':
    main()
