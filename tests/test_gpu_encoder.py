"""-m gpu: SAM ViT encoder (HIP) against the HF oracle on identical seeded weights/inputs.

ViT-B runs the oracle live (9 s); ViT-L / ViT-H / ViT-H + LoRA compare with the oracle's outputs for exactly these fixtures,
sampled and committed by tests/golden/make_golden_encoder.py (round 6: 100 s of CPU oracle forwards out of the GPU suite;
the end-to-end tests of test_gpu_baseline_configs.py keep one live oracle run per encoder variant)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
sys.path.insert(0, GOLDEN)


def test_encoder_matches_oracle_live_vitb(dev):
    from oracle import hf_sam
    from rsprompter_amd.sam_encoder import RSSamVisionEncoder
    from rsprompter_amd.synth import synth_state_dict
    arch = 'base'
    m = RSSamVisionEncoder(f'sam_vit_{arch}', extra_config=dict(output_hidden_states=True))
    sd = synth_state_dict(m.vision_encoder, seed=0)
    m.vision_encoder.load_state_dict(sd)
    o = hf_sam.build_vision_encoder(arch)
    o.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 3, 1024, 1024, generator=g)
    emb_ref, hs_ref = hf_sam.run_vision_encoder(o, x)
    m = m.to(dev)
    out = m(x.to(dev))
    emb, hs = out[0], out[1]
    assert emb.shape == emb_ref.shape and len(hs) == len(hs_ref)
    errs = [float((h.cpu() - r).abs().max()) for h, r in zip(hs, hs_ref)]
    e_emb = float((emb.cpu() - emb_ref).abs().max())
    print('hidden-state max abs err per layer:', ['%.2e' % e for e in errs])
    print('embedding max abs err: %.3e (range %.2f)' % (e_emb, float(emb_ref.abs().max())))
    assert max(errs) < 1e-3 and e_emb < 1e-3  # north-star tolerance: 1e-3 fp32


@pytest.mark.parametrize('arch,lora', [('large', False), ('huge', False), ('huge', True)])
def test_encoder_matches_oracle_golden(dev, arch, lora):
    """Every hidden state (sampled at every 16th grid position, all channels) and the image embedding (every 4th position)
    against the HF oracle's for the same seeded weights and input.  + LoRA (BASELINE.json configs[4], models.py:785-792):
    peft's eval-mode LoRA Linear is base(x) + B(A(x)) alpha/r, i.e. a Linear with weight W + (alpha/r) B A -- the oracle ran
    with those merged weights, the HIP encoder gets base and adapter tensors separately under peft's key layout."""
    import make_golden_encoder as mg
    path = os.path.join(GOLDEN, mg.golden_name(arch, lora))
    g = torch.load(path, map_location='cpu', weights_only=True)
    m, sd, _, x = mg.fixture(arch, lora)
    m.vision_encoder.load_state_dict(sd)
    out = m.to(dev)(x.to(dev))
    emb, hs = out[0], out[1]
    assert len(hs) == len(g['hidden_samples']) and bool(torch.isfinite(emb).all())
    s, es = g['hs_stride'], g['emb_stride']
    errs = [float((h[:, ::s, ::s, :].cpu() - r).abs().max()) for h, r in zip(hs, g['hidden_samples'])]
    assert all(bool(torch.isfinite(h).all()) for h in hs)
    e_emb = float((emb[:, :, ::es, ::es].cpu() - g['embedding_sample']).abs().max())
    print(f'ViT-{arch}{" + LoRA" if lora else ""}: hidden-state max abs err per layer (sampled):', ['%.2e' % e for e in errs])
    print('embedding max abs err (sampled): %.3e (range %.2f)' % (e_emb, g['embedding_absmax']))
    assert max(errs) < 1e-3 and e_emb < 1e-3  # north-star tolerance: 1e-3 fp32
    if lora:
        # the adapter tensors travel under peft's names
        keys = m.state_dict().keys()
        assert any('base_model.model' in k and 'lora_A.default' in k for k in keys)
