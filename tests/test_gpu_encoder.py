"""-m gpu: SAM ViT encoder (HIP) against the HF oracle on identical seeded weights/inputs."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('arch', ['base'])
def test_encoder_matches_oracle(dev, arch):
    from oracle import hf_sam
    from rsprompter_amd.sam_encoder import RSSamVisionEncoder
    from rsprompter_amd.synth import synth_state_dict
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
