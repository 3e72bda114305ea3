"""-m gpu: SAM ViT encoder (HIP) against the HF oracle on identical seeded weights/inputs."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('arch', ['base', 'large', 'huge'])
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


def test_encoder_huge_with_lora_matches_merged_oracle(dev):
    """BASELINE.json configs[4]: ViT-H + LoRA(qkv, r16, alpha32) (models.py:785-792).  peft's eval-mode LoRA Linear is
    base(x) + B(A(x)) * alpha/r, i.e. a Linear with weight W + (alpha/r) B A: the HF oracle runs with those merged
    weights, the HIP encoder gets base and adapter tensors separately under peft's key layout."""
    from oracle import hf_sam
    from rsprompter_amd.sam_encoder import RSSamVisionEncoder
    from rsprompter_amd.synth import synth_state_dict
    cfg = dict(r=16, lora_alpha=32, target_modules=['qkv'], lora_dropout=0.05, bias='none')
    m = RSSamVisionEncoder('sam_vit_huge', extra_config=dict(output_hidden_states=True), peft_config=cfg)
    enc = m.vision_encoder
    sd = synth_state_dict(enc, seed=0)
    g = torch.Generator().manual_seed(5)
    for k in list(sd):
        if 'lora_' in k:
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.05
    enc.load_state_dict(sd)
    merged = {}
    for k, v in sd.items():
        if 'lora_' in k:
            continue
        merged[k] = v.clone()
    for i in range(enc.depth):
        a = sd[f'layers.{i}.attn.qkv.lora_A.default.weight']
        b = sd[f'layers.{i}.attn.qkv.lora_B.default.weight']
        merged[f'layers.{i}.attn.qkv.weight'] = (sd[f'layers.{i}.attn.qkv.weight'].double()
                                                 + (32 / 16) * (b.double() @ a.double())).float()
    o = hf_sam.build_vision_encoder('huge')
    o.load_state_dict(merged, strict=True)
    x = torch.randn(1, 3, 1024, 1024, generator=g)
    emb_ref, hs_ref = hf_sam.run_vision_encoder(o, x)
    out = m.to(dev)(x.to(dev))
    e_emb = float((out[0].cpu() - emb_ref).abs().max())
    e_hs = max(float((h.cpu() - r).abs().max()) for h, r in zip(out[1], hs_ref))
    print('ViT-H + LoRA: embedding err %.3e, hidden-state err %.3e' % (e_emb, e_hs))
    assert e_emb < 1e-3 and e_hs < 1e-3
    # the adapter tensors travel under peft's names
    keys = m.state_dict().keys()
    assert any('base_model.model' in k and 'lora_A.default' in k for k in keys)
