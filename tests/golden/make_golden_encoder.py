"""Golden vectors for tests/test_gpu_encoder.py (round 6: the GPU suite runs against the driver's wall-clock limit, and the CPU
oracle's ViT-H / ViT-L forward costs 25-40 s per tile): the HF SAM vision encoder (oracle/hf_sam.py -- the reference's own
third-party encoder, models.py:772-775) executed HERE on the tests' seeded weights and input, sampled:

  python tests/golden/make_golden_encoder.py huge         -> encoder_huge.pt
  python tests/golden/make_golden_encoder.py large        -> encoder_large.pt
  python tests/golden/make_golden_encoder.py huge --lora  -> encoder_huge_lora.pt   (LoRA(qkv, r16, alpha32) merged: models.py:785-792)

Stored: every hidden state (L + 1 tensors [1, 64, 64, D]) at every 16th grid position (all channels) together with its
max |.|, and the image embedding [1, 256, 64, 64] at every 4th position.  ViT-B stays a live-oracle test (9 s)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

HS_STRIDE, EMB_STRIDE = 16, 4


def fixture(arch, lora):
    """(state dict for the HIP module, state dict for the HF oracle, input) exactly as tests/test_gpu_encoder.py builds them"""
    from rsprompter_amd.sam_encoder import RSSamVisionEncoder
    from rsprompter_amd.synth import synth_state_dict
    if not lora:
        m = RSSamVisionEncoder(f'sam_vit_{arch}', extra_config=dict(output_hidden_states=True))
        sd = synth_state_dict(m.vision_encoder, seed=0)
        g = torch.Generator().manual_seed(11)
        return m, sd, sd, torch.randn(1, 3, 1024, 1024, generator=g)
    cfg = dict(r=16, lora_alpha=32, target_modules=['qkv'], lora_dropout=0.05, bias='none')
    m = RSSamVisionEncoder(f'sam_vit_{arch}', extra_config=dict(output_hidden_states=True), peft_config=cfg)
    enc = m.vision_encoder
    sd = synth_state_dict(enc, seed=0)
    g = torch.Generator().manual_seed(5)
    for k in list(sd):
        if 'lora_' in k:
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.05
    merged = {k: v.clone() for k, v in sd.items() if 'lora_' not in k}
    for i in range(enc.depth):
        a = sd[f'layers.{i}.attn.qkv.lora_A.default.weight']
        b = sd[f'layers.{i}.attn.qkv.lora_B.default.weight']
        merged[f'layers.{i}.attn.qkv.weight'] = (sd[f'layers.{i}.attn.qkv.weight'].double() + (32 / 16) * (b.double() @ a.double())).float()
    x = torch.randn(1, 3, 1024, 1024, generator=g)
    return m, sd, merged, x


def golden_name(arch, lora):
    return f'encoder_{arch}' + ('_lora' if lora else '') + '.pt'


def main(arch, lora):
    from oracle import hf_sam
    t = time.time()
    _, _, osd, x = fixture(arch, lora)
    o = hf_sam.build_vision_encoder(arch)
    o.load_state_dict(osd, strict=True)
    emb, hs = hf_sam.run_vision_encoder(o, x)
    out = dict(arch=arch, lora=bool(lora), hs_stride=HS_STRIDE, emb_stride=EMB_STRIDE,
               hidden_samples=[h[:, ::HS_STRIDE, ::HS_STRIDE, :].contiguous().float() for h in hs],
               hidden_absmax=[float(h.abs().max()) for h in hs],
               embedding_sample=emb[:, :, ::EMB_STRIDE, ::EMB_STRIDE].contiguous().float(), embedding_absmax=float(emb.abs().max()))
    path = os.path.join(ROOT, 'tests', 'golden', golden_name(arch, lora))
    torch.save(out, path)
    print(f'{path}: {len(hs)} hidden states, embedding range {out["embedding_absmax"]:.2f}, {os.path.getsize(path) / 1e3:.0f} kB, {time.time() - t:.0f} s')


if __name__ == '__main__':
    pos = [a for a in sys.argv[1:] if not a.startswith('--')]
    main(pos[0] if pos else 'huge', '--lora' in sys.argv)
