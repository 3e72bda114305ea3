"""HF SAM modules configured like the reference builds them (oracle; tests only).

reference call sites: models.py:772-775 (SamVisionEncoder(vision_config)),
:908-911 (SamMaskDecoder(mask_decoder_config)), :753-756 (SamPositionalEmbedding),
:890-893 (SamPromptEncoder -> no_mask_embed / mask_embed).
`SamConfig.from_pretrained` needs a hub/local config; offline we construct the
same configs from the arch name exactly like models.py:1005 infers it.
"""
import torch
from transformers.models.sam.configuration_sam import (SamMaskDecoderConfig, SamPromptEncoderConfig,
                                                       SamVisionConfig)
from transformers.models.sam import modeling_sam as hf

ARCH = {
    'base': dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 global_attn_indexes=[2, 5, 8, 11]),
    'large': dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
                  global_attn_indexes=[5, 11, 17, 23]),
    'huge': dict(hidden_size=1280, num_hidden_layers=32, num_attention_heads=16,
                 global_attn_indexes=[7, 15, 23, 31]),
}


def infer_arch(name):
    name = str(name)
    return 'base' if 'base' in name else 'large' if 'large' in name else 'huge'


def vision_config(arch, **extra):
    cfg = SamVisionConfig(**ARCH[arch])
    cfg._attn_implementation = 'eager'   # HF:803-831 is what 4.38.1 (the pinned version) runs
    for k, v in extra.items():
        setattr(cfg, k, v)
    return cfg


def build_vision_encoder(arch, lora=None, **extra):
    """lora=dict(r=16, alpha=32): every `attn.qkv` becomes peft's eval-mode LoRA Linear and the encoder is wrapped the way
    `get_peft_model` wraps it (models.py:785-797), so the state_dict carries peft 0.8.2's key layout
    (`base_model.model.layers.N.attn.qkv.{base_layer,lora_A.default,lora_B.default}`)."""
    m = hf.SamVisionEncoder(vision_config(arch, **extra))
    if lora:
        from .vitsam import LoraLinear, PeftWrapped
        for layer in m.layers:
            q = layer.attn.qkv
            layer.attn.qkv = LoraLinear(q.in_features, q.out_features, lora.get('r', 16), lora.get('alpha', 32))
        m = PeftWrapped(m)
    return m.eval()


def build_mask_decoder():
    cfg = SamMaskDecoderConfig()
    cfg._attn_implementation = 'eager'
    return hf.SamMaskDecoder(cfg).eval()


def build_positional_embedding(arch):
    return hf.SamPositionalEmbedding(vision_config(arch)).eval()


def build_mask_embedding():
    return hf.SamMaskEmbedding(SamPromptEncoderConfig()).eval()


@torch.no_grad()
def run_vision_encoder(model, pixel_values):
    """returns (image_embeddings [B,256,g,g], hidden_states tuple of L+1 [B,g,g,D])."""
    from .vitsam import PeftWrapped
    inner = model.base_model.model if isinstance(model, PeftWrapped) else model      # peft wrapper (lora=...)
    out = inner(pixel_values, output_hidden_states=True)
    return out.last_hidden_state, tuple(out.hidden_states)
