"""SAM ViT image encoder on hand-written gfx950 kernels.

Drop-in for the reference's `RSSamVisionEncoder` (mmdet/rsprompter/models.py:762-809),
which wraps HuggingFace `SamVisionEncoder` (HF:1013-1072; layer HF:885-972,
attention HF:700-831, neck HF:975-992).  Same registry name, same ctor kwargs,
same `state_dict` keys (`vision_encoder.patch_embed.projection.weight`, ...),
same output object (`[0]` = image embeddings [B,256,64,64], `[1]` = L+1 hidden
states [B,64,64,D]).

Data layout: the residual stream is one fp32 [B*4096, D] matrix
(NHWC); every Linear is `rsp_gemm`; window partition / unpartition are row maps
folded into the qkv GEMM's A-loader and the proj GEMM's epilogue, so no padded
[B*25,14,14,D] tensor is ever copied.
"""
import math
import os

import torch

from . import ops
from .nnutil import (HIPModule, SAM_ARCH, add_param, infer_sam_arch, load_checkpoint_into,
                     nchw_view)
from .registry import MODELS


class SamVisionEncoderOutput(tuple):
    """Tuple-like stand-in of transformers' SamVisionEncoderOutput (models.py:99-101)."""

    def __new__(cls, last_hidden_state, hidden_states=None):
        items = (last_hidden_state,) if hidden_states is None else (last_hidden_state, hidden_states)
        self = super().__new__(cls, items)
        self.last_hidden_state = last_hidden_state
        self.hidden_states = hidden_states
        return self


def resize_rel_pos(rel_pos, size):
    """HF get_rel_pos (HF:729-759): linear interpolation of the table to 2*size-1 rows.
    Load-time constant preparation (identity when the table already has that length)."""
    n = 2 * size - 1
    if rel_pos.shape[0] == n:
        return rel_pos
    r = torch.nn.functional.interpolate(
        rel_pos.reshape(1, rel_pos.shape[0], -1).transpose(1, 2).float(), size=n, mode='linear')
    return r.reshape(-1, n).permute(1, 0).contiguous()


NAMES = {
    # HuggingFace SamVisionEncoder attribute paths (HF:1013-1072) ...
    'hf': dict(ln1='layer_norm1', ln2='layer_norm2', lin1='mlp.lin1', lin2='mlp.lin2', conv1='neck.conv1',
               nln1='neck.layer_norm1', conv2='neck.conv2', nln2='neck.layer_norm2'),
    # ... and the in-repo mmpretrain ViTSAM's (vit_sam.py:224-313,514-530; key renames models.py:840-851)
    'mmpretrain': dict(ln1='ln1', ln2='ln2', lin1='ffn.layers.0.0', lin2='ffn.layers.1',
                       conv1='channel_reduction.0', nln1='channel_reduction.1', conv2='channel_reduction.2',
                       nln2='channel_reduction.3'),
}


def _get(root, dotted):
    for part in dotted.split('.'):
        root = getattr(root, part)
    return root


class SamVisionEncoderHIP(HIPModule):
    def __init__(self, arch='base', image_size=1024, patch_size=16, window_size=14,
                 out_channels=256, output_hidden_states=False, layer_norm_eps=1e-6, naming='hf'):
        super().__init__()
        self.nm = NAMES[naming]
        nm = self.nm
        a = SAM_ARCH[arch]
        self.arch = arch
        self.D, self.depth, self.heads = a['hidden'], a['depth'], a['heads']
        self.global_idx, self.mlp_dim = tuple(a['global_idx']), a['mlp']
        self.dh = self.D // self.heads
        self.image_size, self.patch_size, self.window_size = image_size, patch_size, window_size
        self.grid = image_size // patch_size
        self.out_channels = out_channels
        self.output_hidden_states = output_hidden_states
        self.eps = layer_norm_eps
        D, g = self.D, self.grid
        add_param(self, 'patch_embed.projection.weight', (D, 3, patch_size, patch_size))
        add_param(self, 'patch_embed.projection.bias', (D,))
        add_param(self, 'pos_embed', (1, g, g, D))
        for i in range(self.depth):
            s = g if i in self.global_idx else window_size
            p = f'layers.{i}.'
            add_param(self, p + nm['ln1'] + '.weight', (D,), 1.0)
            add_param(self, p + nm['ln1'] + '.bias', (D,))
            add_param(self, p + 'attn.qkv.weight', (3 * D, D))
            add_param(self, p + 'attn.qkv.bias', (3 * D,))
            add_param(self, p + 'attn.proj.weight', (D, D))
            add_param(self, p + 'attn.proj.bias', (D,))
            add_param(self, p + 'attn.rel_pos_h', (2 * s - 1, self.dh))
            add_param(self, p + 'attn.rel_pos_w', (2 * s - 1, self.dh))
            add_param(self, p + nm['ln2'] + '.weight', (D,), 1.0)
            add_param(self, p + nm['ln2'] + '.bias', (D,))
            add_param(self, p + nm['lin1'] + '.weight', (self.mlp_dim, D))
            add_param(self, p + nm['lin1'] + '.bias', (self.mlp_dim,))
            add_param(self, p + nm['lin2'] + '.weight', (D, self.mlp_dim))
            add_param(self, p + nm['lin2'] + '.bias', (D,))
        add_param(self, nm['conv1'] + '.weight', (out_channels, D, 1, 1))
        add_param(self, nm['nln1'] + '.weight', (out_channels,), 1.0)
        add_param(self, nm['nln1'] + '.bias', (out_channels,))
        add_param(self, nm['conv2'] + '.weight', (out_channels, out_channels, 3, 3))
        add_param(self, nm['nln2'] + '.weight', (out_channels,), 1.0)
        add_param(self, nm['nln2'] + '.bias', (out_channels,))
        self.lora = None  # optional dict name -> (A [r,D], B [3D,r], scale), merged at pack time
        self._maps = {}

    # ------------------------------------------------------------------ packing
    def _pack(self):
        dev = self.pos_embed.device
        ops.require_device(dev)
        # the four big GEMMs of every block run the fp8-corrected product (ops.F8_CORR): weights and
        # the activation planes feeding them (LN / attention / GELU epilogues) carry the cat8 second plane
        # ops.F8_CORR: False | True / 'all' (qkv, proj, lin1, lin2) | 'mlp' (lin1 + lin2 only: 58 % of the encoder's GEMM FLOPs;
        # round 6 study)
        pol = ops.F8_CORR
        f8 = bool(pol) and pol != 'mlp'          # qkv / proj (and the LayerNorm-1 / attention planes feeding them)
        f8m = bool(pol)                          # lin1 / lin2 (LayerNorm-2 planes, lin1's GELU planes)
        P = {'f8': f8, 'f8_mlp': f8m}
        w = self.patch_embed.projection.weight
        P['patch'] = ops.PackedWeight(w.reshape(w.shape[0], -1), self.patch_embed.projection.bias)
        P['pos'] = self.pos_embed.detach().reshape(-1, self.D).contiguous()
        P['layers'] = []
        for i in range(self.depth):
            L = getattr(self.layers, str(i))
            nm = self.nm
            ln1, ln2 = _get(L, nm['ln1']), _get(L, nm['ln2'])
            lin1, lin2 = _get(L, nm['lin1']), _get(L, nm['lin2'])
            s = self.grid if i in self.global_idx else self.window_size
            wq = L.attn.qkv.weight.detach()
            if self.lora is not None and i in self.lora:
                A, Bm, sc = self.lora[i]
                wq = wq + sc * (Bm.to(wq) @ A.to(wq))   # load-time merge W += (alpha/r) B A (models.py:785-797)
            rph = resize_rel_pos(L.attn.rel_pos_h.detach(), s).contiguous()
            rpw = resize_rel_pos(L.attn.rel_pos_w.detach(), s).contiguous()
            # windowed layers: the rel-pos terms are computed inside the attention kernel from tables split once here
            reltab = (ops.pack_relpos_tables(rph.float(), rpw.float(), s, self.dh)
                      if (s == 14 and self.dh in (64, 80)) else None)
            P['layers'].append(dict(
                S=s, reltab=reltab,
                ln1=(ln1.weight.detach(), ln1.bias.detach()),
                ln2=(ln2.weight.detach(), ln2.bias.detach()),
                qkv=ops.PackedWeight(wq, L.attn.qkv.bias, f8=f8),
                proj=ops.PackedWeight(L.attn.proj.weight, L.attn.proj.bias, f8=f8),
                lin1=ops.PackedWeight(lin1.weight, lin1.bias, f8=f8m),
                lin2=ops.PackedWeight(lin2.weight, lin2.bias, f8=f8m),
                rph=rph, rpw=rpw,
            ))
        nm = self.nm
        c1, c2, n1, n2 = (_get(self, nm[k]) for k in ('conv1', 'conv2', 'nln1', 'nln2'))
        P['neck1'] = ops.PackedWeight(c1.weight.reshape(self.out_channels, self.D))
        # 3x3 conv weight [O, I, ky, kx] -> [O, (ky, kx, I)] to match the NHWC implicit-GEMM K order
        P['neck2'] = ops.PackedWeight(c2.weight.permute(0, 2, 3, 1).reshape(self.out_channels, -1))
        P['nln1'] = (n1.weight.detach(), n1.bias.detach())
        P['nln2'] = (n2.weight.detach(), n2.bias.detach())
        self._packed = P
        self._maps = {}

    def _window_map(self, B, device):
        """row map of window_partition (HF:900-922): GEMM row (b, wy, wx, iy, ix) -> token row or -1 (pad)."""
        key = (B, str(device))
        if key in self._maps:
            return self._maps[key]
        g, w = self.grid, self.window_size
        nw = (g + w - 1) // w
        idx = torch.arange(B * nw * nw * w * w, dtype=torch.int64)
        ix = idx % w
        iy = (idx // w) % w
        wx = (idx // (w * w)) % nw
        wy = (idx // (w * w * nw)) % nw
        b = idx // (w * w * nw * nw)
        y, x = wy * w + iy, wx * w + ix
        src = torch.where((y < g) & (x < g), b * g * g + y * g + x, torch.full_like(idx, -1))
        # the inverse: token row -> its row in window order (every token lies in exactly one window), and the padded rows
        real = src >= 0
        tok2win = torch.empty(B * g * g, dtype=torch.int64)
        tok2win[src[real]] = idx[real]
        self._maps[key] = (src.to(torch.int32).to(device), nw, tok2win.to(torch.int32).to(device),
                           idx[~real].to(torch.int32).to(device))
        return self._maps[key]

    def _kv_planes(self, layer, B, rows, L, pad_rows, device):
        """the K | V plane buffer windowed layer `layer` keeps (padded rows = the bias, written once).  Footprint: 2 D fp32-
        equivalents per window row -- 0.4 GB per layer at ViT-H / B = 8, 11 GB over its 28 windowed layers -- so only ONE
        batch size's set stays resident: a call with another B (a short last batch) drops the previous set first.
        The buffers belong to the module: two forwards of one module on different streams must not overlap."""
        key = ('kv', layer, B, str(device))
        if key not in self._maps:
            for k in [k for k in self._maps if k[0] == 'kv' and k[2:] != key[2:]]:
                del self._maps[k]
            kv = ops.empty_planes((rows, 2 * self.D), device)      # (fp16 hi / lo: the attention kernels' operand format)
            ops.fill_bias_rows(L['qkv'].bias, pad_rows, 3 * self.D, out=None, planes=kv, c_ncols=self.D, pl_col0=self.D)
            self._maps[key] = kv
        return self._maps[key]

    # ------------------------------------------------------------------ forward
    def forward(self, pixel_values, output_hidden_states=None):
        if pixel_values.dim() != 4 or pixel_values.shape[1] != 3:
            raise ValueError('Make sure that the channel dimension of the pixel values match with the one set '
                             'in the configuration.')
        B, _, H, W = pixel_values.shape
        if H != self.image_size or W != self.image_size:
            raise ValueError(f"Input image size ({H}*{W}) doesn't match model ({self.image_size}*{self.image_size}).")
        if self._packed is None:
            self._pack()
        want_hidden = bool(self.output_hidden_states if output_hidden_states is None else output_hidden_states)
        return self._run(pixel_values, want_hidden)

    def _run(self, pixel_values, want_hidden):
        """the launch sequence: no host synchronisation, no host-side decision that depends on device data"""
        P = self._packed
        B = pixel_values.shape[0]
        g, D, nh, dh = self.grid, self.D, self.heads, self.dh
        T = g * g
        x_in = pixel_values.to(torch.float32).contiguous()
        patches = ops.patchify(x_in, self.patch_size)
        # conv16x16 + bias + pos_embed (HF:118-129, 1064-1066): pos_embed is a residual broadcast over the batch
        x = ops.gemm(patches, P['patch'], res=P['pos'], res_mod=T)
        del patches
        hidden = [x] if want_hidden else None
        scale = dh ** -0.5
        f8, f8m = P['f8'], P['f8_mlp']
        for i in range(self.depth):
            L = P['layers'][i]
            S = L['S']
            # GEMM A operands travel as fp16 (hi, lo) planes: LN / attention / GELU epilogues emit them
            xn = ops.layernorm(x, L['ln1'][0], L['ln1'][1], self.eps, planes=True, f32=False, f8=f8)
            # qkv projection: q leaves as fp32 (rel-pos + the attention's Q operand), K | V as the fp16 planes the
            # attention kernel DMAs -- no fp32 K / V tensor, no split pass, no transposed V (csrc/attn_stream.hip)
            if S == g:  # global attention layer
                q, kv = ops.gemm(xn, L['qkv'], out_planes=True, c_ncols=D, pl_col0=D)
                Bp, rowmap = B, None
            else:
                # windowed: window_partition (HF:900-922) is a row SCATTER in the qkv GEMM's epilogue -- the GEMM runs over
                # the B * T real tokens only; the padded rows of the windows (16 % at 1024 px: zero tokens, so qkv = bias,
                # HF:913-915) are filled with the bias by a copy kernel instead of being multiplied
                _, nw, tok2win, pad_rows = self._window_map(B, x.device)
                Bp = B * nw * nw
                # (only K | V: the q rows of padded tokens are never read -- rel-pos and attention below work on the real
                # tokens only.)  The K | V planes of a windowed layer live in a buffer the layer keeps per batch size: its
                # padded rows -- a constant of the layer -- are written ONCE, the GEMM's row scatter never touches them
                # (round 4 filled them per call: 28 launches, 0.8 ms of a ViT-H step; 0.4 GB per layer and batch of 8).
                kv = self._kv_planes(i, B, Bp * S * S, L, pad_rows, x.device)
                q, kv = ops.gemm(xn, L['qkv'], c_rowmap=tok2win, out_rows=Bp * S * S, out_planes=kv, c_ncols=D,
                                 pl_col0=D)
                rowmap = tok2win
            # windows of the last grid row / column hold padding: only their real tokens are queries (the proj GEMM
            # below gathers nothing else)
            wg = None if S == g else (nw, g - (nw - 1) * S)
            if S != g and L['reltab'] is not None:
                # windowed layer: rel-pos terms (HF:761-801), bias, softmax and PV in ONE kernel (csrc/attn_win.hip)
                rel = None
                att = ops.vit_window_attention(q, kv, L['reltab'], Bp, nh, dh, scale, planes=True, f8=f8, win_grid=wg)
            else:
                rel = ops.vit_relpos(q, L['rph'], L['rpw'], Bp, S, nh, dh, q_ld=D, rows=rowmap)
                att = ops.vit_attention_planes(q, kv, rel, Bp, S, nh, dh, scale, planes=True, f8=f8, win_grid=wg)
            qkv = (q, kv)
            # proj + window_unpartition + crop + residual (HF:830, 924-952, 969): a row GATHER of the real tokens from
            # window order (the padded rows are never multiplied)
            x1 = ops.gemm(att, L['proj'], res=x, a_rowmap=rowmap, M=B * T)
            del qkv, rel, att, xn
            xn2 = ops.layernorm(x1, L['ln2'][0], L['ln2'][1], self.eps, planes=True, f32=False, f8=f8m)
            hmid = ops.gemm(xn2, L['lin1'], act=ops.ACT_GELU, out_planes=True, out_f32=False, out_f8=f8m)
            x = ops.gemm(hmid, L['lin2'], res=x1)
            del hmid, xn2, x1
            if want_hidden:
                hidden.append(x)
        # neck (HF:985-992): 1x1 conv -> LN over C -> 3x3 conv -> LN over C, all NHWC
        y = ops.gemm(x, P['neck1'], bias=None)
        y = ops.layernorm(y, P['nln1'][0], P['nln1'][1], 1e-6)
        y = ops.gemm(y.view(B, g, g, self.out_channels), P['neck2'], bias=None, conv=(3, 1, 1))
        y = ops.layernorm(y, P['nln2'][0], P['nln2'][1], 1e-6)
        emb = nchw_view(y.view(B, g, g, self.out_channels))
        hs = tuple(h.view(B, g, g, D) for h in hidden) if want_hidden else None
        return SamVisionEncoderOutput(emb, hs)


def _lora_defaults(peft_config):
    """the reference's LoRA defaults (models.py:786-795 / 855-864): r=16, alpha=32, target qkv."""
    cfg = dict(r=16, lora_alpha=32, target_modules=['qkv'])
    cfg.update(peft_config or {})
    if list(cfg.get('target_modules', ['qkv'])) != ['qkv']:
        raise NotImplementedError('only LoRA on `qkv` (the reference configuration) is implemented')
    return cfg


class _PeftKeyLayout:
    """state_dict key translation between this module and the peft 0.8.2 wrapper layout the reference
    checkpoints use: `<prefix>base_model.model.<path>` and `qkv.base_layer.*`."""

    def _install_peft_hooks(self):
        self._register_load_state_dict_pre_hook(self._peft_load_hook)
        self._register_state_dict_hook(self._peft_save_hook)

    def _peft_load_hook(self, state_dict, prefix, *args):
        root = prefix + 'vision_encoder.'
        for k in [k for k in state_dict if k.startswith(root + 'base_model.model.')]:
            nk = root + k[len(root + 'base_model.model.'):].replace('.qkv.base_layer.', '.qkv.')
            state_dict[nk] = state_dict.pop(k)

    @staticmethod
    def _peft_save_hook(module, state_dict, prefix, local_metadata):
        root = prefix + 'vision_encoder.'
        for k in [k for k in state_dict if k.startswith(root)]:
            tail = k[len(root):]
            if tail.startswith('base_model.model.'):
                continue
            if '.attn.qkv.weight' in tail or '.attn.qkv.bias' in tail:
                tail = tail.replace('.attn.qkv.', '.attn.qkv.base_layer.')
            state_dict[root + 'base_model.model.' + tail] = state_dict.pop(k)
        return state_dict


def _attach_lora(enc, peft_config):
    cfg = _lora_defaults(peft_config)
    r = cfg['r']
    enc.lora_scale = cfg['lora_alpha'] / r
    for i in range(enc.depth):
        add_param(enc, f'layers.{i}.attn.qkv.lora_A.default.weight', (r, enc.D))
        add_param(enc, f'layers.{i}.attn.qkv.lora_B.default.weight', (3 * enc.D, r))
    enc.lora = _LoraView(enc)


def _resize_on_load(enc, state_dict, prefix):
    """load-time interpolation of pos_embed (bicubic) and rel_pos_* (linear) when the checkpoint was
    trained at another resolution (vit_sam.py:612-662, mmpretrain/models/utils/embed.py:16-59)."""
    name = prefix + 'pos_embed'
    if name in state_dict and tuple(state_dict[name].shape) != tuple(enc.pos_embed.shape):
        src = state_dict[name].float().permute(0, 3, 1, 2)
        dst = torch.nn.functional.interpolate(src, size=enc.pos_embed.shape[1:3], align_corners=False, mode='bicubic')
        state_dict[name] = dst.permute(0, 2, 3, 1).contiguous()
    own = enc.state_dict()
    for k, cur in own.items():
        if 'rel_pos_' in k and prefix + k in state_dict and state_dict[prefix + k].shape[0] != cur.shape[0]:
            state_dict[prefix + k] = resize_rel_pos(state_dict[prefix + k].float(), (cur.shape[0] + 1) // 2)


@MODELS.register_module()
class RSSamVisionEncoder(HIPModule, _PeftKeyLayout):
    """Registry-compatible wrapper (reference models.py:762-809)."""

    def __init__(self, hf_pretrain_name, extra_config=None, peft_config=None, init_cfg=None):
        super().__init__()
        extra_config = dict(extra_config or {})
        arch = infer_sam_arch(hf_pretrain_name)
        self.vision_encoder = SamVisionEncoderHIP(
            arch=arch, output_hidden_states=bool(extra_config.get('output_hidden_states', False)))
        if init_cfg is not None:
            load_checkpoint_into(self.vision_encoder, init_cfg.get('checkpoint'),
                                 revise_keys=[(r'^module\.', ''), (r'^vision_encoder\.', '')])
        self.peft_config = peft_config
        if peft_config is not None and isinstance(peft_config, dict):
            _attach_lora(self.vision_encoder, peft_config)
            self._install_peft_hooks()
        self.vision_encoder.is_init = True

    def forward(self, *args, **kwargs):
        return self.vision_encoder(*args, **kwargs)


@MODELS.register_module()
class MMPretrainSamVisionEncoder(HIPModule, _PeftKeyLayout):
    """models.py:812-878: the in-repo mmpretrain `ViTSAM` (vit_sam.py:317-602) at `img_size` (512 in the
    *-peft-512 configs: 32x32 grid, 3x3 windows of 14 on the padded 42x42 grid, global S=32), LoRA on qkv,
    returning the 1-tuple (channel_reduction output,) that extract_feat expects (models.py:102-104)."""

    def __init__(self, hf_pretrain_name, img_size=1024, peft_config=None, init_cfg=None):
        super().__init__()
        arch = str(hf_pretrain_name).split('-')[-1].split('_')[-1]      # models.py:824
        if arch not in SAM_ARCH:
            arch = infer_sam_arch(hf_pretrain_name)
        self.vision_encoder = SamVisionEncoderHIP(arch=arch, image_size=img_size, naming='mmpretrain')
        enc = self.vision_encoder
        enc._register_load_state_dict_pre_hook(lambda sd, prefix, *a: _resize_on_load(enc, sd, prefix))
        if init_cfg is not None:
            load_checkpoint_into(enc, init_cfg.get('checkpoint'), revise_keys=[
                (r'^module\.', ''), (r'^vision_encoder\.', ''), (r'.layer_norm1.', '.ln1.'),
                (r'.layer_norm2.', '.ln2.'), (r'.mlp.lin1.', '.ffn.layers.0.0.'), (r'.mlp.lin2.', '.ffn.layers.1.'),
                (r'neck.conv1.', 'channel_reduction.0.'), (r'neck.ln1.', 'channel_reduction.1.'),
                (r'neck.conv2.', 'channel_reduction.2.'), (r'neck.ln2.', 'channel_reduction.3.')])
        if peft_config is not None and isinstance(peft_config, dict):
            _attach_lora(enc, peft_config)
            self._install_peft_hooks()
        enc.is_init = True

    def forward(self, x):
        out = self.vision_encoder(x, output_hidden_states=False)
        return (out[0],)


class _LoraView:
    """index -> (A, B, scale) view over the LoRA parameters registered on the encoder."""

    def __init__(self, enc):
        self.enc = enc

    def __contains__(self, i):
        return True

    def __getitem__(self, i):
        q = getattr(self.enc.layers, str(i)).attn.qkv
        return q.lora_A.default.weight.detach(), q.lora_B.default.weight.detach(), self.enc.lora_scale
