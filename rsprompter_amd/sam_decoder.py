"""SAM mask decoder / positional embedding / prompt-encoder leftovers on HIP kernels.

Reference wrappers: mmdet/rsprompter/models.py  RSSamPositionalEmbedding :744-759,
RSSamPromptEncoder :881-896, RSSamMaskDecoder :899-914 around HuggingFace
SamPositionalEmbedding HF:552-566, SamMaskEmbedding HF:584-593, SamMaskDecoder HF:432-543,
SamTwoWayTransformer HF:351-405, SamTwoWayAttentionBlock HF:272-348, SamAttention HF:194-270.

MI355X design: the reference repeats the [256,64,64] image embedding, its
positional encoding and the dense prompt once per RoI (`repeat_interleave`, models.py:1680-1683)
before entering the decoder.  Here the per-image tensors stay per image: RoI -> image index maps
are applied inside the kernels (attention `kv_batch_map` / `q_batch_map`, GEMM `res_bmap`), the
positional encoding enters every key/query projection as a precomputed broadcast term
`pe @ W^T + b`, and layer-0 image-side projections are computed once per image.
"""
import math

import numpy as np
import torch

from . import ops
from .nnutil import HIPModule, add_param, infer_sam_arch, nchw_view, nhwc_view
from .registry import MODELS

HID = 256          # SamMaskDecoderConfig.hidden_size
HEADS = 8
MLP_DIM = 2048
N_MASK_TOKENS = 4  # num_multimask_outputs + 1
# Both product paths since round 5 (first GPU run + A/B on one box: profiles/r5_decoder_paths_ab.txt, -0.9 ms per ViT-H
# step together): the layer-1 / final token -> image attention with the K | V projections folded in (csrc/t2i_fold.hip)
# and the upscaler tail as one kernel (csrc/upscale.hip, sam_upscale_fused_kernel).  The instance attributes `t2i_fold` /
# `upscale_fused` remain so that the parity tests can hold the fused forms against the kernel chains they replace (those
# chains also serve the shapes the fused kernels do not take: more than 12 tokens, N % 32 != 0, R * N * 512 >= 2^31).


def _upscale2_k_order():
    """input-channel order of the second ConvTranspose's packed weight for rsp_sam_upscale_fused (include/rsp_hip.h): packed
    column 16 s + 8 hh + j holds channel 32 (s >> 1) + 8 ((8 (s & 1) + j) >> 2) + 4 hh + (j & 3)"""
    return [32 * ((kp >> 4) >> 1) + 8 * ((8 * ((kp >> 4) & 1) + (kp & 7)) >> 2) + 4 * ((kp >> 3) & 1) + (kp & 3) for kp in range(64)]


def _add_linear(root, name, cout, cin):
    add_param(root, name + '.weight', (cout, cin))
    add_param(root, name + '.bias', (cout,))


def _add_ln(root, name, c):
    add_param(root, name + '.weight', (c,), 1.0)
    add_param(root, name + '.bias', (c,))


def _add_attn(root, name, internal):
    for p in ('q_proj', 'k_proj', 'v_proj'):
        _add_linear(root, f'{name}.{p}', internal, HID)
    _add_linear(root, f'{name}.out_proj', HID, internal)


def _g(root, dotted):
    for p in dotted.split('.'):
        root = getattr(root, p)
    return root


class SamMaskDecoderHIP(HIPModule):
    """Parameters in HF `SamMaskDecoder` layout."""

    def __init__(self):
        super().__init__()
        add_param(self, 'iou_token.weight', (1, HID))
        add_param(self, 'mask_tokens.weight', (N_MASK_TOKENS, HID))
        for i in range(2):
            p = f'transformer.layers.{i}'
            _add_attn(self, p + '.self_attn', HID)
            _add_attn(self, p + '.cross_attn_token_to_image', HID // 2)
            _add_attn(self, p + '.cross_attn_image_to_token', HID // 2)
            for j in range(1, 5):
                _add_ln(self, f'{p}.layer_norm{j}', HID)
            _add_linear(self, p + '.mlp.lin1', MLP_DIM, HID)
            _add_linear(self, p + '.mlp.lin2', HID, MLP_DIM)
        _add_attn(self, 'transformer.final_attn_token_to_image', HID // 2)
        _add_ln(self, 'transformer.layer_norm_final_attn', HID)
        add_param(self, 'upscale_conv1.weight', (HID, HID // 4, 2, 2))
        add_param(self, 'upscale_conv1.bias', (HID // 4,))
        add_param(self, 'upscale_conv2.weight', (HID // 4, HID // 8, 2, 2))
        add_param(self, 'upscale_conv2.bias', (HID // 8,))
        _add_ln(self, 'upscale_layer_norm', HID // 4)
        for i in range(N_MASK_TOKENS):
            p = f'output_hypernetworks_mlps.{i}'
            _add_linear(self, p + '.proj_in', HID, HID)
            _add_linear(self, p + '.layers.0', HID, HID)
            _add_linear(self, p + '.proj_out', HID // 8, HID)
        _add_linear(self, 'iou_prediction_head.proj_in', HID, HID)
        _add_linear(self, 'iou_prediction_head.layers.0', HID, HID)
        _add_linear(self, 'iou_prediction_head.proj_out', N_MASK_TOKENS, HID)
        self._pe_cache = {}
        # token -> image attention of layer 1 / final with the K | V projections folded in (csrc/t2i_fold.hip)
        self.t2i_fold = True
        # the upscaler tail in one kernel (csrc/upscale.hip, sam_upscale_fused_kernel)
        self.upscale_fused = True
        # parity tests: keep the stage tensors of the last decode() (per-RoI key planes, tokens, upscaler planes, hyper-network
        # vectors) in `_last_stages` so that a failure names its kernel; pins GBs at R = 800, hence off by default
        self.keep_stages = False
        self._last_stages = None
        # tests: decode() chunks above this many prompt sets (None: the folded attention's addressing limit, 1023 at 64 x 64)
        self.max_prompt_sets = None

    # ------------------------------------------------------------------ packing
    def _pw(self, name, with_bias=True):
        m = _g(self, name)
        return ops.PackedWeight(m.weight, m.bias if with_bias else None)

    def _pack(self):
        from .necks import convt_weights4
        P = {}
        for i in range(2):
            p = f'transformer.layers.{i}'
            for a in ('self_attn', 'cross_attn_token_to_image', 'cross_attn_image_to_token'):
                for pr in ('q_proj', 'k_proj', 'v_proj', 'out_proj'):
                    P[f'{i}.{a}.{pr}'] = self._pw(f'{p}.{a}.{pr}')
            P[f'{i}.lin1'] = self._pw(p + '.mlp.lin1')
            P[f'{i}.lin2'] = self._pw(p + '.mlp.lin2')
        for pr in ('q_proj', 'k_proj', 'v_proj', 'out_proj'):
            P[f'final.{pr}'] = self._pw(f'transformer.final_attn_token_to_image.{pr}')
        P['up1'] = convt_weights4(self.upscale_conv1.weight, self.upscale_conv1.bias)
        P['up2'] = convt_weights4(self.upscale_conv2.weight, self.upscale_conv2.bias)
        # ... and with its input channels in the order the fused upscaler consumes them (ops.upscale2_k_order)
        w2 = self.upscale_conv2.weight.detach()[_upscale2_k_order()]
        P['up2p'] = convt_weights4(w2, self.upscale_conv2.bias)
        # K and V projections of the token->image attentions share their A operand: one [256 -> 128+128] GEMM
        for pre, p in (('0.cross_attn_token_to_image', 'transformer.layers.0.cross_attn_token_to_image'),
                       ('1.cross_attn_token_to_image', 'transformer.layers.1.cross_attn_token_to_image'),
                       ('final', 'transformer.final_attn_token_to_image')):
            kp, vp = _g(self, p + '.k_proj'), _g(self, p + '.v_proj')
            P[pre + '.kv_proj'] = ops.PackedWeight(torch.cat([kp.weight, vp.weight], 0))
            # the folded form (csrc/t2i_fold.hip) projects the QUERIES into key space instead: q' = Wk_h^T tq
            P[pre + '.k_projT'] = ops.PackedWeight(kp.weight.detach().t().contiguous())
        for i in range(N_MASK_TOKENS):
            for l in ('proj_in', 'layers.0', 'proj_out'):
                P[f'hyper{i}.{l}'] = self._pw(f'output_hypernetworks_mlps.{i}.{l}')
        for l in ('proj_in', 'layers.0', 'proj_out'):
            P[f'iou.{l}'] = self._pw(f'iou_prediction_head.{l}')
        self._packed = P
        self._pe_cache = {}

    def _pe_terms(self, pe_rows):
        """pe @ W^T + b for every image-side projection that consumes keys + pe (HF:326-343, 397-400)."""
        key = (pe_rows.data_ptr(), tuple(pe_rows.shape))
        if key not in self._pe_cache:
            P = self._packed
            t = {}
            for name in ('0.cross_attn_token_to_image.k_proj', '1.cross_attn_token_to_image.k_proj',
                         '0.cross_attn_image_to_token.q_proj', '1.cross_attn_image_to_token.q_proj',
                         'final.k_proj'):
                t[name] = ops.gemm(pe_rows, P[name])      # includes the projection bias
            for pre, p in (('0.cross_attn_token_to_image', 'transformer.layers.0.cross_attn_token_to_image'),
                           ('1.cross_attn_token_to_image', 'transformer.layers.1.cross_attn_token_to_image'),
                           ('final', 'transformer.final_attn_token_to_image')):
                vb = _g(self, p + '.v_proj').bias
                t[pre + '.kv_proj'] = torch.cat([t[pre + '.k_proj'], vb.unsqueeze(0).expand(pe_rows.shape[0], -1)],
                                                1).contiguous()
                t[pre + '.pek_planes'] = ops.to_planes(t[pre + '.k_proj'])       # PEK of the folded token -> image form
            self._pe_cache = {key: t}
        return self._pe_cache[key]

    # ------------------------------------------------------------------ pieces
    def _ln(self, x, name, eps=1e-6, planes=False):
        m = _g(self, name)
        return ops.layernorm(x, m.weight, m.bias, eps, planes=planes)

    def _t2i(self, tq, kv, ao, R, T, N, kv_map=None):
        """tokens -> image attention on the fused [K | V] projection (HF:326-331, 397-400)."""
        d2, dh2 = HID // 2, (HID // 2) // HEADS
        if T <= ops.SAM_T2I_MAX_TOKENS:
            return ops.sam_t2i_attention(tq, kv, ao, R=R, T=T, N=N, scale=dh2 ** -0.5, kv_map=kv_map)
        kvs = (N * 2 * d2, 2 * d2, dh2)
        return ops.attention(tq, kv, kv[:, d2:], ao, B=R, nh=HEADS, dh=dh2, Tq=T, Tk=N, scale=dh2 ** -0.5,
                             q_strides=(T * d2, d2, dh2), k_strides=kvs, v_strides=kvs,
                             o_strides=(T * d2, d2, dh2), kv_batch_map=kv_map)

    def _t2i_folded(self, pre, tq, keys_pl, pe_t, R, T, N):
        """tokens -> image attention over PER-RoI keys without their K | V projection (csrc/t2i_fold.hip): the queries are
        projected into key space (q' = Wk_h^T tq, one small GEMM on the block-diagonal tq), the kernel returns
        sum_n p[n] keys[n] per (head, token) column, v_proj is applied to that -- HF:326-331 / 397-400 re-associated.
        tq [R*T, 128] projected queries; returns the attention output [R*T, 128] (before out_proj)."""
        P = self._packed
        dh2 = (HID // 2) // HEADS
        # block diagonal: column h * T + t carries head h's 16 query values (softmax scale inside), zeros elsewhere -- one
        # kernel straight into planes (round 5: mul / permute / zeros / index_put / split in torch)
        tqx = ops.sam_fold_expand(tq, R, T, dh2 ** -0.5)
        qp = ops.gemm(tqx, P[pre + '.k_projT'], bias=None, out_planes=True, out_f32=False)      # [R*96, 256] planes
        u = ops.sam_t2i_fold(keys_pl, pe_t[pre + '.pek_planes'], qp, tqx, R=R, N=N, ncols=HEADS * T)
        full = ops.gemm(u, P[pre + '.v_proj'])                      # [R*96, 128]: every head's Wv on every column
        return ops.sam_fold_gather(full, R, T)                      # keep the column's own head: [R*T, 128]

    def _i2t(self, qi, kt, vt, ai, R, T, N, q_map=None):
        """image -> tokens attention; the result feeds the out_proj GEMM as planes (HF:340-345)."""
        d2, dh2 = HID // 2, (HID // 2) // HEADS
        if T <= ops.SAM_I2T_MAX_TOKENS:
            return ops.sam_i2t_attention(qi, kt, vt, R=R, T=T, N=N, scale=dh2 ** -0.5, q_map=q_map, out_planes=ai)
        return ops.attention(qi, kt, vt, None, B=R, nh=HEADS, dh=dh2, Tq=N, Tk=T, scale=dh2 ** -0.5,
                             q_strides=(N * d2, d2, dh2), k_strides=(T * d2, d2, dh2), v_strides=(T * d2, d2, dh2),
                             o_strides=(N * d2, d2, dh2), q_batch_map=q_map, out_planes=ai)

    def _token_attn(self, q_in, k_in, v_in, pfx, R, T, res=None):
        """SamAttention among the T prompt tokens of each RoI (self attention, internal dim 256)."""
        P = self._packed
        q = ops.gemm(q_in, P[pfx + '.q_proj'])
        k = ops.gemm(k_in, P[pfx + '.k_proj'])
        v = ops.gemm(v_in, P[pfx + '.v_proj'])
        dh = HID // HEADS
        o = torch.empty_like(q)
        st = (T * HID, HID, dh)
        ops.attention(q, k, v, o, B=R, nh=HEADS, dh=dh, Tq=T, Tk=T, scale=dh ** -0.5,
                      q_strides=st, k_strides=st, v_strides=st, o_strides=st)
        return ops.gemm(o, P[pfx + '.out_proj'], res=res)

    def decode(self, image_embeddings, image_pe, sparse, dense_vec, roi_img, want_iou=True, src_rows=None, hw=None,
               multimask_output=False, src_is_identity=False):
        """The SAM mask decoder over R prompt sets (see _decode_chunk for the arguments).  Round 6: more prompt sets than the
        folded token -> image attention can address (R * N * 512 bytes of key planes < 2^31: 1023 at N = 4096) are decoded in
        chunks of equal size, each on the product path -- BASELINE configs[2] (16 tiles x 100 queries = 1600 prompt sets) used to
        fall back to the round-2 kernel chain (K | V projection GEMMs over all per-RoI keys + sam_t2i_kernel) for that reason.
        src_is_identity: roi_img is arange(R) over `src_rows` (the query variant: one dense-prompted source per prompt set), so a
        chunk only needs its own rows of `src_rows`."""
        R = sparse.shape[0]
        if src_rows is None:
            N = image_embeddings.shape[-2] * image_embeddings.shape[-1]
        else:
            N = hw[0] * hw[1]
        max_r = self.max_prompt_sets or max(1, (2 ** 31 - 1) // (N * 512))
        if R <= max_r or not self.t2i_fold:
            return self._decode_chunk(image_embeddings, image_pe, sparse, dense_vec, roi_img, want_iou, src_rows, hw,
                                      multimask_output)
        n_chunks = -(-R // max_r)
        per = -(-R // n_chunks)
        masks, ious = [], []
        for r0 in range(0, R, per):
            r1 = min(R, r0 + per)
            if src_rows is not None and src_is_identity:
                src_c = src_rows[r0 * N:r1 * N]
                map_c = roi_img[:r1 - r0]                     # arange(r1 - r0)
            else:
                src_c, map_c = src_rows, roi_img[r0:r1]
            m, i = self._decode_chunk(image_embeddings, image_pe, sparse[r0:r1], dense_vec, map_c, want_iou, src_c, hw,
                                      multimask_output)
            masks.append(m)
            ious.append(i)
        return torch.cat(masks, 0), (torch.cat(ious, 0) if want_iou else None)

    def _decode_chunk(self, image_embeddings, image_pe, sparse, dense_vec, roi_img, want_iou=True, src_rows=None, hw=None,
                      multimask_output=False):
        """image_embeddings [B,256,h,w] (logical NCHW, channels-last), image_pe [1|B,256,h,w] (input
        independent; batch entry 0 is used), sparse [R, n_pts, 256], dense_vec [256] (the broadcast
        `no_mask_embed`, models.py:1680), roi_img int32 [R] image index of every RoI (sorted).
        Returns low_res_masks [R, 1, 4h, 4w] (mask token 0) and iou [R, 1]; multimask_output=True: the masks of mask
        tokens 1..3, [R, 3, 4h, 4w], and iou [R, 3] (HF:537-542)."""
        if self._packed is None:
            self._pack()
        P = self._packed
        if src_rows is None:
            emb = nhwc_view(image_embeddings)
            B, h, w, C = emb.shape
            dev = emb.device
        else:
            # query variant: `image_embeddings + dense prompt` already formed per source (rsp_sam_mask_embed);
            # src_rows [Bs*h*w, 256], roi_img maps every prompt set to its source
            h, w = hw
            C = src_rows.shape[-1]
            B = src_rows.shape[0] // (h * w)
            dev = src_rows.device
        N = h * w
        R, npts = sparse.shape[0], sparse.shape[1]
        T = 1 + N_MASK_TOKENS + npts
        pe_rows = nhwc_view(image_pe[:1]).reshape(N, C)
        if (pe_rows.data_ptr(), tuple(pe_rows.shape)) not in self._pe_cache and image_pe.shape[0] > 1:
            # the image-wide positional embedding is ONE table repeated over the batch (models.py:85-95, 1685); its
            # projections are folded into broadcast terms below, so a caller with per-image tables must not get a
            # silently wrong answer (checked once per new table, not per step)
            if not bool((image_pe == image_pe[:1]).all()):
                raise NotImplementedError('image_positional_embeddings must be the same table for every batch entry')
        pe_t = self._pe_terms(pe_rows)
        d2, dh2 = HID // 2, (HID // 2) // HEADS

        # tokens = [iou, mask x4, sparse prompts] (HF:489-496)
        out_tok = torch.cat([self.iou_token.weight, self.mask_tokens.weight], 0)
        tokens0 = torch.cat([out_tok.unsqueeze(0).expand(R, -1, -1), sparse.reshape(R, npts, HID)], 1)
        tokens0 = tokens0.reshape(R * T, HID).contiguous()
        # keys of layer 0: image embedding + dense prompt, ONE copy per image (HF:499)
        if src_rows is None:
            src = ops.add_rows(emb.reshape(B * N, C), dense_vec.reshape(1, C), vmod=1)
        else:
            src = src_rows
        src_pl = ops.to_planes(src)

        # ---------------- layer 0 (HF:306-348 with skip_first_layer_pe) ----------------
        q = self._token_attn(tokens0, tokens0, tokens0, '0.self_attn', R, T)          # replaces queries
        q = self._ln(q, 'transformer.layers.0.layer_norm1')
        # tokens -> image
        qpe = ops.add_rows(q, tokens0)
        tq = ops.gemm(qpe, P['0.cross_attn_token_to_image.q_proj'])
        kv_img = ops.gemm(src_pl, P['0.cross_attn_token_to_image.kv_proj'], bias=None,
                          res=pe_t['0.cross_attn_token_to_image.kv_proj'], res_mod=N)    # per image, [K | V]
        ao = torch.empty_like(tq)
        self._t2i(tq, kv_img, ao, R, T, N, kv_map=roi_img)
        q = ops.gemm(ao, P['0.cross_attn_token_to_image.out_proj'], res=q)
        q = self._ln(q, 'transformer.layers.0.layer_norm2')
        hmid = ops.gemm(q, P['0.lin1'], act=ops.ACT_RELU)
        q = ops.gemm(hmid, P['0.lin2'], res=q)
        q = self._ln(q, 'transformer.layers.0.layer_norm3')
        # image -> tokens: image-side queries are per image, keys/values per RoI
        qpe = ops.add_rows(q, tokens0)
        qi = ops.gemm(src_pl, P['0.cross_attn_image_to_token.q_proj'], bias=None,
                      res=pe_t['0.cross_attn_image_to_token.q_proj'], res_mod=N)
        kt = ops.gemm(qpe, P['0.cross_attn_image_to_token.k_proj'])
        vt = ops.gemm(q, P['0.cross_attn_image_to_token.v_proj'])
        m4 = _g(self, 'transformer.layers.0.layer_norm4')
        fused = T <= ops.SAM_I2T_FUSED_MAX_TOKENS
        ai = None
        if fused:
            # attention + out_proj (folded into the values) + residual (the per-IMAGE src rows) + layer_norm4 in one
            # kernel: neither the [R*N, 128] attention output nor the fp32 [R*N, 256] keys are ever written
            op = _g(self, 'transformer.layers.0.cross_attn_image_to_token.out_proj')
            keys_pl = ops.sam_i2t_fused(qi, kt, vt, op.weight.detach(), op.bias.detach(), m4.weight, m4.bias, R=R, T=T,
                                        N=N, scale=dh2 ** -0.5, eps=1e-6, q_map=roi_img, res=src, res_map=roi_img)
        else:
            ai = ops.empty_planes((R * N, d2), dev)  # attention output goes straight to the out_proj GEMM as planes
            self._i2t(qi, kt, vt, ai, R, T, N, q_map=roi_img)
            keys = ops.gemm(ai, P['0.cross_attn_image_to_token.out_proj'], res=src, res_bmap=roi_img, res_brows=N)
            # layer_norm4 emits planes only: they are both the A operand of layer 1's projections and (hi + lo) the
            # residual of its out_proj GEMM, so the fp32 copy of the per-RoI keys is never written
            keys_pl = ops.layernorm(keys, m4.weight, m4.bias, 1e-6, planes=True, f32=False)   # [R*N, 256] planes
            del keys
        del qi, kv_img

        # ---------------- layer 1 ----------------
        qpe = ops.add_rows(q, tokens0)
        q = self._token_attn(qpe, qpe, q, '1.self_attn', R, T, res=q)
        q = self._ln(q, 'transformer.layers.1.layer_norm1')
        qpe = ops.add_rows(q, tokens0)
        tq = ops.gemm(qpe, P['1.cross_attn_token_to_image.q_proj'])
        # (the kernel addresses a key plane with 32-bit byte offsets: R * N * 512 < 2^31, i.e. 1023 RoIs at N = 4096)
        fold = self.t2i_fold and T <= ops.SAM_T2I_FOLD_MAX_TOKENS and N % 32 == 0 and R * N * 512 < 2 ** 31
        kv = None
        if fold:
            ao = self._t2i_folded('1.cross_attn_token_to_image', tq, keys_pl, pe_t, R, T, N)
        else:
            kv = ops.gemm(keys_pl, P['1.cross_attn_token_to_image.kv_proj'], bias=None,
                          res=pe_t['1.cross_attn_token_to_image.kv_proj'], res_mod=N)
            self._t2i(tq, kv, ao, R, T, N)
        q = ops.gemm(ao, P['1.cross_attn_token_to_image.out_proj'], res=q)
        q = self._ln(q, 'transformer.layers.1.layer_norm2')
        hmid = ops.gemm(q, P['1.lin1'], act=ops.ACT_RELU)
        q = ops.gemm(hmid, P['1.lin2'], res=q)
        q = self._ln(q, 'transformer.layers.1.layer_norm3')
        qpe = ops.add_rows(q, tokens0)
        qi = ops.gemm(keys_pl, P['1.cross_attn_image_to_token.q_proj'], bias=None,
                      res=pe_t['1.cross_attn_image_to_token.q_proj'], res_mod=N)
        kt = ops.gemm(qpe, P['1.cross_attn_image_to_token.k_proj'])
        vt = ops.gemm(q, P['1.cross_attn_image_to_token.v_proj'])
        m4 = _g(self, 'transformer.layers.1.layer_norm4')
        if fused:
            op = _g(self, 'transformer.layers.1.cross_attn_image_to_token.out_proj')
            keys_pl = ops.sam_i2t_fused(qi, kt, vt, op.weight.detach(), op.bias.detach(), m4.weight, m4.bias, R=R, T=T,
                                        N=N, scale=dh2 ** -0.5, eps=1e-6, res_planes=keys_pl)
        else:
            self._i2t(qi, kt, vt, ai, R, T, N)
            keys = ops.gemm(ai, P['1.cross_attn_image_to_token.out_proj'], res=keys_pl)
            keys_pl = ops.layernorm(keys, m4.weight, m4.bias, 1e-6, planes=True, f32=False)   # planes only from here on
            del keys

        # ---------------- final token -> image attention (HF:396-404; LayerNorm default eps 1e-5) ----
        qpe = ops.add_rows(q, tokens0)
        tq = ops.gemm(qpe, P['final.q_proj'])
        if fold:
            ao = self._t2i_folded('final', tq, keys_pl, pe_t, R, T, N)
        else:
            kv = ops.gemm(keys_pl, P['final.kv_proj'], bias=None, res=pe_t['final.kv_proj'], res_mod=N, out=kv)
            self._t2i(tq, kv, ao, R, T, N)
        q = ops.gemm(ao, P['final.out_proj'], res=q)
        q = self._ln(q, 'transformer.layer_norm_final_attn', eps=1e-5)
        del kv, qi, ai
        q3 = q.view(R, T, HID)
        stages = dict(keys=keys_pl, tokens=q3, up=None, hyper=[]) if self.keep_stages else None

        # ---------------- upscaling + hyper-network (HF:513-531) ----------------
        # mask token 0 is the only mask kept with multimask_output=False, tokens 1..3 otherwise (HF:537-542)
        toks = (1, 2, 3) if multimask_output else (0,)
        # both ConvTransposes run as one GEMM each (columns = (dy, dx, co), planes in); the second one never
        # stores its [R, 4h, 4w, 32] result: GELU and the product with hyper_in happen in its epilogue
        fused_up = self.upscale_fused and len(toks) == 1
        if not fused_up:
            up = ops.conv_transpose2x2(keys_pl.view(R, h, w, HID), *P['up1'], act=ops.ACT_GELU,
                                       ln=(self.upscale_layer_norm.weight, self.upscale_layer_norm.bias, 1e-6))
            del keys_pl                                                                 # [R, 2h, 2w, 64] planes
            if stages is not None:
                stages['up'] = up
        outs = []
        for i in toks:
            mt = q3[:, 1 + i, :].contiguous()
            hy = ops.gemm(mt, P[f'hyper{i}.proj_in'], act=ops.ACT_RELU)
            hy = ops.gemm(hy, P[f'hyper{i}.layers.0'], act=ops.ACT_RELU)
            hy = ops.gemm(hy, P[f'hyper{i}.proj_out'])
            if stages is not None:
                stages['hyper'].append(hy)
            if fused_up:
                # one pass over the keys: no [R, 2h, 2w, 64] intermediate (csrc/upscale.hip, sam_upscale_fused_kernel)
                outs.append(ops.sam_upscale_fused(keys_pl, P['up1'][0], P['up1'][1], self.upscale_layer_norm.weight,
                                                  self.upscale_layer_norm.bias, 1e-6, P['up2p'][0], P['up2p'][1], hy,
                                                  h, w).view(R, 1, 4 * h, 4 * w))
                continue
            # (multimask: the last ConvTranspose is recomputed per token -- a rarely used option, no [R,4h,4w,32] tensor)
            outs.append(ops.conv_transpose2x2(up, *P['up2'], act=ops.ACT_GELU, hyper=hy).view(R, 1, 4 * h, 4 * w))
        masks = outs[0] if len(outs) == 1 else torch.cat(outs, 1)
        self._last_stages = stages
        iou = None
        if want_iou:
            it = q3[:, 0, :].contiguous()
            io = ops.gemm(it, P['iou.proj_in'], act=ops.ACT_RELU)
            io = ops.gemm(io, P['iou.layers.0'], act=ops.ACT_RELU)
            io = ops.gemm(io, P['iou.proj_out'])
            iou = io[:, 1:4].contiguous() if multimask_output else io[:, 0:1]
        return masks, iou

    def forward(self, image_embeddings, image_positional_embeddings, sparse_prompt_embeddings,
                dense_prompt_embeddings, multimask_output=False, attention_similarity=None,
                target_embedding=None, output_attentions=None):
        """HF-compatible signature (HF:461-543) for callers that already repeated the image tensors per
        prompt set: every batch entry is treated as its own image.  point_batch_size 1, constant dense prompt;
        multimask_output=True returns the three masks of mask tokens 1..3 (HF:537-542)."""
        if attention_similarity is not None or target_embedding is not None:
            raise NotImplementedError('attention_similarity / target_embedding (HF SamAttention hooks) are not implemented')
        R = image_embeddings.shape[0]
        if sparse_prompt_embeddings.dim() == 4:
            if sparse_prompt_embeddings.shape[1] != 1:
                raise NotImplementedError('point_batch_size must be 1')
            sparse = sparse_prompt_embeddings[:, 0]
        else:
            sparse = sparse_prompt_embeddings
        d = nhwc_view(dense_prompt_embeddings)
        if d.shape[0] != 1 and not bool((d[0, 0, 0] == d[-1, -1, -1]).all()):
            raise NotImplementedError('per-pixel dense prompts go through decode_dense()')
        dense_vec = d[0, 0, 0].contiguous()
        roi_img = torch.arange(R, dtype=torch.int32, device=image_embeddings.device)
        masks, iou = self.decode(image_embeddings, image_positional_embeddings, sparse.contiguous(), dense_vec,
                                 roi_img, multimask_output=bool(multimask_output))
        return masks.unsqueeze(1), iou.unsqueeze(1), None


@MODELS.register_module()
class RSSamMaskDecoder(HIPModule):
    def __init__(self, hf_pretrain_name, extra_config=None, init_cfg=None):
        super().__init__()
        self.mask_decoder = SamMaskDecoderHIP()

    def forward(self, *args, **kwargs):
        return self.mask_decoder(*args, **kwargs)


def image_wide_table(G, size):
    """models.py:85-95 + HF:552-566 for the [2, F] Gaussian matrix G: logical [1, 2F, size, size] table on G's device
    (constant preparation on the host, exactly the reference's fp32 expression)."""
    g = G.detach().float().cpu()
    grid = torch.ones((size, size), dtype=torch.float32)
    y = (grid.cumsum(dim=0) - 0.5) / size
    x = (grid.cumsum(dim=1) - 0.5) / size
    c = torch.stack([x, y], dim=-1)
    c = 2 * c - 1
    c = c @ g
    c = 2 * np.pi * c
    pe = torch.cat([torch.sin(c), torch.cos(c)], dim=-1)          # [size, size, 2F] == NHWC
    return nchw_view(pe.unsqueeze(0).contiguous().to(G.device))


class _PosEmb(HIPModule):
    def __init__(self):
        super().__init__()
        add_param(self, 'positional_embedding', (2, 128))


@MODELS.register_module()
class RSSamPositionalEmbedding(HIPModule):
    """models.py:744-759.  The image-wide table (models.py:85-95 + HF:552-566) depends only on the
    [2,128] Gaussian matrix, so it is computed once per (size, weights) on the host at pack time
    (constant preparation, exactly the reference's fp32 expression) and kept on the device."""

    def __init__(self, hf_pretrain_name, extra_config=None, init_cfg=None):
        super().__init__()
        self.shared_image_embedding = _PosEmb()
        self._cache = {}

    def _apply(self, fn, *a, **kw):
        self._cache = {}
        return super()._apply(fn, *a, **kw)

    def image_wide(self, size):
        G = self.shared_image_embedding.positional_embedding
        key = (size, G.data_ptr(), G._version)
        if key not in self._cache:
            self._cache = {key: image_wide_table(G, size)}
        return self._cache[key]

    def forward(self, input_coords, input_shape=None):
        raise NotImplementedError('only the image-wide table is on the RSPrompter path (models.py:85-95)')


class _PromptEncoder(HIPModule):
    def __init__(self):
        super().__init__()
        add_param(self, 'no_mask_embed.weight', (1, HID))
        # SamMaskEmbedding (HF:569-593), used by the query variant's sam_mask_embed (models.py:305)
        add_param(self, 'mask_embed.conv1.weight', (4, 1, 2, 2))
        add_param(self, 'mask_embed.conv1.bias', (4,))
        add_param(self, 'mask_embed.conv2.weight', (16, 4, 2, 2))
        add_param(self, 'mask_embed.conv2.bias', (16,))
        add_param(self, 'mask_embed.conv3.weight', (HID, 16, 1, 1))
        add_param(self, 'mask_embed.conv3.bias', (HID,))
        _add_ln(self, 'mask_embed.layer_norm1', 4)
        _add_ln(self, 'mask_embed.layer_norm2', 16)


@MODELS.register_module()
class RSSamPromptEncoder(HIPModule):
    """models.py:881-896; only `no_mask_embed` / `mask_embed` are used downstream (:1635, :305)."""

    def __init__(self, hf_pretrain_name, extra_config=None, init_cfg=None):
        super().__init__()
        self.prompt_encoder = _PromptEncoder()

    def forward(self, *args, **kwargs):
        raise NotImplementedError('RSPrompter only borrows no_mask_embed / mask_embed from the prompt encoder')
