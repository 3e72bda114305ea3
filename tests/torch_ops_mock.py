"""TEST-ONLY stand-in for `rsprompter_amd.ops` built from plain torch CPU ops.

It lets the `-m "not gpu"` suite exercise the HOST orchestration (weight packing, row maps,
BN folding, window partition maps, decoder wiring ...) against the oracle without a GPU.
It is never imported by the package; the product path has no such fallback.
"""
import math

import torch
import torch.nn.functional as F

ACT_NONE, ACT_RELU, ACT_GELU, ACT_SIGMOID, ACT_RELU_POST = 0, 1, 2, 3, 4
DEFAULT_A_SCALE_LOG2 = 6
F8_CORR = True          # the flag only travels through the host wiring here (planes are fp32 tensors in the mock)


def require_device(dev):
    pass


def _act(x, act):
    return {0: lambda t: t, 1: F.relu, 2: F.gelu, 3: torch.sigmoid, 4: lambda t: t}[act](x)      # 4: after the residual


class PackedWeight:
    def __init__(self, w, bias=None, device=None, f8=False):
        self.w = w.detach().float()
        self.N, self.K = self.w.shape
        self.bias = None if bias is None else bias.detach().float()
        self.scale_log2 = 0


def gemm(a, w, *, out=None, bias='auto', res=None, act=0, a_rowmap=None, c_rowmap=None, M=None, out_rows=None,
         res_mod=0, a_scale_log2=6, conv=None, res_bmap=None, res_brows=0, out_planes=False, out_f32=True,
         dma='auto', tile_hint=0, c_ncols=0, pl_col0=0, out_f8=False):
    if conv is not None:
        k, s, p = conv
        B, H, W, C = a.shape
        wt = w.w.view(w.N, k, k, C).permute(0, 3, 1, 2)
        y = F.conv2d(a.permute(0, 3, 1, 2), wt, None, stride=s, padding=p).permute(0, 2, 3, 1)
        y = y.reshape(-1, w.N)
    else:
        src = a
        if a_rowmap is not None:
            idx = a_rowmap.long()
            src = torch.where((idx >= 0)[:, None], a[idx.clamp(min=0)], torch.zeros(1))
        y = src[:M] @ w.w.t() if M is not None else src @ w.w.t()
    if bias == 'auto':
        bias = w.bias
    if bias is not None:
        y = y + bias
    y = _act(y, act)
    m = y.shape[0]
    rows = torch.arange(m)
    crow = rows if c_rowmap is None else c_rowmap.long()
    keep = crow >= 0
    if out is None:
        out = torch.zeros((m if out_rows is None else out_rows, w.N))
    if res is not None:
        rr = crow.clamp(min=0)
        if res_mod > 0:
            rr = rr % res_mod
        if res_bmap is not None:
            rr = res_bmap.long()[rr // res_brows] * res_brows + rr % res_brows
        y = y + res[rr]
    if act == ACT_RELU_POST:
        y = F.relu(y)
    out[crow[keep]] = y[keep]
    if isinstance(out_planes, torch.Tensor):      # caller-owned plane tensor: rows nobody maps keep their contents
        out_planes[crow[keep]] = y[keep][:, pl_col0:]
        return out[:, :c_ncols].contiguous() if c_ncols else out, out_planes
    if c_ncols or pl_col0:       # column-range outputs: fp32 = first c_ncols columns, "planes" = columns from pl_col0 on
        return out[:, :c_ncols].contiguous(), out[:, pl_col0:].contiguous()
    return (out, out) if (out_planes and out_f32) else out


def layernorm(x, gamma, beta, eps=1e-6, act=0, out=None, planes=False, f32=True, f8=False):
    # the mock keeps "planes" as plain fp32 tensors: only the host wiring is under test
    y = _act(F.layer_norm(x, (x.shape[-1],), gamma, beta, eps), act)
    return (y, y) if (planes and f32) else y


def vit_relpos(qkv, rph, rpw, Bp, S, nh, dh, q_ld=None, rows=None):
    T = S * S
    if q_ld is not None and q_ld == nh * dh:
        q = qkv.view(Bp, T, nh, dh).permute(0, 2, 1, 3).reshape(Bp * nh, S, S, dh)
    else:
        q = qkv.view(Bp, T, 3, nh, dh)[:, :, 0].permute(0, 2, 1, 3).reshape(Bp * nh, S, S, dh)
    idx = torch.arange(S)[:, None] - torch.arange(S)[None, :] + (S - 1)
    rh = torch.einsum('bhwc,hkc->bhwk', q, rph[idx])
    rw = torch.einsum('bhwc,wkc->bhwk', q, rpw[idx])
    return torch.cat([rh, rw], -1).reshape(Bp * nh, T, 2 * S)


def vit_attention(qkv, rel, Bp, S, nh, dh, scale, planes=False):
    T = S * S
    q, k, v = qkv.view(Bp, T, 3, nh, dh).permute(2, 0, 3, 1, 4).reshape(3, Bp * nh, T, dh).unbind(0)
    attn = (q * scale) @ k.transpose(-2, -1)
    bias = rel[..., :S].reshape(-1, T, S, 1) + rel[..., S:].reshape(-1, T, 1, S)
    attn = (attn.view(-1, T, S, S) + bias).view(-1, T, T).softmax(-1)
    return (attn @ v).view(Bp, nh, T, dh).permute(0, 2, 1, 3).reshape(Bp * T, nh * dh)


def vit_attention_planes(q, kv, rel, Bp, S, nh, dh, scale, planes=False, f8=False, win_grid=None):
    T = S * S
    qkv = torch.cat([q.view(Bp * T, 1, nh * dh), kv.view(Bp * T, 2, nh * dh)], 1)
    return vit_attention(qkv.reshape(Bp * T, 3 * nh * dh), rel, Bp, S, nh, dh, scale)


def pack_relpos_tables(rph, rpw, S, dh):
    """stand-in: the two tables as they are (ops.pack_relpos_tables splits them into fp16 planes for the kernel)"""
    return (rph.float(), rpw.float(), S)


def vit_window_attention(q, kv, rel_tab, Bp, nh, dh, scale, planes=False, f8=False, win_grid=None, variant=0):
    rph, rpw, S = rel_tab
    rel = vit_relpos(q, rph, rpw, Bp, S, nh, dh, q_ld=nh * dh)
    return vit_attention_planes(q, kv, rel, Bp, S, nh, dh, scale)


def patchify(img, patch):
    return F.unfold(img, patch, stride=patch).transpose(1, 2).reshape(-1, img.shape[1] * patch * patch)


def conv_transpose2x2(x, w_dy, bias, act=0, a_scale_log2=6, out_planes=False, hyper=None, ln=None):
    B, H, W, C = x.shape
    four = not isinstance(w_dy, (tuple, list))
    if four:
        cout = w_dy.N // 4
        y = x.reshape(-1, C) @ w_dy.w.t()
        if bias is not None:
            y = y + bias
        y = y.view(B, H, W, 2, 2, cout)                               # [.., dy, dx, co]
        if ln is not None:
            y = F.layer_norm(y, (cout,), ln[0], ln[1], ln[2])
        y = _act(y, act)
        out = y.permute(0, 1, 3, 2, 4, 5).reshape(B, 2 * H, 2 * W, cout)
    else:
        cout = w_dy[0].N // 2
        out = torch.zeros(B, 2 * H, 2 * W, cout)
        for dy in (0, 1):
            y = x.reshape(-1, C) @ w_dy[dy].w.t()
            if bias is not None:
                y = y + bias
            y = _act(y, act).view(B, H, W, 2, cout)
            out[:, dy::2] = y.reshape(B, H, 2 * W, cout)
    if hyper is not None:
        return torch.einsum('bhwc,bc->bhw', out, hyper)
    return out


def empty_planes(shape, device, scale_log2=6, f8=False):
    return torch.zeros(shape)


def to_planes(x, scale_log2=6, f8=False):
    return x


def attention(q, k, v, out, *, B, nh, dh, Tq, Tk, scale, q_strides, k_strides, v_strides, o_strides,
              kv_batch_map=None, q_batch_map=None, out_planes=None, mask=None):
    if out is None:
        out = out_planes
    def view(t, st, T, bmap):
        nb = (int(bmap.max()) + 1) if bmap is not None else B
        tt = torch.as_strided(t, (nb, T, nh, dh), (st[0], st[1], st[2], 1))
        if bmap is not None:
            tt = tt[bmap.long()]
        return tt.permute(0, 2, 1, 3)
    qq, kk, vv = view(q, q_strides, Tq, q_batch_map), view(k, k_strides, Tk, kv_batch_map), view(v, v_strides, Tk, kv_batch_map)
    sc = (qq * scale) @ kk.transpose(-1, -2)
    if mask is not None:                                    # [B, Tq, Tk] bytes, non-zero = blocked, all heads
        sc = sc.masked_fill(mask.view(B, 1, Tq, Tk).bool(), float('-inf'))
    o = sc.softmax(-1) @ vv
    ov = torch.as_strided(out, (B, Tq, nh, dh), (o_strides[0], o_strides[1], o_strides[2], 1))
    ov.copy_(o.permute(0, 2, 1, 3))
    return out


SAM_T2I_MAX_TOKENS = 12
SAM_I2T_MAX_TOKENS = 16


def sam_t2i_attention(q, kv, out, *, R, T, N, scale, kv_map=None):
    qq = q.view(R, T, 8, 16).permute(0, 2, 1, 3)
    kvv = kv.view(-1, N, 2, 8, 16)
    if kv_map is not None:
        kvv = kvv[kv_map.long()]
    kk, vv = kvv[:, :, 0].permute(0, 2, 1, 3), kvv[:, :, 1].permute(0, 2, 1, 3)
    o = ((qq * scale) @ kk.transpose(-1, -2)).softmax(-1) @ vv
    out.view(R, T, 8, 16).copy_(o.permute(0, 2, 1, 3))
    return out


def sam_i2t_attention(q, k, v, *, R, T, N, scale, q_map=None, out=None, out_planes=None):
    qq = q.view(-1, N, 8, 16)
    if q_map is not None:
        qq = qq[q_map.long()]
    qq = qq.permute(0, 2, 1, 3)
    kk, vv = k.view(R, T, 8, 16).permute(0, 2, 1, 3), v.view(R, T, 8, 16).permute(0, 2, 1, 3)
    o = ((qq * scale) @ kk.transpose(-1, -2)).softmax(-1) @ vv
    dst = out_planes if out_planes is not None else out
    dst.view(R, N, 8, 16).copy_(o.permute(0, 2, 1, 3))
    return dst


SAM_I2T_FUSED_MAX_TOKENS = 10


def sam_i2t_fused(q, k, v, wo, bo, gamma, beta, *, R, T, N, scale, eps=1e-6, q_map=None, res=None, res_map=None,
                  res_planes=None, planes=True, f32=False):
    """LayerNorm(residual + out_proj(image -> token attention)) composed from the plain pieces (HF:340-348)"""
    att = torch.empty(R * N, 128)
    sam_i2t_attention(q, k, v, R=R, T=T, N=N, scale=scale, q_map=q_map, out=att)
    y = att @ wo.t() + bo
    if res is not None:
        rr = res.view(-1, N, 256)
        rr = rr[res_map.long()] if res_map is not None else rr
        y = y + rr.reshape(R * N, 256)
    else:
        y = y + res_planes.reshape(R * N, 256)
    y = F.layer_norm(y, (256,), gamma, beta, eps)
    return (y, y) if (planes and f32) else y


def add_rows(x, v, vmod=None, out=None):
    C = x.shape[-1]
    rows = x.numel() // C
    vmod = v.numel() // C if vmod is None else vmod
    return (x.reshape(rows, C) + v.reshape(-1, C)[torch.arange(rows) % vmod]).view(x.shape)


def sincos_pairs(x):
    return torch.sin(x[..., ::2]) + x[..., 1::2]


def hyper_mask(up, hyper):
    return torch.einsum('rpc,rc->rp', up, hyper)


def pool2(x, mode):
    y = x.permute(0, 3, 1, 2)
    y = F.max_pool2d(y, 2, 2) if mode == 0 else F.max_pool2d(y, 1, stride=2)
    return y.permute(0, 2, 3, 1).contiguous()


def roi_align(feats_nhwc, pes, rois, P, strides, finest_scale=56):
    """SingleRoIExtractor + mmcv RoIAlign through the oracle's C restatement; the extra PE is added to the level
    before sampling (sampling is linear, so this equals sampling feat + PE)."""
    from oracle import glue
    levels = []
    for i, f in enumerate(feats_nhwc):
        x = f.permute(0, 3, 1, 2).contiguous()
        if pes is not None and pes[i] is not None:
            x = x + pes[i].permute(2, 0, 1).unsqueeze(0)
        levels.append(x)
    out = glue.roi_extract(levels, rois, P, strides, finest_scale)        # [K, C, P, P]
    return out.permute(0, 2, 3, 1).contiguous()


def mask_post(low_res, batch_input_shape, crop_hw, out_hw, thr, want_prob=False):
    """models.py:1746-1784: sigmoid -> bilinear to batch_input_shape -> crop -> bilinear to ori_shape -> >= thr"""
    k = low_res.shape[0]
    if k == 0:
        m = torch.zeros((0, out_hw[0], out_hw[1]), dtype=torch.bool)
        return (m, torch.zeros((0, out_hw[0], out_hw[1]))) if want_prob else m
    p = torch.sigmoid(low_res)[:, None]
    p = F.interpolate(p, size=tuple(batch_input_shape), mode='bilinear', align_corners=False)
    p = p[..., :crop_hw[0], :crop_hw[1]]
    p = F.interpolate(p, size=tuple(out_hw), mode='bilinear', align_corners=False)[:, 0]
    m = p >= thr
    return (m, p) if want_prob else m


def div_boxes(boxes, sf4):
    return boxes / torch.tensor([float(v) for v in sf4])


def fill_bias_rows(bias, rows, N, out=None, planes=None, c_ncols=0, pl_col0=0):
    r = rows.long()
    if out is not None:
        out[r] = bias[:c_ncols or N]
    if planes is not None:
        planes[r] = bias[pl_col0:]


def scale_boxes(boxes, f4):
    return boxes * torch.tensor([float(v) for v in f4])


def mask_post_logits(low_res, img_shape, crop_hw, out_hw, thr=0.0, want_val=False):
    """models.py:1185-1206: bilinear to img_shape -> crop -> bilinear to ori_shape -> > thr (on the logits)"""
    if low_res.shape[0] == 0:
        return torch.zeros((0, out_hw[0], out_hw[1]), dtype=torch.bool)
    p = F.interpolate(low_res[:, None], size=tuple(img_shape), mode='bilinear', align_corners=False)
    p = p[..., :crop_hw[0], :crop_hw[1]]
    p = F.interpolate(p, size=tuple(out_hw), mode='bilinear', align_corners=False)[:, 0]
    return (p > thr, p) if want_val else p > thr


def resnet_stem(x, w_taps, bias):
    w = w_taps.t().reshape(64, 3, 7, 7)
    return F.relu(F.conv2d(x, w, bias, stride=2, padding=3)).permute(0, 2, 3, 1).contiguous()


def maxpool_nhwc(x, k=3, s=2, p=1):
    return F.max_pool2d(x.permute(0, 3, 1, 2), k, s, p).permute(0, 2, 3, 1).contiguous()


def upsample_nearest_add_(dst, src):
    dst += F.interpolate(src.permute(0, 3, 1, 2), size=dst.shape[1:3], mode='nearest').permute(0, 2, 3, 1)
    return dst


def sam_embed_boxes(boxes, gauss, pe_top_left, pe_bottom_right, input_size):
    """HF SamPromptEncoder._embed_boxes + SamPositionalEmbedding.forward"""
    c = (boxes + 0.5).reshape(-1, 2, 2).clone()
    c[..., 0] = c[..., 0] / input_size[1]
    c[..., 1] = c[..., 1] / input_size[0]
    c = (2 * c - 1) @ gauss
    c = 2 * math.pi * c
    e = torch.cat([torch.sin(c), torch.cos(c)], -1)
    e[:, 0] += pe_top_left.reshape(-1)
    e[:, 1] += pe_bottom_right.reshape(-1)
    return e


def preprocess(imgs, mean, std, swap_rb, pad_divisor=1, pad_value=0.0, device=None):
    from oracle import glue
    return glue.data_preprocess(list(imgs), list(mean), list(std), bool(swap_rb), pad_divisor, pad_value)


def _padded(rows, max_out, B):
    """per-image lists of (boxes, scores, ids, src) -> the fixed-shape dict the HIP NMS returns"""
    out = dict(boxes=torch.zeros(B, max_out, 4), scores=torch.zeros(B, max_out), ids=torch.zeros(B, max_out, dtype=torch.int32),
               src=torch.zeros(B, max_out, dtype=torch.int32), count=torch.zeros(B, dtype=torch.int32))
    for b, (bx, sc, ids, src) in enumerate(rows):
        n = bx.shape[0]
        out['boxes'][b, :n], out['scores'][b, :n] = bx, sc
        out['ids'][b, :n], out['src'][b, :n] = ids.to(torch.int32), src.to(torch.int32)
        out['count'][b] = n
    return out


def _coder_kwargs(coder):
    """delta2bbox keywords of a DeltaXYWHBBoxCoder facade (wh_ratio_clip is the class default everywhere)"""
    return dict(means=tuple(coder.means), stds=tuple(coder.stds), clip_border=coder.clip_border,
                add_ctr_clamp=coder.add_ctr_clamp, ctr_clamp=coder.ctr_clamp)


class RpnSelector:
    """rpn_head.py:134-304 through the oracle's restatement (pinned on the reference's own code)."""

    def __init__(self, base_anchors, strides, nms_pre, max_per_img, iou_thr, min_bbox_size, coder, device):
        self.coder = _coder_kwargs(coder)
        self.base, self.strides = base_anchors.float(), list(strides)
        self.A = self.base.shape[1]
        self.nms_pre, self.max_per_img, self.iou_thr, self.min_bbox_size = nms_pre, max_per_img, iou_thr, min_bbox_size

    def __call__(self, heads, sizes, ld, img_hw):
        from oracle import glue
        B, A = img_hw.shape[0], self.A
        priors = []
        for lvl, (H, W) in enumerate(sizes):            # anchor_generator.py grid_priors: (y, x, a) order
            s = self.strides[lvl]
            sx = torch.arange(W, dtype=torch.float32) * s
            sy = torch.arange(H, dtype=torch.float32) * s
            yy, xx = torch.meshgrid(sy, sx, indexing='ij')
            shifts = torch.stack([xx, yy, xx, yy], -1).reshape(-1, 1, 4)
            priors.append((shifts + self.base[lvl][None]).reshape(-1, 4))
        rows = []
        for b in range(B):
            cls = [h.view(B, H, W, ld)[b, :, :, :A].permute(2, 0, 1) for h, (H, W) in zip(heads, sizes)]
            reg = [h.view(B, H, W, ld)[b, :, :, A:5 * A].permute(2, 0, 1) for h, (H, W) in zip(heads, sizes)]
            r = glue.rpn_predict_single(cls, reg, priors, (int(img_hw[b, 0]), int(img_hw[b, 1])), nms_pre=self.nms_pre,
                                        max_per_img=self.max_per_img, iou_thr=self.iou_thr,
                                        min_bbox_size=self.min_bbox_size, coder=self.coder)
            rows.append((r['bboxes'], r['scores'], r['level_ids'], r['anchor_index']))
        return _padded(rows, self.max_per_img, B)


def bbox_post(head, ld, rois, roi_start, img_hw, num_classes, score_thr, coder, iou_thr, max_out,
              scale_factors=None):
    from oracle import glue
    B, nc = img_hw.shape[0], num_classes
    rows = []
    for b in range(B):
        r0, r1 = int(roi_start[b]), int(roi_start[b + 1])
        if r1 == r0:
            rows.append((torch.zeros(0, 4), torch.zeros(0), torch.zeros(0, dtype=torch.long), torch.zeros(0, dtype=torch.long)))
            continue
        dets, labels, cand = glue.bbox_head_predict_single(rois[r0:r1], head[r0:r1, :nc + 1].contiguous(),
                                                           head[r0:r1, nc + 1:5 * nc + 1].contiguous(),
                                                           (int(img_hw[b, 0]), int(img_hw[b, 1])), nc, score_thr, iou_thr,
                                                           max_out, coder=_coder_kwargs(coder),
                                                           scale_factor=None if scale_factors is None else scale_factors[b])
        rows.append((dets[:, :4], dets[:, 4], labels, cand))
    return _padded(rows, max_out, B)


# ----------------------------------------------------------------------------- query prompter ops
class PlaneWeight:
    """rows [r0, r0+n) of an activation used as a GEMM weight (planes are plain fp32 tensors in the mock)"""

    def __init__(self, planes, r0=0, n=None):
        n = planes.shape[0] - r0 if n is None else n
        self.w = planes.reshape(planes.shape[0], -1)[r0:r0 + n].float()
        self.N, self.K = self.w.shape
        self.bias = None
        self.scale_log2 = 0


def groupnorm(x, gamma, beta, groups, eps=1e-5, relu=False, add=None):
    B, C = x.shape[0], x.shape[-1]
    y = F.group_norm(x.reshape(B, -1, C).transpose(1, 2), groups, gamma, beta, eps).transpose(1, 2).reshape(x.shape).contiguous()
    if add is not None:
        y = y + add
    return F.relu(y) if relu else y


def resize_bilinear(x_nhwc, size):
    y = F.interpolate(x_nhwc.permute(0, 3, 1, 2), size=tuple(size), mode='bilinear', align_corners=False)
    return y.permute(0, 2, 3, 1).contiguous()


def msdeform_attn(value, offs_weights, ref_points, B, Ntok, level_hw, head_dim=16):
    """mmcv MultiScaleDeformableAttention core (8 heads, 4 points) through the oracle's sampling restatement"""
    from oracle.query import MSDeformAttn
    H, L, P = 8, len(level_hw), 4
    v = value.view(B, Ntok, H, -1)
    off = offs_weights[:, :H * L * P * 2].reshape(B, Ntok, H, L, P, 2)
    w = offs_weights[:, H * L * P * 2:H * L * P * 3].reshape(B, Ntok, H, L * P).softmax(-1).view(B, Ntok, H, L, P)
    shapes = torch.tensor([list(hw) for hw in level_hw])
    norm = torch.stack([shapes[:, 1], shapes[:, 0]], -1).float()
    loc = ref_points[None, :, None, None, None, :] + off / norm[None, None, None, :, None, :]
    return MSDeformAttn._sample(v, shapes, loc, w).reshape(B * Ntok, -1)


def query_attn_mask(mask_pred_plus, size):
    """models.py:381-392 + :439-442 -> uint8 [B, Nq, h*w], 1 = blocked, fully blocked rows cleared"""
    am = F.interpolate(mask_pred_plus, tuple(size), mode='bilinear', align_corners=False).flatten(2).sigmoid() < 0.5
    am = am & (am.sum(-1) != am.shape[-1]).unsqueeze(-1)
    return am.to(torch.uint8)


def sam_mask_embed(mask_pred_plus, emb_rows, roi_img, prm, he, we, eps=1e-6):
    """SamMaskEmbedding (HF:569-601) on mask_pred_plus + the image embedding of the prompt set's image"""
    R = mask_pred_plus.shape[0]
    C = emb_rows.shape[-1]

    def ln2d(x, w, b):
        return F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), w, b, eps).permute(0, 3, 1, 2)
    x = mask_pred_plus[:, None]
    x = F.gelu(ln2d(F.conv2d(x, prm['conv1_w'], prm['conv1_b'], stride=2), prm['ln1_w'], prm['ln1_b']))
    x = F.gelu(ln2d(F.conv2d(x, prm['conv2_w'], prm['conv2_b'], stride=2), prm['ln2_w'], prm['ln2_b']))
    dense = F.conv2d(x, prm['conv3_w'].view(C, 16, 1, 1), prm['conv3_b'])                 # [R, C, he, we]
    src = emb_rows.view(-1, he * we, C)[roi_img.long()] + dense.permute(0, 2, 3, 1).reshape(R, he * we, C)
    return src.reshape(R * he * we, C)


def query_topk(cls, k):
    """softmax[:, :-1] -> top-k over (query, class), canonical order (score desc, flat index asc)"""
    B = cls.shape[0]
    sc = cls.softmax(-1)[..., :-1].reshape(B, -1)
    ranked, order = sc.sort(dim=1, descending=True, stable=True)
    return ranked[:, :k].contiguous(), order[:, :k].to(torch.int32).contiguous()


def query_mask_post(low_res, qidx, cls_score, batch_input_shape, crop_hw, out_hw, want_logits=False):
    """models.py:652-656 + 684-695 + maskformer_fusion_head.py:164-176 for the selected queries of one image"""
    from oracle.query import mask2bbox
    m = low_res[qidx.long()][:, None]
    m = F.interpolate(m, size=tuple(batch_input_shape), mode='bilinear', align_corners=False)
    m = m[..., :crop_hw[0], :crop_hw[1]]
    if tuple(out_hw) != tuple(crop_hw):
        m = F.interpolate(m, size=tuple(out_hw), mode='bilinear', align_corners=False)
    m = m[:, 0]
    binary = (m > 0).float()
    ms = (m.sigmoid() * binary).flatten(1).sum(1) / (binary.flatten(1).sum(1) + 1e-6)
    masks = binary.bool()
    out = (masks, cls_score * ms, mask2bbox(masks))
    return out + (m,) if want_logits else out


def resize_pad(img_hwc, new_hw, pad_hw, pad_val=(0.0, 0.0, 0.0), out=None, normalise=None):
    """stand-in of ops.resize_pad through the oracle's cv2 / mmcv restatement (oracle/pipeline.py)."""
    import numpy as np
    from oracle import pipeline as op
    assert normalise is None
    img = img_hwc.cpu().numpy().astype(np.float32)
    res = op.cv2_resize_linear_f32(img, int(new_hw[1]), int(new_hw[0]))
    canvas = np.empty((int(pad_hw[0]), int(pad_hw[1]), 3), dtype=np.float32)
    canvas[...] = np.asarray(pad_val, dtype=np.float32)
    canvas[:res.shape[0], :res.shape[1]] = res
    return torch.from_numpy(np.ascontiguousarray(canvas.transpose(2, 0, 1)))


def paste_masks(logits_nhwc, labels, boxes, img_hw, thr=0.5):
    from oracle import samseg
    k = logits_nhwc.shape[0]
    lg = logits_nhwc.permute(0, 3, 1, 2)
    if labels is not None and lg.shape[1] > 1:
        lg = lg[range(k), labels.long()][:, None]
    meta = dict(ori_shape=(int(img_hw[0]), int(img_hw[1])), scale_factor=(1.0, 1.0))
    masks, _, probs = samseg.fcn_predict_single(lg, boxes, None, meta, thr, rescale=True, class_agnostic=True)
    return masks if thr >= 0 else (probs * 255).to(torch.uint8)


# ----------------------------------------------------------------------------- the fused decoder forms (product since round 5)
SAM_T2I_FOLD_MAX_TOKENS = 12


def sam_t2i_fold(keys, pek, qp, tqx, *, R, N, ncols):
    """u[r * 96 + c] = sum_n softmax_n(keys[r, n] . qp[r * 96 + c] + pek[n] . tqx[r * 96 + c]) keys[r, n] (rsp_sam_t2i_fold; the
    softmax scale travels inside tqx / qp); rows of the columns >= ncols stay zero"""
    k = keys[:R * N].view(R, N, 256)
    s = torch.einsum('rnd,rcd->rcn', k, qp[:R * 96].view(R, 96, 256)) + torch.einsum('nd,rcd->rcn', pek, tqx[:R * 96].view(R, 96, 128))
    u = torch.einsum('rcn,rnd->rcd', s.softmax(-1), k)
    u[:, ncols:] = 0
    return u.reshape(R * 96, 256)


def sam_fold_expand(tq, R, T, scale):
    """tq [R*T, 128] -> block-diagonal [R*96, 128]: row r*96 + h*T + t = scale * tq[r, t, head h] in columns 16 h .. (rsp_sam_fold_expand)"""
    out = torch.zeros((R, 96, 8, 16), dtype=tq.dtype)
    t4 = tq.view(R, T, 8, 16) * scale
    for h in range(8):
        out[:, h * T:(h + 1) * T, h] = t4[:, :, h]
    return out.view(R * 96, 128)


def sam_fold_gather(full, R, T):
    """full [R*96, 128] -> [R*T, 128]: column h*T + t's own head (rsp_sam_fold_gather)"""
    f4 = full.view(R, 96, 8, 16)
    ao = torch.empty((R, T, 8, 16), dtype=full.dtype)
    for h in range(8):
        ao[:, :, h] = f4[:, h * T:(h + 1) * T, h]
    return ao.reshape(R * T, 128)


def sam_upscale_fused(x, w1, bias1, gamma, beta, eps, w2p, bias2, hyper, h, w):
    """ConvT(256 -> 64) + LN2d + GELU + ConvT(64 -> 32) + GELU + <., hyper> (rsp_sam_upscale_fused); w2p's K columns are in
    sam_decoder._upscale2_k_order()"""
    from rsprompter_amd.sam_decoder import _upscale2_k_order
    rows = x.shape[0]
    R = rows // (h * w)
    y = (x @ w1.w.t() + bias1).view(rows, 4, 64)                              # [(dy, dx), co]
    y = F.gelu(F.layer_norm(y, (64,), gamma, beta, eps))
    z = F.gelu(y[..., _upscale2_k_order()] @ w2p.w.t() + bias2).view(rows, 2, 2, 2, 2, 32)     # [dy, dx, dy2, dx2, c2]
    m = torch.einsum('nabcdk,nk->nabcd', z, hyper.repeat_interleave(h * w, 0))
    return m.view(R, h, w, 2, 2, 2, 2).permute(0, 1, 3, 5, 2, 4, 6).reshape(R, 4 * h, 4 * w)

