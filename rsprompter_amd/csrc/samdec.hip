// SAM mask-decoder tail + mask post-processing (HBM-bound kernels).
//   hyper-network mask product  HF:523-531  (masks = hyper_in @ upscaled_embedding; only mask
//                               token 0 is kept because multimask_output=False, HF:537-542)
//   mask post-process           models.py:1746-1784 (sigmoid -> bilinear to batch_input_shape ->
//                               crop -> bilinear to ori_shape -> >= thr)
#include "rsp_common.h"

namespace {

// out[r, pix] = sum_c up[r, pix, c] * hyper[r, c];  8 lanes per pixel (C == 32: one float4 each)
__global__ __launch_bounds__(256) void hyper_mask_kernel(const float* __restrict__ up,
                                                         const float* __restrict__ hyper,
                                                         float* __restrict__ out, int npix, int C) {
  const int r = blockIdx.y;
  const int part = threadIdx.x & 7;
  const float* hv = hyper + (int64_t)r * C;
  const float* ur = up + (int64_t)r * npix * C;
  for (int pix = blockIdx.x * 32 + (threadIdx.x >> 3); pix < npix; pix += gridDim.x * 32) {
    float acc = 0.f;
    for (int c = part * 4; c < C; c += 32) {
      const f32x4 u = *reinterpret_cast<const f32x4*>(ur + (int64_t)pix * C + c);
      const f32x4 h = *reinterpret_cast<const f32x4*>(hv + c);
      acc += u[0] * h[0] + u[1] * h[1] + u[2] * h[2] + u[3] * h[3];
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (part == 0) out[(int64_t)r * npix + pix] = acc;
  }
}

struct Lin { int i0, i1; float l0, l1; };
// torch upsample_bilinear2d(align_corners=False) source index / weights
__device__ __forceinline__ Lin lin_coef(int dst, float scale, int in_size) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  Lin c;
  c.i0 = (int)src;
  if (c.i0 > in_size - 1) c.i0 = in_size - 1;
  c.i1 = c.i0 + (c.i0 < in_size - 1 ? 1 : 0);
  c.l1 = src - (float)c.i0;
  c.l0 = 1.0f - c.l1;
  return c;
}

struct MaskPostP {
  const float* low;   // [k, h, w] logits
  uint8_t* out;       // [k, oh, ow] bool
  float* prob;        // optional [k, oh, ow]
  int k, h, w, Hb, Wb, ch, cw, oh, ow;
  float thr;
  int strict;         // 1: value > thr (SAMDet, models.py:1206), 0: value >= thr (models.py:1779)
};

__global__ __launch_bounds__(256) void sigmoid_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = 1.0f / (1.0f + expf(-x[i]));
}

// `sig` = sigmoid(low-res logits) (models.py:1758); IDENT: crop == output size, so the second
// interpolation is the identity (src index == dst index, weight 1) and only stage 1 is evaluated.
template <bool IDENT>
__global__ __launch_bounds__(256) void mask_post_kernel(const MaskPostP p) {
  const int m = blockIdx.y;
  const float* low = p.low + (int64_t)m * p.h * p.w;
  const float s1h = (float)p.h / (float)p.Hb, s1w = (float)p.w / (float)p.Wb;
  const float s2h = (float)p.ch / (float)p.oh, s2w = (float)p.cw / (float)p.ow;
  const int64_t total = (int64_t)p.oh * p.ow;
  auto stage1 = [&](int Y, int X) -> float {
    const Lin ay = lin_coef(Y, s1h, p.h);
    const Lin ax = lin_coef(X, s1w, p.w);
    const float v00 = low[ay.i0 * p.w + ax.i0], v01 = low[ay.i0 * p.w + ax.i1];
    const float v10 = low[ay.i1 * p.w + ax.i0], v11 = low[ay.i1 * p.w + ax.i1];
    return ay.l0 * (ax.l0 * v00 + ax.l1 * v01) + ay.l1 * (ax.l0 * v10 + ax.l1 * v11);
  };
  auto pixel = [&](int oy, int ox) -> float {
    if (IDENT) return stage1(oy, ox);
    const Lin cy = lin_coef(oy, s2h, p.ch), cx = lin_coef(ox, s2w, p.cw);
    const float a00 = stage1(cy.i0, cx.i0), a01 = stage1(cy.i0, cx.i1);
    const float a10 = stage1(cy.i1, cx.i0), a11 = stage1(cy.i1, cx.i1);
    return cy.l0 * (cx.l0 * a00 + cx.l1 * a01) + cy.l1 * (cx.l0 * a10 + cx.l1 * a11);
  };
  if ((p.ow & 3) == 0) {
    // four pixels of one row per thread: one 32-bit store of the bool mask instead of four byte stores
    const int qw = p.ow >> 2;
    const int nq = p.oh * qw;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += gridDim.x * blockDim.x) {
      const int oy = i / qw, ox = (i - oy * qw) << 2;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = pixel(oy, ox + e);
      uint32_t bits = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) bits |= ((p.strict ? v[e] > p.thr : v[e] >= p.thr) ? 1u : 0u) << (8 * e);
      const int64_t o = (int64_t)m * total + (int64_t)oy * p.ow + ox;
      *reinterpret_cast<uint32_t*>(p.out + o) = bits;
      if (p.prob) *reinterpret_cast<f32x4*>(p.prob + o) = f32x4{v[0], v[1], v[2], v[3]};
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int oy = (int)(i / p.ow), ox = (int)(i - (int64_t)oy * p.ow);
    const float val = pixel(oy, ox);
    p.out[(int64_t)m * total + i] = (p.strict ? val > p.thr : val >= p.thr) ? 1 : 0;
    if (p.prob) p.prob[(int64_t)m * total + i] = val;
  }
}

// The identity-crop case (crop == output size: every 1024-px tile) as a strip kernel (round 5): a thread owns 4 consecutive
// output columns and walks MP_ROWS output rows.  The x coefficients are computed once, the two horizontally interpolated
// source rows only when the source row pair changes (every 4th output row at the usual 256 -> 1024), and a pixel is
// ay.l0 * h0 + ay.l1 * h1 -- the SAME fp32 expression tree as stage1() above, so the masks are bit-identical; the generic
// kernel spends ~40 VALU instructions and four gathers per pixel on it (1.3 ms per ViT-H step for 838 MB of masks).
constexpr int MP_ROWS = 16;
__global__ __launch_bounds__(256) void mask_post_strip_kernel(const MaskPostP p) {
  const int m = blockIdx.y;
  const float* low = p.low + (int64_t)m * p.h * p.w;
  const float s1h = (float)p.h / (float)p.Hb, s1w = (float)p.w / (float)p.Wb;
  const int qw = p.ow >> 2;
  const int ntile = (p.oh + MP_ROWS - 1) / MP_ROWS;
  const int64_t total = (int64_t)p.oh * p.ow;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;       // (row tile, column quad): quads fastest -> coalesced rows
  if (i >= ntile * qw) return;
  const int ty = i / qw, ox = (i - ty * qw) << 2;
  Lin ax[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) ax[e] = lin_coef(ox + e, s1w, p.w);
  int r0 = -1, r1 = -1;
  float h0[4], h1[4];
  const int oy_end = min((ty + 1) * MP_ROWS, p.oh);
  for (int oy = ty * MP_ROWS; oy < oy_end; ++oy) {
    const Lin ay = lin_coef(oy, s1h, p.h);
    if (ay.i0 != r0 || ay.i1 != r1) {
      r0 = ay.i0; r1 = ay.i1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        h0[e] = ax[e].l0 * low[r0 * p.w + ax[e].i0] + ax[e].l1 * low[r0 * p.w + ax[e].i1];
        h1[e] = ax[e].l0 * low[r1 * p.w + ax[e].i0] + ax[e].l1 * low[r1 * p.w + ax[e].i1];
      }
    }
    float v[4];
    uint32_t bits = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = ay.l0 * h0[e] + ay.l1 * h1[e];
      bits |= ((p.strict ? v[e] > p.thr : v[e] >= p.thr) ? 1u : 0u) << (8 * e);
    }
    const int64_t o = (int64_t)m * total + (int64_t)oy * p.ow + ox;
    *reinterpret_cast<uint32_t*>(p.out + o) = bits;
    if (p.prob) *reinterpret_cast<f32x4*>(p.prob + o) = f32x4{v[0], v[1], v[2], v[3]};
  }
}

}  // namespace

extern "C" int rsp_hyper_mask(const float* up, const float* hyper, float* out, int32_t R, int32_t npix,
                              int32_t C, rsp_stream_t stream) {
  if (!up || !hyper || !out || R < 0 || npix <= 0 || C <= 0 || (C & 3)) return RSP_EINVAL;
  if (R == 0) return RSP_OK;
  int gx = (npix + 31) / 32;
  if (gx > 512) gx = 512;
  hipLaunchKernelGGL(hyper_mask_kernel, dim3(gx, R), dim3(256), 0, (hipStream_t)stream, up, hyper, out, npix, C);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

namespace {
int launch_mask_post(const MaskPostP& p, hipStream_t stream) {
  if ((int64_t)p.oh * p.ow > 0x7fffffffLL) return RSP_EINVAL;
  int64_t gx = ((int64_t)p.oh * p.ow / ((p.ow & 3) == 0 ? 4 : 1) + 255) / 256;
  if (gx > 4096) gx = 4096;
  if (p.ch == p.oh && p.cw == p.ow && (p.ow & 3) == 0) {
    const int64_t nthr = (int64_t)((p.oh + MP_ROWS - 1) / MP_ROWS) * (p.ow >> 2);
    hipLaunchKernelGGL(mask_post_strip_kernel, dim3((unsigned)((nthr + 255) / 256), p.k), dim3(256), 0, stream, p);
  } else if (p.ch == p.oh && p.cw == p.ow)
    hipLaunchKernelGGL((mask_post_kernel<true>), dim3((unsigned)gx, p.k), dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL((mask_post_kernel<false>), dim3((unsigned)gx, p.k), dim3(256), 0, stream, p);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}
}  // namespace

extern "C" int rsp_mask_post(const float* low_res, float* sig_ws, int32_t k, int32_t h, int32_t w, int32_t Hb,
                             int32_t Wb, int32_t crop_h, int32_t crop_w, int32_t out_h, int32_t out_w, float thr,
                             uint8_t* out_mask, float* out_prob, rsp_stream_t stream) {
  if (!low_res || !sig_ws || !out_mask || k < 0 || h <= 0 || w <= 0 || Hb <= 0 || Wb <= 0 || crop_h <= 0 || crop_w <= 0 ||
      crop_h > Hb || crop_w > Wb || out_h <= 0 || out_w <= 0)
    return RSP_EINVAL;
  if (k == 0) return RSP_OK;
  {
    const int64_t n = (int64_t)k * h * w;
    int64_t gs = (n + 255) / 256;
    if (gs > 4096) gs = 4096;
    hipLaunchKernelGGL(sigmoid_kernel, dim3((unsigned)gs), dim3(256), 0, (hipStream_t)stream, low_res, sig_ws, n);
  }
  MaskPostP p;
  p.low = sig_ws; p.out = out_mask; p.prob = out_prob; p.k = k; p.h = h; p.w = w; p.Hb = Hb; p.Wb = Wb;
  p.ch = crop_h; p.cw = crop_w; p.oh = out_h; p.ow = out_w; p.thr = thr; p.strict = 0;
  return launch_mask_post(p, (hipStream_t)stream);
}

// SAMDet.predict (models.py:1185-1206): the same resize -> crop -> resize chain on the raw logits, then `> thr` (thr = 0)
extern "C" int rsp_mask_post_logits(const float* low_res, int32_t k, int32_t h, int32_t w, int32_t Hb, int32_t Wb,
                                    int32_t crop_h, int32_t crop_w, int32_t out_h, int32_t out_w, float thr,
                                    uint8_t* out_mask, float* out_val, rsp_stream_t stream) {
  if (!low_res || !out_mask || k < 0 || h <= 0 || w <= 0 || Hb <= 0 || Wb <= 0 || crop_h <= 0 || crop_w <= 0 ||
      crop_h > Hb || crop_w > Wb || out_h <= 0 || out_w <= 0)
    return RSP_EINVAL;
  if (k == 0) return RSP_OK;
  MaskPostP p;
  p.low = low_res; p.out = out_mask; p.prob = out_val; p.k = k; p.h = h; p.w = w; p.Hb = Hb; p.Wb = Wb;
  p.ch = crop_h; p.cw = crop_w; p.oh = out_h; p.ow = out_w; p.thr = thr; p.strict = 1;
  return launch_mask_post(p, (hipStream_t)stream);
}


// ---------------------------------------------------------------------------------------------------------------
// FCNMaskHead._predict_by_feat_single + _do_paste_mask (mmdet/models/roi_heads/mask_heads/fcn_mask_head.py:276-480),
// the mask post-processing of the standard Mask R-CNN head the SAMSeg sibling models use (models.py:1219-1244):
// sigmoid of the (class-selected) 28x28 logits, F.grid_sample(bilinear, zeros, align_corners=False) of the probability
// map at every pixel centre of the image with the box as the sampling window, >= thr.  One thread per 4 output pixels.
namespace {

__global__ __launch_bounds__(256) void paste_masks_kernel(const float* __restrict__ logits, const int32_t* __restrict__ labels,
                                                          const float* __restrict__ boxes, int Hm, int Wm, int C,
                                                          int img_h, int img_w, float thr, uint8_t* __restrict__ out) {
  const int n = blockIdx.y;
  const int cls = (labels && C > 1) ? labels[n] : 0;
  const float x0 = boxes[n * 4 + 0], y0 = boxes[n * 4 + 1], x1 = boxes[n * 4 + 2], y1 = boxes[n * 4 + 3];
  const float* m = logits + (int64_t)n * Hm * Wm * C + cls;          // NHWC: pixel stride C
  // the reference's CPU path pastes one instance per chunk with skip_empty=True (:390-404, :452-461): only the region
  // [floor(x0) - 1, ceil(x1) + 1) x [floor(y0) - 1, ceil(y1) + 1), clamped to the image, is written (the rest stays 0)
  const int rx0 = (int)fmaxf(floorf(x0) - 1.0f, 0.0f), ry0 = (int)fmaxf(floorf(y0) - 1.0f, 0.0f);
  const int rx1 = (int)fminf(ceilf(x1) + 1.0f, (float)img_w), ry1 = (int)fminf(ceilf(y1) + 1.0f, (float)img_h);
  const int64_t npix4 = ((int64_t)img_h * img_w + 3) / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix4; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t word = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t pix = i * 4 + e;
      if (pix >= (int64_t)img_h * img_w) break;
      const int py = (int)(pix / img_w), px = (int)(pix - (int64_t)py * img_w);
      if (px < rx0 || px >= rx1 || py < ry0 || py >= ry1) continue;
      // normalised coordinates exactly as the reference forms them: (p + 0.5 - lo) / (hi - lo) * 2 - 1, inf -> 0
      float gx = ((float)px + 0.5f - x0) / (x1 - x0) * 2.0f - 1.0f;
      float gy = ((float)py + 0.5f - y0) / (y1 - y0) * 2.0f - 1.0f;
      if (isinf(gx)) gx = 0.f;
      if (isinf(gy)) gy = 0.f;
      const float ix = ((gx + 1.0f) * (float)Wm - 1.0f) / 2.0f, iy = ((gy + 1.0f) * (float)Hm - 1.0f) / 2.0f;
      const float fx = floorf(ix), fy = floorf(iy);
      const int xw = (int)fx, yn = (int)fy;
      const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
      auto prob = [&](int yy, int xx) -> float {
        if (xx < 0 || xx >= Wm || yy < 0 || yy >= Hm) return 0.f;
        const float l = m[((int64_t)yy * Wm + xx) * C];
        return 1.0f / (1.0f + expf(-l));
      };
      float v = prob(yn, xw) * (wx0 * wy0);
      v += prob(yn, xw + 1) * (wx1 * wy0);
      v += prob(yn + 1, xw) * (wx0 * wy1);
      v += prob(yn + 1, xw + 1) * (wx1 * wy1);
      // thr < 0 (:390-394): the probability itself as (p * 255) truncated to uint8
      const uint32_t byte = thr >= 0.f ? (uint32_t)(v >= thr) : (uint32_t)(uint8_t)(v * 255.0f);
      word |= byte << (8 * e);
    }
    uint8_t* o = out + (int64_t)n * img_h * img_w + i * 4;
    if (i * 4 + 3 < (int64_t)img_h * img_w && ((((int64_t)n * img_h * img_w) & 3) == 0)) {
      *reinterpret_cast<uint32_t*>(o) = word;
    } else {
      for (int e = 0; e < 4 && i * 4 + e < (int64_t)img_h * img_w; ++e) o[e] = (word >> (8 * e)) & 0xffu;
    }
  }
}

}  // namespace

extern "C" int rsp_paste_masks(const float* logits, const int32_t* labels, const float* boxes, int32_t k, int32_t Hm,
                               int32_t Wm, int32_t C, int32_t img_h, int32_t img_w, float thr, uint8_t* out,
                               rsp_stream_t stream) {
  if (!logits || !boxes || !out || k < 0 || Hm <= 0 || Wm <= 0 || C <= 0 || img_h <= 0 || img_w <= 0) return RSP_EINVAL;
  if (k == 0) return RSP_OK;
  const int64_t npix4 = ((int64_t)img_h * img_w + 3) / 4;
  const unsigned gx = (unsigned)((npix4 + 255) / 256 > 1024 ? 1024 : (npix4 + 255) / 256);
  hipLaunchKernelGGL(paste_masks_kernel, dim3(gx, (unsigned)k), dim3(256), 0, (hipStream_t)stream, logits, labels, boxes, Hm, Wm,
                     C, img_h, img_w, thr, out);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Host-glue kernels of the folded token -> image attention (sam_decoder.py::_t2i_folded, csrc/t2i_fold.hip).  Round 5 built
// the block-diagonal query with torch (mul, permute, zeros, index_put, split: 6 launches) and picked every column's own head
// out of the v_proj result with an advanced index + permute (4 launches), twice per decoder call.
//   expand: tq [R*T, 128] (projected queries, head h at columns 16 h ..) -> fp16 planes of the block-diagonal matrix
//           [R*96, 128]: row r*96 + h*T + t holds scale * tq[r, t, 16 h .. 16 h + 15] in columns 16 h .., zeros elsewhere;
//           rows of columns >= 8 T are zero
//   gather: full [R*96, 128] (v_proj applied to every column) -> ao [R*T, 128] with ao[r*T + t, 16 h + d] = full[r*96 + h*T + t, 16 h + d]
namespace {
__global__ __launch_bounds__(256) void sam_fold_expand_kernel(const float* __restrict__ tq, half_t* __restrict__ hi, half_t* __restrict__ lo,
                                                                int64_t rows, int T, float scale, float pscale) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;        // one thread = 4 consecutive columns of one output row
  if (i >= rows * 32) return;
  const int64_t row = i >> 5;
  const int c = (int)(i & 31) * 4;
  const int64_t r = row / 96;
  const int col = (int)(row - r * 96);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (col < 8 * T) {
    const int h = col / T, t = col - h * T;
    if ((c >> 4) == h) {
      v = *reinterpret_cast<const f32x4*>(tq + (r * T + t) * 128 + c);
    }
  }
  // element by element: `v * scale` on the vector type becomes v_pk_mul_f32 with the scalar cross-selected through op_sel,
  // the instruction class DESIGN 9.1 keeps out of every kernel that shares a SIMD (tests/test_isa_guard_cpu.py)
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = (v[e] * scale) * pscale;
  rsp_store_planes4(hi, lo, ((int64_t)(c >> 5) * rows + row) * 32 + (c & 31), v, false);
}

__global__ __launch_bounds__(256) void sam_fold_gather_kernel(const float* __restrict__ full, float* __restrict__ ao, int64_t n4, int T) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;        // 4 consecutive columns of one (RoI, token) row
  if (i >= n4) return;
  const int64_t row = i >> 5;                                        // r * T + t
  const int c = (int)(i & 31) * 4;
  const int64_t r = row / T;
  const int t = (int)(row - r * T), h = c >> 4;
  *reinterpret_cast<f32x4*>(ao + row * 128 + c) = *reinterpret_cast<const f32x4*>(full + (r * 96 + h * T + t) * 128 + c);
}
}  // namespace

extern "C" int rsp_sam_fold_expand(const float* tq, uint16_t* out_hi, uint16_t* out_lo, int32_t out_scale_log2, int32_t R, int32_t T,
                                   float scale, rsp_stream_t stream) {
  if (!tq || !out_hi || !out_lo || R < 0 || T <= 0 || 8 * T > 96 || !RSP_PLANE_WORD_VALID(out_scale_log2) ||
      RSP_PLANE_IS_F8(out_scale_log2))
    return RSP_EINVAL;
  if (R == 0) return RSP_OK;
  const int64_t rows = (int64_t)R * 96, n = rows * 32;
  hipLaunchKernelGGL(sam_fold_expand_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, tq,
                     reinterpret_cast<half_t*>(out_hi), reinterpret_cast<half_t*>(out_lo), rows, T, scale,
                     ldexpf(1.0f, RSP_PLANE_EXP(out_scale_log2)));
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

extern "C" int rsp_sam_fold_gather(const float* full, float* ao, int32_t R, int32_t T, rsp_stream_t stream) {
  if (!full || !ao || R < 0 || T <= 0 || 8 * T > 96) return RSP_EINVAL;
  if (R == 0) return RSP_OK;
  const int64_t n4 = (int64_t)R * T * 32;
  hipLaunchKernelGGL(sam_fold_gather_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, full, ao, n4, T);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}
