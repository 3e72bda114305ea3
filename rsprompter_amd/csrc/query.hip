// Kernels of the query prompter (RSMask2FormerHead / MSDeformAttnPixelDecoder / fusion head):
//   GroupNorm on channels-last maps          mmcv ConvModule(norm_cfg=GN) in msdeformattn_pixel_decoder.py:73-111
//   bilinear resize + add (FPN top-down)      msdeformattn_pixel_decoder.py:234-241
//   multi-scale deformable attention sampling mmcv MultiScaleDeformableAttention
//   cross-attention mask from mask_pred_plus  models.py:386-391, 439-442
//   SAM mask embedding -> dense prompt + image embedding add  HF:584-593, models.py:359-362, HF:499
//   class softmax + top-k, mask statistics    maskformer_fusion_head.py:149-176, structures/mask/utils.py:56-77
// All HBM / latency bound; fp32 arithmetic.
#include "rsp_common.h"

namespace {

// ------------------------------------------------------------------------------------ GroupNorm
// stats[b, g] = (sum, sum of squares) in double; x: [B, HW, C], group g = channels [g*cg, (g+1)*cg)
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, double* __restrict__ stats,
                                                      int HW, int C, int G, int rows_per_block) {
  const int b = blockIdx.y;
  const int c4n = C / 4, cg = C / G;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(HW, r0 + rows_per_block);
  // thread owns a fixed float4 channel slot; C/4 must divide 256 (C in {128, 256})
  const int slot = threadIdx.x % c4n, rsub = threadIdx.x / c4n, rstep = 256 / c4n;
  float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
  for (int r = r0 + rsub; r < r1; r += rstep) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((int64_t)b * HW + r) * C + slot * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { s[j] += v[j]; q[j] += v[j] * v[j]; }
  }
  // channels of one float4 slot belong to the same group when cg % 4 == 0, else per channel
  if (cg % 4 == 0) {
    const int g = (slot * 4) / cg;
    atomicAdd(&stats[((int64_t)b * G + g) * 2 + 0], (double)((s[0] + s[1]) + (s[2] + s[3])));
    atomicAdd(&stats[((int64_t)b * G + g) * 2 + 1], (double)((q[0] + q[1]) + (q[2] + q[3])));
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int g = (slot * 4 + j) / cg;
      atomicAdd(&stats[((int64_t)b * G + g) * 2 + 0], (double)s[j]);
      atomicAdd(&stats[((int64_t)b * G + g) * 2 + 1], (double)q[j]);
    }
  }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, const double* __restrict__ stats,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ add, float* __restrict__ y, int64_t total4,
                                                      int HW, int C, int G, float eps, int relu) {
  const int c4n = C / 4, cg = C / G;
  const double n = (double)HW * cg;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    const int64_t row = i / c4n;
    const int b = (int)(row / HW);
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + row * C + c);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int g = (c + j) / cg;
      const double sm = stats[((int64_t)b * G + g) * 2], sq = stats[((int64_t)b * G + g) * 2 + 1];
      const double mean = sm / n;
      const double var = fmax(sq / n - mean * mean, 0.0);
      const float rstd = (float)(1.0 / sqrt(var + (double)eps));
      float t = (v[j] - (float)mean) * rstd * gamma[c + j] + beta[c + j];
      if (relu) t = t > 0.f ? t : 0.f;
      o[j] = t;
    }
    if (add) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(add + row * C + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] += a[j];
    }
    *reinterpret_cast<f32x4*>(y + row * C + c) = o;
  }
}

// ------------------------------------------------------------------------------------ bilinear resize (NHWC)
struct Lin2 { int i0, i1; float l0, l1; };
__device__ __forceinline__ Lin2 lin2(int dst, float scale, int in_size) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  Lin2 c;
  c.i0 = (int)src;
  if (c.i0 > in_size - 1) c.i0 = in_size - 1;
  c.i1 = c.i0 + (c.i0 < in_size - 1 ? 1 : 0);
  c.l1 = src - (float)c.i0;
  c.l0 = 1.0f - c.l1;
  return c;
}

// y[b, oy, ox, :] = F.interpolate(x, (Ho, Wo), bilinear, align_corners=False)[b, :, oy, ox]   (channels-last)
__global__ __launch_bounds__(256) void resize_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H,
                                                         int W, int Ho, int Wo, int C) {
  const int c4n = C / 4;
  const int64_t total = (int64_t)B * Ho * Wo * c4n;
  const float sh = (float)H / (float)Ho, sw = (float)W / (float)Wo;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    int64_t t = i / c4n;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const Lin2 cy = lin2(oy, sh, H), cx = lin2(ox, sw, W);
    const float* base = x + (int64_t)b * H * W * C + c;
    const f32x4 v00 = *reinterpret_cast<const f32x4*>(base + ((int64_t)cy.i0 * W + cx.i0) * C);
    const f32x4 v01 = *reinterpret_cast<const f32x4*>(base + ((int64_t)cy.i0 * W + cx.i1) * C);
    const f32x4 v10 = *reinterpret_cast<const f32x4*>(base + ((int64_t)cy.i1 * W + cx.i0) * C);
    const f32x4 v11 = *reinterpret_cast<const f32x4*>(base + ((int64_t)cy.i1 * W + cx.i1) * C);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      o[j] = cy.l0 * (cx.l0 * v00[j] + cx.l1 * v01[j]) + cy.l1 * (cx.l0 * v10[j] + cx.l1 * v11[j]);
    *reinterpret_cast<f32x4*>(y + i * 4) = o;
  }
}

// ------------------------------------------------------------------------------------ MSDeformAttn sampling
struct MsdaP {
  const float* value;   // [B, Ntok, H*D]
  const float* ow;      // [B, Ntok, ldow]: cols [0, H*L*P*2) offsets (h, l, p, xy), then H*L*P attention logits
  const float* ref;     // [Ntok, 2] (x, y) normalised reference point (same for every level: valid_ratios == 1)
  float* out;           // [B, Ntok, H*D]
  int B, Ntok, ldow;
  int lvl_h[5], lvl_w[5], lvl_start[5];
};
constexpr int MS_H = 8, MS_P = 4, MS_LMAX = 5;

// one wave per (b, token): lane = head * 8 + slot, every lane owns MS_D / 8 consecutive channels of its head
// (MS_D = 16: embed 128, the RSPrompter query prompter; MS_D = 32: embed 256, the standard Mask2Former pixel decoder)
// MS_L = self_attn_cfg.num_levels = num_transformer_feat_level (mask2former_head.py:106-107): 3 in every shipped config
template <int MS_D, int MS_L>
__global__ __launch_bounds__(256) void msda_kernel(const MsdaP p) {
  constexpr int DPL = MS_D / 8;
  const int lane = threadIdx.x & 63;
  const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= (int64_t)p.B * p.Ntok) return;
  const int b = (int)(tok / p.Ntok), q = (int)(tok - (int64_t)b * p.Ntok);
  const int h = lane >> 3, dp = (lane & 7) * DPL;
  const float* owr = p.ow + tok * p.ldow;
  const float* logit = owr + MS_H * MS_L * MS_P * 2 + h * (MS_L * MS_P);
  float w[MS_L * MS_P];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < MS_L * MS_P; ++i) { w[i] = logit[i]; m = fmaxf(m, w[i]); }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MS_L * MS_P; ++i) { w[i] = expf(w[i] - m); sum += w[i]; }
  const float rx = p.ref[q * 2], ry = p.ref[q * 2 + 1];
  const float* vb = p.value + (int64_t)b * p.Ntok * (MS_H * MS_D) + h * MS_D + dp;
  float acc[DPL];
#pragma unroll
  for (int e = 0; e < DPL; ++e) acc[e] = 0.f;
#pragma unroll
  for (int l = 0; l < MS_L; ++l) {
    const int Hl = p.lvl_h[l], Wl = p.lvl_w[l];
    const float* vl = vb + (int64_t)p.lvl_start[l] * (MS_H * MS_D);
#pragma unroll
    for (int pt = 0; pt < MS_P; ++pt) {
      const float* o2 = owr + ((h * MS_L + l) * MS_P + pt) * 2;
      const float locx = rx + o2[0] / (float)Wl, locy = ry + o2[1] / (float)Hl;
      // grid_sample(align_corners=False): pixel = ((2*loc - 1) + 1) * size / 2 - 0.5
      const float gx = 2.f * locx - 1.f, gy = 2.f * locy - 1.f;
      const float x = ((gx + 1.f) * (float)Wl - 1.f) * 0.5f, y = ((gy + 1.f) * (float)Hl - 1.f) * 0.5f;
      const float xf = floorf(x), yf = floorf(y);
      const int x0 = (int)xf, y0 = (int)yf;
      const float lx = x - xf, ly = y - yf;
      const float wt = w[l * MS_P + pt] / sum;
      float sv[DPL];
#pragma unroll
      for (int e = 0; e < DPL; ++e) sv[e] = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int xx = x0 + (t & 1), yy = y0 + (t >> 1);
        if (xx >= 0 && xx < Wl && yy >= 0 && yy < Hl) {
          const float bw = ((t & 1) ? lx : 1.f - lx) * ((t >> 1) ? ly : 1.f - ly);
          const float* vp = vl + (int64_t)(yy * Wl + xx) * (MS_H * MS_D);
#pragma unroll
          for (int e = 0; e < DPL; ++e) sv[e] += bw * vp[e];
        }
      }
#pragma unroll
      for (int e = 0; e < DPL; ++e) acc[e] += wt * sv[e];
    }
  }
  float* o = p.out + tok * (MS_H * MS_D) + h * MS_D + dp;
#pragma unroll
  for (int e = 0; e < DPL; ++e) o[e] = acc[e];
}

// ------------------------------------------------------------------------------------ cross-attention mask
// mask[b, q, k] = sigmoid(bilinear(mask_pred_plus[b, q], (h, w))[k]) < 0.5 ; rows that would be fully blocked
// are cleared (models.py:386-391, 439-442).  One block per (b, q) row.
__global__ __launch_bounds__(256) void attn_mask_kernel(const float* __restrict__ mpp, uint8_t* __restrict__ mask, int Hs,
                                                       int Ws, int h, int w) {
  __shared__ int s_cnt;
  const int64_t row = blockIdx.x;
  const float* src = mpp + row * Hs * Ws;
  uint8_t* dst = mask + row * h * w;
  const float sh = (float)Hs / (float)h, sw = (float)Ws / (float)w;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  int cnt = 0;
  for (int k = threadIdx.x; k < h * w; k += blockDim.x) {
    const int oy = k / w, ox = k - oy * w;
    const Lin2 cy = lin2(oy, sh, Hs), cx = lin2(ox, sw, Ws);
    const float v = cy.l0 * (cx.l0 * src[cy.i0 * Ws + cx.i0] + cx.l1 * src[cy.i0 * Ws + cx.i1]) +
                    cy.l1 * (cx.l0 * src[cy.i1 * Ws + cx.i0] + cx.l1 * src[cy.i1 * Ws + cx.i1]);
    const uint8_t blocked = (1.0f / (1.0f + expf(-v))) < 0.5f ? 1 : 0;
    dst[k] = blocked;
    cnt += blocked;
  }
  atomicAdd(&s_cnt, cnt);
  __syncthreads();
  if (s_cnt == h * w)
    for (int k = threadIdx.x; k < h * w; k += blockDim.x) dst[k] = 0;
}

// ------------------------------------------------------------------------------------ SAM mask embedding
struct MaskEmbP {
  const float* mpp;        // [R, Hs, Ws] (Hs = 4*he)
  const float* emb;        // [B, he*we, C] image embeddings (channels-last)
  const int32_t* roi_img;  // [R]
  const float *w1, *b1, *g1, *be1;   // conv1 [4,1,2,2], bias [4], LN [4]
  const float *w2, *b2, *g2, *be2;   // conv2 [16,4,2,2], bias [16], LN [16]
  const float *w3, *b3;              // conv3 [C,16], bias [C]
  float* out;              // [R, he*we, C] = emb[img] + dense prompt
  int R, he, we, C;
  float eps;
};

// one wave per 64 consecutive output pixels of one RoI: each lane runs conv1/LN/GELU/conv2/LN/GELU for its own
// pixel, then the wave walks the 64 pixels and every lane produces 4 of the C=256 output channels
__global__ __launch_bounds__(256) void mask_embed_kernel(const MaskEmbP p) {
  const int lane = threadIdx.x & 63;
  const int npix = p.he * p.we;
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int waves_per_roi = (npix + 63) / 64;
  const int r = (int)(wave_id / waves_per_roi);
  if (r >= p.R) return;
  const int pix0 = (int)(wave_id - (int64_t)r * waves_per_roi) * 64;
  const int pix = pix0 + lane;
  const int Ws = p.we * 4;
  float h2[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) h2[i] = 0.f;
  if (pix < npix) {
    const int oy = pix / p.we, ox = pix - oy * p.we;
    const float* src = p.mpp + (int64_t)r * (p.he * 4) * Ws + (int64_t)(oy * 4) * Ws + ox * 4;
    float acc2[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc2[i] = p.b2[i];
#pragma unroll
    for (int sy = 0; sy < 2; ++sy)
#pragma unroll
      for (int sx = 0; sx < 2; ++sx) {
        // conv1 (k2 s2) at the (sy, sx) position of the 2x2 block feeding this output pixel
        const float i00 = src[(sy * 2) * Ws + sx * 2], i01 = src[(sy * 2) * Ws + sx * 2 + 1];
        const float i10 = src[(sy * 2 + 1) * Ws + sx * 2], i11 = src[(sy * 2 + 1) * Ws + sx * 2 + 1];
        float h1[4];
        float mean = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          h1[c] = p.b1[c] + p.w1[c * 4 + 0] * i00 + p.w1[c * 4 + 1] * i01 + p.w1[c * 4 + 2] * i10 + p.w1[c * 4 + 3] * i11;
          mean += h1[c];
        }
        mean *= 0.25f;
        float var = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) { const float d = h1[c] - mean; var += d * d; }
        const float rstd = 1.0f / sqrtf(var * 0.25f + p.eps);
#pragma unroll
        for (int c = 0; c < 4; ++c) h1[c] = rsp_gelu((h1[c] - mean) * rstd * p.g1[c] + p.be1[c]);
        // conv2 (k2 s2): weight [16, 4, 2, 2]
#pragma unroll
        for (int o = 0; o < 16; ++o)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc2[o] += p.w2[((o * 4 + c) * 2 + sy) * 2 + sx] * h1[c];
      }
    float mean = 0.f;
#pragma unroll
    for (int o = 0; o < 16; ++o) mean += acc2[o];
    mean *= (1.0f / 16.f);
    float var = 0.f;
#pragma unroll
    for (int o = 0; o < 16; ++o) { const float d = acc2[o] - mean; var += d * d; }
    const float rstd = 1.0f / sqrtf(var * (1.0f / 16.f) + p.eps);
#pragma unroll
    for (int o = 0; o < 16; ++o) h2[o] = rsp_gelu((acc2[o] - mean) * rstd * p.g2[o] + p.be2[o]);
  }
  // conv3 (1x1, 16 -> C) + bias + image embedding; lane owns channels [4*lane, 4*lane+4) (+256 per round)
  const int img = p.roi_img[r];
  for (int c0 = lane * 4; c0 < p.C; c0 += 256) {
    float w3[4][16];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) w3[j][i] = p.w3[(c0 + j) * 16 + i];
    const f32x4 bias = *reinterpret_cast<const f32x4*>(p.b3 + c0);
    const int nvalid = min(64, npix - pix0);
    for (int j = 0; j < nvalid; ++j) {
      f32x4 o = bias;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float hv = __shfl(h2[i], j, 64);
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] += w3[c][i] * hv;
      }
      const f32x4 e = *reinterpret_cast<const f32x4*>(p.emb + ((int64_t)img * npix + pix0 + j) * p.C + c0);
#pragma unroll
      for (int c = 0; c < 4; ++c) o[c] += e[c];
      *reinterpret_cast<f32x4*>(p.out + ((int64_t)r * npix + pix0 + j) * p.C + c0) = o;
    }
  }
}

// ------------------------------------------------------------------------------------ fusion head
__device__ __forceinline__ uint32_t f2ord_(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// per image: softmax over nc+1 logits, drop the background column, top-k of the Nq*nc scores with
// (score desc, flat index asc) ties (maskformer_fusion_head.py:149-162)
__global__ __launch_bounds__(1024) void query_topk_kernel(const float* __restrict__ cls, int Nq, int nc, int k, int nsort,
                                                         float* __restrict__ out_score, int32_t* __restrict__ out_flat) {
  extern __shared__ unsigned long long keys[];
  const int b = blockIdx.x;
  const int n = Nq * nc;
  for (int i = threadIdx.x; i < nsort; i += blockDim.x) {
    unsigned long long key = 0ull;
    if (i < n) {
      const int q = i / nc, c = i - q * nc;
      const float* row = cls + ((int64_t)b * Nq + q) * (nc + 1);
      float m = row[0];
      for (int j = 1; j <= nc; ++j) m = fmaxf(m, row[j]);
      float s = 0.f;
      for (int j = 0; j <= nc; ++j) s += expf(row[j] - m);
      const float sc = expf(row[c] - m) / s;
      key = ((unsigned long long)f2ord_(sc) << 32) | (uint32_t)(0xffffffffu - (uint32_t)i);
    }
    keys[i] = key;
  }
  __syncthreads();
  for (int kk = 2; kk <= nsort; kk <<= 1)
    for (int j = kk >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < nsort; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = keys[i], c = keys[ixj];
          const bool desc = ((i & kk) == 0);
          if (desc ? (a < c) : (a > c)) { keys[i] = c; keys[ixj] = a; }
        }
      }
      __syncthreads();
    }
  for (int i = threadIdx.x; i < k; i += blockDim.x) {
    const unsigned long long key = keys[i];
    const uint32_t o = (uint32_t)(key >> 32);
    const uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    out_score[(int64_t)b * k + i] = __uint_as_float(u);
    out_flat[(int64_t)b * k + i] = (int32_t)(0xffffffffu - (uint32_t)(key & 0xffffffffull));
  }
}

struct QMaskP {
  const float* low;        // [*, h, w] SAM logits per query (of this image)
  const int32_t* qidx;     // [k] query index of every selected instance
  uint8_t* out;            // [k, oh, ow] bool
  double* stats;           // [k, 2] (sum sigmoid over positive pixels, positive count)
  int32_t* box;            // [k, 4] xmin, ymin, xmax, ymax (init: INT_MAX, INT_MAX, -1, -1)
  float* logits;           // optional [k, oh, ow]
  int h, w, Hb, Wb, ch, cw, oh, ow;
};

// models.py:652-656 + :684-695 + maskformer_fusion_head.py:164-176: logits are interpolated (twice when rescale)
// BEFORE the > 0 test; per-instance sums / extents via block reduction + atomics
template <bool IDENT>
__global__ __launch_bounds__(256) void query_mask_kernel(const QMaskP p) {
  __shared__ double s_sum[4], s_cnt[4];
  __shared__ int s_box[4][4];
  const int m = blockIdx.y;
  const float* low = p.low + (int64_t)p.qidx[m] * p.h * p.w;
  const float s1h = (float)p.h / (float)p.Hb, s1w = (float)p.w / (float)p.Wb;
  const float s2h = (float)p.ch / (float)p.oh, s2w = (float)p.cw / (float)p.ow;
  const int64_t total = (int64_t)p.oh * p.ow;
  auto stage1 = [&](int Y, int X) -> float {
    const Lin2 ay = lin2(Y, s1h, p.h), ax = lin2(X, s1w, p.w);
    return ay.l0 * (ax.l0 * low[ay.i0 * p.w + ax.i0] + ax.l1 * low[ay.i0 * p.w + ax.i1]) +
           ay.l1 * (ax.l0 * low[ay.i1 * p.w + ax.i0] + ax.l1 * low[ay.i1 * p.w + ax.i1]);
  };
  double sum = 0.0, cnt = 0.0;
  int xmin = 0x7fffffff, ymin = 0x7fffffff, xmax = -1, ymax = -1;
  auto pixel = [&](int oy, int ox) -> float {
    if (IDENT) return stage1(oy, ox);
    const Lin2 cy = lin2(oy, s2h, p.ch), cx = lin2(ox, s2w, p.cw);
    return cy.l0 * (cx.l0 * stage1(cy.i0, cx.i0) + cx.l1 * stage1(cy.i0, cx.i1)) +
           cy.l1 * (cx.l0 * stage1(cy.i1, cx.i0) + cx.l1 * stage1(cy.i1, cx.i1));
  };
  float fsum = 0.f;   // per-thread partial (a thread sees <= a few dozen pixels); widened to double once at the end
  int icnt = 0;
  auto account = [&](float v, int oy, int ox) {
    if (v > 0.f) {
      // sigmoid through v_exp_f32 / v_rcp_f32 (1-2 ulp): it only enters the mask-average score
      fsum += __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f));
      icnt += 1;
      xmin = min(xmin, ox); xmax = max(xmax, ox); ymin = min(ymin, oy); ymax = max(ymax, oy);
    }
  };
  if ((p.ow & 3) == 0) {
    // four pixels of one row per thread: one 32-bit store of the bool mask instead of four byte stores
    const int qw = p.ow >> 2;
    const int nq = p.oh * qw;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += gridDim.x * blockDim.x) {
      const int oy = i / qw, ox = (i - oy * qw) << 2;
      float v[4];
      if (IDENT) {          // the row coefficients are shared by the four pixels
        const Lin2 ay = lin2(oy, s1h, p.h);
        const float* r0 = low + ay.i0 * p.w;
        const float* r1 = low + ay.i1 * p.w;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const Lin2 ax = lin2(ox + e, s1w, p.w);
          v[e] = ay.l0 * (ax.l0 * r0[ax.i0] + ax.l1 * r0[ax.i1]) + ay.l1 * (ax.l0 * r1[ax.i0] + ax.l1 * r1[ax.i1]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = pixel(oy, ox + e);
      }
      uint32_t bits = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) { bits |= (v[e] > 0.f ? 1u : 0u) << (8 * e); account(v[e], oy, ox + e); }
      const int64_t o = (int64_t)m * total + (int64_t)oy * p.ow + ox;
      *reinterpret_cast<uint32_t*>(p.out + o) = bits;
      if (p.logits) *reinterpret_cast<f32x4*>(p.logits + o) = f32x4{v[0], v[1], v[2], v[3]};
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
      const int oy = (int)(i / p.ow), ox = (int)(i - (int64_t)oy * p.ow);
      const float v = pixel(oy, ox);
      p.out[(int64_t)m * total + i] = v > 0.f ? 1 : 0;
      if (p.logits) p.logits[(int64_t)m * total + i] = v;
      account(v, oy, ox);
    }
  }
  sum = (double)fsum; cnt = (double)icnt;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sum += __shfl_xor(sum, o, 64); cnt += __shfl_xor(cnt, o, 64);
    xmin = min(xmin, __shfl_xor(xmin, o, 64)); ymin = min(ymin, __shfl_xor(ymin, o, 64));
    xmax = max(xmax, __shfl_xor(xmax, o, 64)); ymax = max(ymax, __shfl_xor(ymax, o, 64));
  }
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_sum[wv] = sum; s_cnt[wv] = cnt; s_box[wv][0] = xmin; s_box[wv][1] = ymin; s_box[wv][2] = xmax; s_box[wv][3] = ymax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 4; ++i) {
      sum += s_sum[i]; cnt += s_cnt[i];
      xmin = min(xmin, s_box[i][0]); ymin = min(ymin, s_box[i][1]); xmax = max(xmax, s_box[i][2]); ymax = max(ymax, s_box[i][3]);
    }
    if (cnt > 0.0) {
      atomicAdd(&p.stats[m * 2], sum);
      atomicAdd(&p.stats[m * 2 + 1], cnt);
      atomicMin(&p.box[m * 4 + 0], xmin); atomicMin(&p.box[m * 4 + 1], ymin);
      atomicMax(&p.box[m * 4 + 2], xmax); atomicMax(&p.box[m * 4 + 3], ymax);
    }
  }
}

// det_score = cls_score * (sum / (cnt + 1e-6)); bbox = [xmin, ymin, xmax+1, ymax+1] or zeros (mask2bbox)
__global__ void query_finalize_kernel(const float* __restrict__ cls_score, const double* __restrict__ stats,
                                      const int32_t* __restrict__ box, float* __restrict__ det_score,
                                      float* __restrict__ bboxes, int k) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  const float ms = (float)stats[i * 2] / ((float)stats[i * 2 + 1] + 1e-6f);
  det_score[i] = cls_score[i] * ms;
  const bool any = box[i * 4 + 2] >= 0;
  bboxes[i * 4 + 0] = any ? (float)box[i * 4 + 0] : 0.f;
  bboxes[i * 4 + 1] = any ? (float)box[i * 4 + 1] : 0.f;
  bboxes[i * 4 + 2] = any ? (float)(box[i * 4 + 2] + 1) : 0.f;
  bboxes[i * 4 + 3] = any ? (float)(box[i * 4 + 3] + 1) : 0.f;
}

__global__ void init_qstats_kernel(double* stats, int32_t* box, int k) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  stats[i * 2] = 0.0; stats[i * 2 + 1] = 0.0;
  box[i * 4 + 0] = 0x7fffffff; box[i * 4 + 1] = 0x7fffffff; box[i * 4 + 2] = -1; box[i * 4 + 3] = -1;
}

inline int grid_for64(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int rsp_groupnorm_nhwc(const float* x, const float* gamma, const float* beta, const float* add, float* y,
                                  double* stats_ws, int32_t B, int32_t HW, int32_t C, int32_t G, float eps,
                                  int32_t relu, rsp_stream_t stream) {
  if (!x || !gamma || !beta || !y || !stats_ws || B <= 0 || HW <= 0 || C <= 0 || (C & 3) || G <= 0 || C % G ||
      256 % (C / 4))
    return RSP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(stats_ws, 0, sizeof(double) * 2 * B * G, s) != hipSuccess) return RSP_ELAUNCH;
  const int rows_per_block = 256;
  hipLaunchKernelGGL(gn_stats_kernel, dim3((HW + rows_per_block - 1) / rows_per_block, B), dim3(256), 0, s, x, stats_ws,
                     HW, C, G, rows_per_block);
  const int64_t total4 = (int64_t)B * HW * (C / 4);
  hipLaunchKernelGGL(gn_apply_kernel, dim3(grid_for64(total4)), dim3(256), 0, s, x, stats_ws, gamma, beta, add, y, total4,
                     HW, C, G, eps, relu);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

extern "C" int rsp_resize_bilinear_nhwc(const float* x, float* y, int32_t B, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                                        int32_t C, rsp_stream_t stream) {
  if (!x || !y || B <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || (C & 3)) return RSP_EINVAL;
  const int64_t total = (int64_t)B * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(resize_nhwc_kernel, dim3(grid_for64(total)), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W, Ho, Wo, C);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

extern "C" int rsp_msdeform_attn_ex(const float* value, const float* offs_weights, int32_t ld_ow, const float* ref_points,
                                    float* out, int32_t B, int32_t Ntok, int32_t num_levels, const int32_t* level_hw /*host [L,2]*/,
                                    int32_t head_dim, rsp_stream_t stream) {
  if (!value || !offs_weights || !ref_points || !out || !level_hw || B <= 0 || Ntok <= 0 || num_levels < 1 ||
      num_levels > MS_LMAX || ld_ow < MS_H * num_levels * MS_P * 3 || !(head_dim == 16 || head_dim == 32))
    return RSP_EINVAL;
  MsdaP p;
  p.value = value; p.ow = offs_weights; p.ref = ref_points; p.out = out; p.B = B; p.Ntok = Ntok; p.ldow = ld_ow;
  int start = 0;
  for (int l = 0; l < MS_LMAX; ++l) {
    if (l < num_levels) {
      p.lvl_h[l] = level_hw[2 * l]; p.lvl_w[l] = level_hw[2 * l + 1]; p.lvl_start[l] = start;
      start += p.lvl_h[l] * p.lvl_w[l];
    } else { p.lvl_h[l] = p.lvl_w[l] = 1; p.lvl_start[l] = 0; }
  }
  if (start != Ntok) return RSP_EINVAL;
  const int64_t waves = (int64_t)B * Ntok;
  const dim3 grid((unsigned)((waves + 3) / 4));
#define MSDA_CASE(L)                                                                                                   \
  case L:                                                                                                              \
    if (head_dim == 16) hipLaunchKernelGGL((msda_kernel<16, L>), grid, dim3(256), 0, (hipStream_t)stream, p);          \
    else hipLaunchKernelGGL((msda_kernel<32, L>), grid, dim3(256), 0, (hipStream_t)stream, p);                         \
    break;
  switch (num_levels) {
    MSDA_CASE(1) MSDA_CASE(2) MSDA_CASE(3) MSDA_CASE(4) MSDA_CASE(5)
  }
#undef MSDA_CASE
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

extern "C" int rsp_msdeform_attn(const float* value, const float* offs_weights, int32_t ld_ow, const float* ref_points,
                                 float* out, int32_t B, int32_t Ntok, int32_t num_levels, const int32_t* level_hw,
                                 rsp_stream_t stream) {
  return rsp_msdeform_attn_ex(value, offs_weights, ld_ow, ref_points, out, B, Ntok, num_levels, level_hw, 16, stream);
}

extern "C" int rsp_query_attn_mask(const float* mask_pred_plus, uint8_t* mask, int64_t rows, int32_t Hs, int32_t Ws, int32_t h,
                                   int32_t w, rsp_stream_t stream) {
  if (!mask_pred_plus || !mask || rows <= 0 || Hs <= 0 || Ws <= 0 || h <= 0 || w <= 0) return RSP_EINVAL;
  hipLaunchKernelGGL(attn_mask_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, mask_pred_plus, mask, Hs, Ws, h, w);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

extern "C" int rsp_sam_mask_embed(const RspMaskEmbedDesc* d, rsp_stream_t stream) {
  if (!d || !d->mask_pred_plus || !d->image_embeddings || !d->roi_img || !d->out || d->R <= 0 || d->he <= 0 || d->we <= 0 ||
      d->C <= 0 || (d->C & 255))      // a round of the output loop = 64 lanes x 4 channels, and its pixel broadcast is a wave
    return RSP_EINVAL;                // shuffle: every lane has to take part in every round (SAM: C = 256)
  MaskEmbP p;
  p.mpp = d->mask_pred_plus; p.emb = d->image_embeddings; p.roi_img = d->roi_img;
  p.w1 = d->conv1_w; p.b1 = d->conv1_b; p.g1 = d->ln1_w; p.be1 = d->ln1_b;
  p.w2 = d->conv2_w; p.b2 = d->conv2_b; p.g2 = d->ln2_w; p.be2 = d->ln2_b;
  p.w3 = d->conv3_w; p.b3 = d->conv3_b;
  p.out = d->out; p.R = d->R; p.he = d->he; p.we = d->we; p.C = d->C; p.eps = d->eps;
  const int64_t waves = (int64_t)d->R * ((d->he * d->we + 63) / 64);
  hipLaunchKernelGGL(mask_embed_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

extern "C" int rsp_query_topk(const float* cls, int32_t B, int32_t Nq, int32_t nc, int32_t k, float* out_score,
                              int32_t* out_flat, rsp_stream_t stream) {
  if (!cls || !out_score || !out_flat || B <= 0 || Nq <= 0 || nc <= 0 || k <= 0 || k > Nq * nc || Nq * nc > 16384) return RSP_EINVAL;
  int nsort = 1;
  while (nsort < Nq * nc) nsort <<= 1;
  const size_t smem = (size_t)nsort * sizeof(unsigned long long);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&query_topk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)smem) != hipSuccess)
    return RSP_ELAUNCH;
  hipLaunchKernelGGL(query_topk_kernel, dim3(B), dim3(1024), smem, (hipStream_t)stream, cls, Nq, nc, k, nsort, out_score, out_flat);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

extern "C" int rsp_query_mask_post(const float* low_res, const int32_t* qidx, const float* cls_score, int32_t k, int32_t h,
                                   int32_t w, int32_t Hb, int32_t Wb, int32_t crop_h, int32_t crop_w, int32_t out_h,
                                   int32_t out_w, void* stats_ws, uint8_t* out_mask, float* out_logits, float* det_score,
                                   float* bboxes, rsp_stream_t stream) {
  if (!low_res || !qidx || !cls_score || !stats_ws || !out_mask || !det_score || !bboxes || k < 0 || h <= 0 || w <= 0 ||
      crop_h <= 0 || crop_w <= 0 || crop_h > Hb || crop_w > Wb || out_h <= 0 || out_w <= 0)
    return RSP_EINVAL;
  if (k == 0) return RSP_OK;
  hipStream_t s = (hipStream_t)stream;
  QMaskP p;
  p.low = low_res; p.qidx = qidx; p.out = out_mask; p.logits = out_logits;
  p.stats = (double*)stats_ws; p.box = (int32_t*)((char*)stats_ws + sizeof(double) * 2 * k);
  p.h = h; p.w = w; p.Hb = Hb; p.Wb = Wb; p.ch = crop_h; p.cw = crop_w; p.oh = out_h; p.ow = out_w;
  hipLaunchKernelGGL(init_qstats_kernel, dim3((k + 63) / 64), dim3(64), 0, s, p.stats, p.box, k);
  if ((int64_t)out_h * out_w > 0x7fffffffLL) return RSP_EINVAL;
  int64_t gx = ((int64_t)out_h * out_w / ((out_w & 3) == 0 ? 4 : 1) + 255) / 256;
  // every block ends with 6 atomics on its mask's statistics: with 1024 blocks per mask those same-address atomics
  // (0.6 M per call) were the whole run time; keep the grid just large enough to fill the chip
  const int64_t cap = k >= 64 ? 64 : (k >= 8 ? 256 : 1024);
  if (gx > cap) gx = cap;
  if (crop_h == out_h && crop_w == out_w)
    hipLaunchKernelGGL((query_mask_kernel<true>), dim3((unsigned)gx, k), dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL((query_mask_kernel<false>), dim3((unsigned)gx, k), dim3(256), 0, s, p);
  hipLaunchKernelGGL(query_finalize_kernel, dim3((k + 63) / 64), dim3(64), 0, s, cls_score, p.stats, p.box, det_score, bboxes, k);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}
