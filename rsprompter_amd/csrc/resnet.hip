// ResNet-50 + FPN pieces of the SAMDet detector (configs/rsprompter/_base_/samdet.py:56-75) that are not
// GEMMs, and the box prompt of the SAM prompt encoder.  Everything else of the detector runs on rsp_gemm: 1x1 convs as
// GEMMs, 3x3 / strided 1x1 convs as implicit GEMMs, eval-mode BatchNorm folded into the weights, ReLU in the epilogue
// (RSP_ACT_RELU_POST for the ReLU after the shortcut, mmdet/models/backbones/resnet.py:283-286).
//   stem        resnet.py:640-647  conv1 7x7 s2 p3 (3 -> 64) + bn1 + relu, NCHW fp32 in, NHWC out
//   maxpool     resnet.py:598      MaxPool2d(3, stride 2, padding 1), NHWC
//   top-down    necks/fpn.py:190-204  laterals[i-1] += F.interpolate(laterals[i], size=prev_shape, mode='nearest')
//   box prompt  HF SamPromptEncoder._embed_boxes (transformers 4.38.1 modeling_sam.py:647-656) +
//               SamPositionalEmbedding.forward (:552-566), reached from SAMDet.predict (models.py:1174-1178)
#include "rsp_common.h"

namespace {

constexpr int ST_TY = 8, ST_TX = 32;                 // output tile of one block: 8 rows x 32 columns, one pixel per thread
constexpr int ST_PH = ST_TY * 2 + 5, ST_PW = ST_TX * 2 + 5, ST_PWP = ST_PW + 2;   // input patch (+ pad against bank aliasing)
constexpr int ST_K = 147;                            // 7 * 7 * 3 taps

// w: [147][64] (tap = (c * 7 + ky) * 7 + kx, BN folded), bias [64]; x: [B, 3, H, W]; y: [B, Ho, Wo, 64]
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y, int H,
                                                        int W, int Ho, int Wo) {
  extern __shared__ float sm[];
  float* sw = sm;                                    // [147][64]
  float* sp = sm + ST_K * 64;                        // [3][ST_PH][ST_PWP]
  const int tid = threadIdx.x;
  const int b = blockIdx.z;
  const int oy0 = blockIdx.y * ST_TY, ox0 = blockIdx.x * ST_TX;
  for (int i = tid; i < ST_K * 64 / 4; i += 256)
    reinterpret_cast<f32x4*>(sw)[i] = reinterpret_cast<const f32x4*>(w)[i];
  const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
  for (int i = tid; i < 3 * ST_PH * ST_PW; i += 256) {
    const int c = i / (ST_PH * ST_PW);
    const int r = i - c * (ST_PH * ST_PW);
    const int py = r / ST_PW, px = r - py * ST_PW;
    const int iy = iy0 + py, ix = ix0 + px;
    float v = 0.f;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[(((int64_t)b * 3 + c) * H + iy) * W + ix];
    sp[(c * ST_PH + py) * ST_PWP + px] = v;
  }
  __syncthreads();
  const int ty = tid >> 5, tx = tid & 31;
  const int oy = oy0 + ty, ox = ox0 + tx;
  float acc[64];
#pragma unroll
  for (int o = 0; o < 64; ++o) acc[o] = 0.f;
  for (int c = 0; c < 3; ++c) {
    for (int ky = 0; ky < 7; ++ky) {
      const float* prow = sp + (c * ST_PH + ty * 2 + ky) * ST_PWP + tx * 2;
      const float* wrow = sw + ((c * 7 + ky) * 7) * 64;
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) {
        const float v = prow[kx];
#pragma unroll
        for (int o4 = 0; o4 < 16; ++o4) {
          const f32x4 w4 = *reinterpret_cast<const f32x4*>(wrow + kx * 64 + o4 * 4);
          acc[o4 * 4 + 0] = fmaf(v, w4[0], acc[o4 * 4 + 0]);
          acc[o4 * 4 + 1] = fmaf(v, w4[1], acc[o4 * 4 + 1]);
          acc[o4 * 4 + 2] = fmaf(v, w4[2], acc[o4 * 4 + 2]);
          acc[o4 * 4 + 3] = fmaf(v, w4[3], acc[o4 * 4 + 3]);
        }
      }
    }
  }
  if (oy < Ho && ox < Wo) {
    float* yp = y + (((int64_t)b * Ho + oy) * Wo + ox) * 64;
#pragma unroll
    for (int o4 = 0; o4 < 16; ++o4) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + o4 * 4);
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float t = acc[o4 * 4 + e] + b4[e]; v[e] = t > 0.f ? t : 0.f; }
      *reinterpret_cast<f32x4*>(yp + o4 * 4) = v;
    }
  }
}

// NHWC max pooling, window k, stride s, padding p (-inf outside, like torch); one thread per (pixel, 4 channels)
__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H,
                                                      int W, int C, int Ho, int Wo, int k, int s, int p) {
  const int c4n = C >> 2;
  const int64_t total = (int64_t)B * Ho * Wo * c4n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    int64_t r = i / c4n;
    const int xo = (int)(r % Wo); r /= Wo;
    const int yo = (int)(r % Ho);
    const int b = (int)(r / Ho);
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int dy = 0; dy < k; ++dy) {
      const int yy = yo * s - p + dy;
      if (yy < 0 || yy >= H) continue;
      for (int dx = 0; dx < k; ++dx) {
        const int xx = xo * s - p + dx;
        if (xx < 0 || xx >= W) continue;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((int64_t)b * H + yy) * W + xx) * C + c4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
      }
    }
    *reinterpret_cast<f32x4*>(y + i * 4) = m;
  }
}

// dst[b, y, x, :] += src[b, floor(y * h / H), floor(x * w / W), :]   (torch 'nearest': src = min(floor(dst * scale), in - 1)
// with scale = in / out in fp32, aten/native/UpSample.h nearest_neighbor_compute_source_index)
__global__ __launch_bounds__(256) void upsample_add_kernel(const float* __restrict__ src, float* __restrict__ dst, int B,
                                                           int h, int w, int H, int W, int C) {
  const int c4n = C >> 2;
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  const int64_t total = (int64_t)B * H * W * c4n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    int64_t r = i / c4n;
    const int xo = (int)(r % W); r /= W;
    const int yo = (int)(r % H);
    const int b = (int)(r / H);
    const int ys = min((int)floorf((float)yo * sh), h - 1), xs = min((int)floorf((float)xo * sw), w - 1);
    const f32x4 s = *reinterpret_cast<const f32x4*>(src + (((int64_t)b * h + ys) * w + xs) * C + c4 * 4);
    f32x4 d = *reinterpret_cast<const f32x4*>(dst + i * 4);
    d[0] += s[0]; d[1] += s[1]; d[2] += s[2]; d[3] += s[3];
    *reinterpret_cast<f32x4*>(dst + i * 4) = d;
  }
}

// out[n, corner, 0:F] = sin(2 pi (cx g[0, f] + cy g[1, f])) + pe_corner[f], out[n, corner, F:2F] = cos(..) + pe_corner[F + f]
// with (cx, cy) = 2 ((box corner + 0.5) / size) - 1
__global__ __launch_bounds__(256) void embed_boxes_kernel(const float* __restrict__ boxes, const float* __restrict__ g,
                                                          const float* __restrict__ pe_tl, const float* __restrict__ pe_br,
                                                          float* __restrict__ out, int n, int F, float size_w, float size_h) {
  const int64_t total = (int64_t)n * 2 * F;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(i % F);
    const int corner = (int)((i / F) & 1);
    const int64_t bi = i / (2 * F);
    const float x = (boxes[bi * 4 + corner * 2 + 0] + 0.5f) / size_w;
    const float y = (boxes[bi * 4 + corner * 2 + 1] + 0.5f) / size_h;
    const float cx = 2.0f * x - 1.0f, cy = 2.0f * y - 1.0f;
    float v = cx * g[f] + cy * g[F + f];
    v = 6.283185307179586f * v;
    const float* pe = corner ? pe_br : pe_tl;
    float* o = out + (bi * 2 + corner) * 2 * F;
    o[f] = sinf(v) + pe[f];
    o[F + f] = cosf(v) + pe[F + f];
  }
}

inline unsigned grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 65535 ? 65535 : g));
}

}  // namespace

extern "C" int rsp_resnet_stem(const float* x, const float* w, const float* bias, float* y, int32_t B, int32_t H, int32_t W,
                               rsp_stream_t stream) {
  if (!x || !w || !bias || !y || B < 0 || H < 1 || W < 1) return RSP_EINVAL;
  if (B == 0) return RSP_OK;
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  const size_t smem = (size_t)(ST_K * 64 + 3 * ST_PH * ST_PWP) * sizeof(float);
  // (a per-device attribute: set on every call, no cached "done" flag that would be wrong for a second device)
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(stem_conv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)smem) != hipSuccess)
    return RSP_ELAUNCH;
  dim3 grid((Wo + ST_TX - 1) / ST_TX, (Ho + ST_TY - 1) / ST_TY, B);
  if (grid.y > 65535u || grid.z > 65535u) return RSP_EINVAL;
  hipLaunchKernelGGL(stem_conv_kernel, grid, dim3(256), smem, (hipStream_t)stream, x, w, bias, y, H, W, Ho, Wo);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

extern "C" int rsp_maxpool_nhwc(const float* x, float* y, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k, int32_t s,
                                int32_t p, rsp_stream_t stream) {
  if (!x || !y || B < 0 || H < 1 || W < 1 || C < 4 || (C & 3) || k < 1 || s < 1 || p < 0 || 2 * p > k) return RSP_EINVAL;
  const int Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
  if (Ho < 1 || Wo < 1) return RSP_EINVAL;
  if (B == 0) return RSP_OK;
  const int64_t total = (int64_t)B * Ho * Wo * (C >> 2);
  hipLaunchKernelGGL(maxpool_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W, C, Ho, Wo, k,
                     s, p);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

extern "C" int rsp_upsample_nearest_add(const float* src, float* dst, int32_t B, int32_t h, int32_t w, int32_t H, int32_t W,
                                        int32_t C, rsp_stream_t stream) {
  if (!src || !dst || B < 0 || h < 1 || w < 1 || H < 1 || W < 1 || C < 4 || (C & 3)) return RSP_EINVAL;
  if (B == 0) return RSP_OK;
  const int64_t total = (int64_t)B * H * W * (C >> 2);
  hipLaunchKernelGGL(upsample_add_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, dst, B, h, w, H, W,
                     C);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

extern "C" int rsp_sam_embed_boxes(const float* boxes, const float* gauss, const float* pe_top_left,
                                   const float* pe_bottom_right, float* out, int32_t n, int32_t num_pos_feats,
                                   int32_t input_h, int32_t input_w, rsp_stream_t stream) {
  if (!boxes || !gauss || !pe_top_left || !pe_bottom_right || !out || n < 0 || num_pos_feats < 1 || input_h < 1 ||
      input_w < 1)
    return RSP_EINVAL;
  if (n == 0) return RSP_OK;
  hipLaunchKernelGGL(embed_boxes_kernel, dim3(grid_for((int64_t)n * 2 * num_pos_feats)), dim3(256), 0, (hipStream_t)stream,
                     boxes, gauss, pe_top_left, pe_bottom_right, out, n, num_pos_feats, (float)input_w, (float)input_h);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}
