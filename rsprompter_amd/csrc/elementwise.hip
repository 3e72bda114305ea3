// HBM-bound data-movement kernels of the path (pre-process, patchify, pooling).
#include "rsp_common.h"

namespace {

// out[(b*gh + py)*gw + px, c*p*p + ky*p + kx] = img[b, c, py*p + ky, px*p + kx]
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img,
                                                       float* __restrict__ out, int B, int C,
                                                       int H, int W, int p, int64_t total4) {
  const int gh = H / p, gw = W / p;
  const int K = C * p * p;
  const int k4n = K / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / k4n;
    const int k = (int)(i - row * k4n) * 4;
    const int c = k / (p * p);
    const int rem = k - c * p * p;
    const int ky = rem / p, kx = rem - ky * p;
    const int px = (int)(row % gw);
    const int64_t t = row / gw;
    const int py = (int)(t % gh);
    const int b = (int)(t / gh);
    const float* src = img + (((int64_t)b * C + c) * H + (py * p + ky)) * W + px * p + kx;
    *reinterpret_cast<f32x4*>(out + row * K + k) = *reinterpret_cast<const f32x4*>(src);
  }
}

// DetDataPreprocessor: dst[b, c, y, x] = (src[c', y, x] - mean[c]) / std[c] inside the image,
// pad_value outside; c' = 2 - c when swap_rb (BGR -> RGB happens BEFORE normalisation, so
// mean/std index the output channel).  One launch per image (sources are separate tensors).
template <typename T>
__global__ __launch_bounds__(256) void preprocess_kernel(const T* __restrict__ src,
                                                         float* __restrict__ dst, int H, int W,
                                                         int Hp, int Wp, float m0, float m1,
                                                         float m2, float s0, float s1, float s2,
                                                         int swap_rb, float pad_value) {
  const int64_t total = (int64_t)3 * Hp * Wp;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % Wp);
    const int64_t t = i / Wp;
    const int y = (int)(t % Hp);
    const int c = (int)(t / Hp);
    float v = pad_value;
    if (y < H && x < W) {
      const int cs = swap_rb ? 2 - c : c;
      const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
      const float sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
      v = ((float)src[((int64_t)cs * H + y) * W + x] - mean) / sd;
    }
    dst[i] = v;
  }
}

}  // namespace

static inline int grid_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

extern "C" int rsp_patchify(const float* img, float* out, int32_t B, int32_t C, int32_t H,
                            int32_t W, int32_t patch, rsp_stream_t stream) {
  if (!img || !out || B <= 0 || C <= 0 || patch <= 0 || (patch & 3) || H % patch || W % patch)
    return RSP_EINVAL;
  const int64_t total4 = (int64_t)B * (H / patch) * (W / patch) * C * patch * patch / 4;
  hipLaunchKernelGGL(patchify_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, img,
                     out, B, C, H, W, patch, total4);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

extern "C" int rsp_preprocess(const void* src, int32_t src_is_u8, float* dst, int32_t H, int32_t W,
                              int32_t Hp, int32_t Wp, const float* mean3, const float* std3,
                              int32_t swap_rb, float pad_value, rsp_stream_t stream) {
  if (!src || !dst || !mean3 || !std3 || H <= 0 || W <= 0 || Hp < H || Wp < W) return RSP_EINVAL;
  const int64_t total = (int64_t)3 * Hp * Wp;
  hipStream_t s = (hipStream_t)stream;
  if (src_is_u8) {
    hipLaunchKernelGGL((preprocess_kernel<uint8_t>), dim3(grid_for(total)), dim3(256), 0, s,
                       (const uint8_t*)src, dst, H, W, Hp, Wp, mean3[0], mean3[1], mean3[2], std3[0],
                       std3[1], std3[2], swap_rb, pad_value);
  } else {
    hipLaunchKernelGGL((preprocess_kernel<float>), dim3(grid_for(total)), dim3(256), 0, s,
                       (const float*)src, dst, H, W, Hp, Wp, mean3[0], mean3[1], mean3[2], std3[0],
                       std3[1], std3[2], swap_rb, pad_value);
  }
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

namespace {

// NHWC 2x2/s2 max pooling (mode 0) or stride-2 subsampling = max_pool2d(k=1, s=2) (mode 1)
__global__ __launch_bounds__(256) void pool2_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                    int B, int H, int W, int C, int mode) {
  const int Ho = mode == 0 ? H / 2 : (H + 1) / 2, Wo = mode == 0 ? W / 2 : (W + 1) / 2;
  const int c4n = C / 4;
  const int64_t total = (int64_t)B * Ho * Wo * c4n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    int64_t t = i / c4n;
    const int xo = (int)(t % Wo); t /= Wo;
    const int yo = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const float* p = x + (((int64_t)b * H + 2 * yo) * W + 2 * xo) * C + c;
    f32x4 v = *reinterpret_cast<const f32x4*>(p);
    if (mode == 0) {
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(p + C);
      const f32x4 v2 = *reinterpret_cast<const f32x4*>(p + (int64_t)W * C);
      const f32x4 v3 = *reinterpret_cast<const f32x4*>(p + (int64_t)W * C + C);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaxf(fmaxf(v[j], v1[j]), fmaxf(v2[j], v3[j]));
    }
    *reinterpret_cast<f32x4*>(y + i * 4) = v;
  }
}

// y[r, c] = x[r, c] + v[(r % vmod), c]   (vmod rows of v; vmod==1: per-channel vector)
__global__ __launch_bounds__(256) void add_rows_kernel(const float* __restrict__ x,
                                                       const float* __restrict__ v,
                                                       float* __restrict__ y, int64_t rows, int C,
                                                       int vmod) {
  const int c4n = C / 4;
  const int64_t total = rows * c4n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / c4n;
    const int c = (int)(i - r * c4n) * 4;
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + r * C + c);
    const f32x4 b = *reinterpret_cast<const f32x4*>(v + (r % vmod) * C + c);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = a[j] + b[j];
    *reinterpret_cast<f32x4*>(y + r * C + c) = o;
  }
}

// point-embedding post-processing (models.py:1670-1672): x [R, n*2c] viewed [R, n, 2c];
// y[r, n, j] = sin(x[r, n, 2j]) + x[r, n, 2j+1]
__global__ __launch_bounds__(256) void sincos_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                     int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    y[i] = sinf(x[2 * i]) + x[2 * i + 1];
  }
}

// copy rows: dst[i, :] = src[idx[i], :]   (idx < 0 -> zeros)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src,
                                                          const int32_t* __restrict__ idx,
                                                          float* __restrict__ dst, int64_t rows, int C) {
  const int c4n = C / 4;
  const int64_t total = rows * c4n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / c4n;
    const int c = (int)(i - r * c4n) * 4;
    const int s = idx[r];
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (s >= 0) v = *reinterpret_cast<const f32x4*>(src + (int64_t)s * C + c);
    *reinterpret_cast<f32x4*>(dst + r * C + c) = v;
  }
}

}  // namespace

extern "C" int rsp_pool2(const float* x, float* y, int32_t B, int32_t H, int32_t W, int32_t C,
                         int32_t mode, rsp_stream_t stream) {
  if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || (mode != 0 && mode != 1)) return RSP_EINVAL;
  if (mode == 0 && ((H | W) & 1)) return RSP_EINVAL;
  const int Ho = mode == 0 ? H / 2 : (H + 1) / 2, Wo = mode == 0 ? W / 2 : (W + 1) / 2;
  const int64_t total = (int64_t)B * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(pool2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W, C, mode);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

extern "C" int rsp_add_rows(const float* x, const float* v, float* y, int64_t rows, int32_t C,
                            int32_t vmod, rsp_stream_t stream) {
  if (!x || !v || !y || rows < 0 || C <= 0 || (C & 3) || vmod <= 0) return RSP_EINVAL;
  if (rows == 0) return RSP_OK;
  hipLaunchKernelGGL(add_rows_kernel, dim3(grid_for(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, v, y, rows, C, vmod);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

extern "C" int rsp_sincos_pairs(const float* x, float* y, int64_t n_out, rsp_stream_t stream) {
  if (!x || !y || n_out < 0) return RSP_EINVAL;
  if (n_out == 0) return RSP_OK;
  hipLaunchKernelGGL(sincos_kernel, dim3(grid_for(n_out)), dim3(256), 0, (hipStream_t)stream, x, y, n_out);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

extern "C" int rsp_gather_rows(const float* src, const int32_t* idx, float* dst, int64_t rows,
                               int32_t C, rsp_stream_t stream) {
  if (!src || !idx || !dst || rows < 0 || C <= 0 || (C & 3)) return RSP_EINVAL;
  if (rows == 0) return RSP_OK;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, src, idx, dst, rows, C);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

namespace {
__global__ void div_boxes_kernel(const float* __restrict__ b, float* __restrict__ o, int64_t n4, float s0,
                                 float s1, float s2, float s3) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const int j = (int)(i & 3);
  const float s = j == 0 ? s0 : (j == 1 ? s1 : (j == 2 ? s2 : s3));
  o[i] = b[i] / s;
}
}  // namespace

/* bboxes /= scale_factor.repeat(2)  (models.py:1763-1764) */
extern "C" int rsp_div_boxes(const float* boxes, float* out, int64_t n, const float* sf4, rsp_stream_t stream) {
  if (!boxes || !out || !sf4 || n < 0) return RSP_EINVAL;
  if (n == 0) return RSP_OK;
  hipLaunchKernelGGL(div_boxes_kernel, dim3((unsigned)((n * 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     boxes, out, n * 4, sf4[0], sf4[1], sf4[2], sf4[3]);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

namespace {
// rows[i] of a GEMM output that the GEMM itself does not write (zero A rows: window padding) = bias: fp32 columns
// [0, c_ncols) of C and the plane columns [pl_col0, N) -- exactly what alpha * 0 + bias gives in the GEMM epilogue
__global__ __launch_bounds__(256) void fill_bias_rows_kernel(const float* __restrict__ bias, const int32_t* __restrict__ rows,
                                                             int n_rows, int N, float* __restrict__ C, int ldc, int c_ncols,
                                                             half_t* __restrict__ hi, half_t* __restrict__ lo, int64_t c_rows,
                                                             int pl_col0, float cs, bool f8) {
  const int q4 = N >> 2;
  const int64_t total = (int64_t)n_rows * q4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int col = (int)(i % q4) * 4;
    const int64_t r = rows[i / q4];
    const f32x4 b = bias ? *reinterpret_cast<const f32x4*>(bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
    if (C && (c_ncols <= 0 || col < c_ncols)) *reinterpret_cast<f32x4*>(C + r * ldc + col) = b;
    if (hi && col >= pl_col0) {
      const int pch = col - pl_col0;
      const int64_t po = ((int64_t)(pch >> 5) * c_rows + r) * 32 + (pch & 31);
      rsp_store_planes4(hi, lo, po, f32x4{b[0] * cs, b[1] * cs, b[2] * cs, b[3] * cs}, f8);
    }
  }
}
}  // namespace

extern "C" int rsp_fill_bias_rows(const float* bias, const int32_t* rows, int32_t n_rows, int32_t N, float* C, int32_t ldc,
                                  int32_t c_ncols, uint16_t* Chi, uint16_t* Clo, int64_t c_rows, int32_t pl_col0,
                                  int32_t c_scale_log2, rsp_stream_t stream) {
  if (!rows || n_rows < 0 || N <= 0 || (N & 3) || (c_ncols & 3) || (pl_col0 & 31) || pl_col0 < 0 || (C && ldc < (c_ncols > 0 ? c_ncols : N)) ||
      (Chi && (!Clo || c_rows <= 0)) || !RSP_PLANE_WORD_VALID(c_scale_log2))
    return RSP_EINVAL;
  if (n_rows == 0 || (!C && !Chi)) return RSP_OK;
  hipLaunchKernelGGL(fill_bias_rows_kernel, dim3(grid_for((int64_t)n_rows * (N >> 2))), dim3(256), 0, (hipStream_t)stream,
                     bias, rows, n_rows, N, C, ldc, c_ncols, reinterpret_cast<half_t*>(Chi), reinterpret_cast<half_t*>(Clo),
                     c_rows, pl_col0, ldexpf(1.0f, RSP_PLANE_EXP(c_scale_log2)), RSP_PLANE_IS_F8(c_scale_log2));
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

namespace {
__global__ void scale_boxes_kernel(const float* __restrict__ b, float* __restrict__ o, int64_t n4, float s0, float s1,
                                   float s2, float s3) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const int j = (int)(i & 3);
  o[i] = b[i] * (j == 0 ? s0 : (j == 1 ? s1 : (j == 2 ? s2 : s3)));
}
}  // namespace

/* scale_boxes (mmdet/structures/bbox/transforms.py:391-414): boxes * factor.repeat(2); the R-CNN head's rescale
   multiplies by fp32(1 / scale_factor) (bbox_head.py:549-552), which is not the same rounding as a division */
extern "C" int rsp_scale_boxes(const float* boxes, float* out, int64_t n, const float* f4, rsp_stream_t stream) {
  if (!boxes || !out || !f4 || n < 0) return RSP_EINVAL;
  if (n == 0) return RSP_OK;
  hipLaunchKernelGGL(scale_boxes_kernel, dim3((unsigned)((n * 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, boxes,
                     out, n * 4, f4[0], f4[1], f4[2], f4[3]);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

namespace {
// bool bytes -> bits, little-endian within a byte (== numpy.packbits(bitorder='little'))
__global__ __launch_bounds__(256) void pack_bits_kernel(const uint8_t* __restrict__ src,
                                                        uint8_t* __restrict__ dst, int64_t nbytes_out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nbytes_out;
       i += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long x = *reinterpret_cast<const unsigned long long*>(src + i * 8);
    unsigned int o = 0;
#pragma unroll
    for (int b = 0; b < 8; ++b) o |= (unsigned int)(((x >> (8 * b)) & 0xffull) != 0ull) << b;
    dst[i] = (uint8_t)o;
  }
}
}  // namespace

/* instance masks bool[k,H,W] -> bit-packed u8[k*H*W/8] for the result all-gather */
extern "C" int rsp_pack_bits(const uint8_t* src, uint8_t* dst, int64_t n_bits, rsp_stream_t stream) {
  if (!src || !dst || n_bits < 0 || (n_bits & 7)) return RSP_EINVAL;
  if (n_bits == 0) return RSP_OK;
  hipLaunchKernelGGL(pack_bits_kernel, dim3(grid_for(n_bits / 8)), dim3(256), 0, (hipStream_t)stream, src, dst,
                     n_bits / 8);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// Test-pipeline front end: `Resize(scale, keep_ratio=True)` + `Pad(size, pad_val)` of
// configs/rsprompter/_base_/rsprompter_anchor.py:231-241 (mmcv.imrescale -> cv2.resize INTER_LINEAR on the float32
// image, mmcv.impad constant border) and, optionally fused, the DetDataPreprocessor arithmetic (BGR->RGB,
// (x - mean) / std, data_preprocessor.py:110-149).  src: one decoded image, HWC interleaved (what cv2 / PIL hand
// over), uint8 or fp32;  dst: [3, Hp, Wp] planar fp32.  Pixel (y, x) of the resized [Hn, Wn] region is cv2's
//   fx = (x + 0.5) * (W / Wn) - 0.5, sx = floor(fx), fx -= sx; sx < 0 -> (0, 0); sx >= W - 1 -> (W - 1, 0)
// (same for y) blended in fp32 horizontally first, then vertically -- cv2's resizeGeneric_ order; everything right of /
// below the resized region is the per-channel pad value.  normalise = 0: dst keeps the source channel order and
// range (what PackDetInputs hands to the model); 1: dst channel c = (src[swap ? 2 - c : c] - mean[c]) / std[c] and
// the padding is normalised the same way (what the data preprocessor would make of it).
namespace {

template <typename T>
__global__ __launch_bounds__(256) void resize_pad_kernel(const T* __restrict__ src, float* __restrict__ dst, int H,
                                                         int W, int Hn, int Wn, int Hp, int Wp, float p0, float p1,
                                                         float p2, int normalise, int swap_rb, float m0, float m1,
                                                         float m2, float s0, float s1, float s2) {
  const double sx_scale = (double)W / (double)Wn, sy_scale = (double)H / (double)Hn;   // cv2: double scales
  const int64_t total = (int64_t)Hp * Wp;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % Wp), y = (int)(i / Wp);
    float v[3] = {p0, p1, p2};
    if (y < Hn && x < Wn) {
      float fx = (float)(((double)x + 0.5) * sx_scale - 0.5);
      int x0 = (int)floorf(fx);
      fx -= (float)x0;
      if (x0 < 0) { x0 = 0; fx = 0.f; }
      if (x0 >= W - 1) { x0 = W - 1; fx = 0.f; }
      float fy = (float)(((double)y + 0.5) * sy_scale - 0.5);
      int y0 = (int)floorf(fy);
      fy -= (float)y0;
      if (y0 < 0) { y0 = 0; fy = 0.f; }
      if (y0 >= H - 1) { y0 = H - 1; fy = 0.f; }
      const int x1 = x0 + 1 < W ? x0 + 1 : W - 1, y1 = y0 + 1 < H ? y0 + 1 : H - 1;
      const T* r0 = src + ((int64_t)y0 * W) * 3;
      const T* r1 = src + ((int64_t)y1 * W) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float top = (float)r0[x0 * 3 + c] * (1.f - fx) + (float)r0[x1 * 3 + c] * fx;
        const float bot = (float)r1[x0 * 3 + c] * (1.f - fx) + (float)r1[x1 * 3 + c] * fx;
        v[c] = top * (1.f - fy) + bot * fy;
      }
    }
    if (normalise) {
      const float a = swap_rb ? v[2] : v[0], b = v[1], c2 = swap_rb ? v[0] : v[2];
      v[0] = (a - m0) / s0; v[1] = (b - m1) / s1; v[2] = (c2 - m2) / s2;
    }
    dst[i] = v[0];
    dst[total + i] = v[1];
    dst[2 * total + i] = v[2];
  }
}

}  // namespace

extern "C" int rsp_resize_pad(const void* src, int32_t src_is_u8, float* dst, int32_t H, int32_t W, int32_t Hn,
                              int32_t Wn, int32_t Hp, int32_t Wp, const float* pad3, int32_t normalise,
                              int32_t swap_rb, const float* mean3, const float* std3, rsp_stream_t stream) {
  if (!src || !dst || !pad3 || H <= 0 || W <= 0 || Hn <= 0 || Wn <= 0 || Hp < Hn || Wp < Wn) return RSP_EINVAL;
  if (normalise && (!mean3 || !std3)) return RSP_EINVAL;
  const float m0 = normalise ? mean3[0] : 0.f, m1 = normalise ? mean3[1] : 0.f, m2 = normalise ? mean3[2] : 0.f;
  const float s0 = normalise ? std3[0] : 1.f, s1 = normalise ? std3[1] : 1.f, s2 = normalise ? std3[2] : 1.f;
  hipStream_t s = (hipStream_t)stream;
  const int64_t total = (int64_t)Hp * Wp;
  if (src_is_u8) {
    hipLaunchKernelGGL((resize_pad_kernel<uint8_t>), dim3(grid_for(total)), dim3(256), 0, s, (const uint8_t*)src, dst,
                       H, W, Hn, Wn, Hp, Wp, pad3[0], pad3[1], pad3[2], normalise, swap_rb, m0, m1, m2, s0, s1, s2);
  } else {
    hipLaunchKernelGGL((resize_pad_kernel<float>), dim3(grid_for(total)), dim3(256), 0, s, (const float*)src, dst, H,
                       W, Hn, Wn, Hp, Wp, pad3[0], pad3[1], pad3[2], normalise, swap_rb, m0, m1, m2, s0, s1, s2);
  }
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}
