// Row LayerNorm on [rows, C] fp32 (channels-last => also every "LayerNorm2d").
// One wave per row, row kept in registers, two-pass (mean, then centred
// variance) exactly like the reference's (x-u)/sqrt(var+eps) formulation
// (models.py:45-50, HF:147-170); HBM-bound: 1 read + 1 write per element.
#include "rsp_common.h"

namespace {

template <int NCHUNK>  // float4 chunks per lane: supports C <= NCHUNK * 256
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        float* __restrict__ y, int64_t rows, int C,
                                                        float eps, int act, half_t* __restrict__ yhi,
                                                        half_t* __restrict__ ylo, float pscale, bool f8) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * C;
  float* yr = y + row * C;
  f32x4 v[NCHUNK];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NCHUNK; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < C) {
      v[i] = *reinterpret_cast<const f32x4*>(xr + c);
      sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    } else {
      v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  sum = rsp_wave_sum(sum);
  const float mean = sum / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NCHUNK; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < C) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float dlt = v[i][j] - mean;
        sq += dlt * dlt;
      }
    }
  }
  sq = rsp_wave_sum(sq);
  const float rstd = 1.0f / sqrtf(sq / (float)C + eps);
#pragma unroll
  for (int i = 0; i < NCHUNK; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < C) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
      const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c);
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float t = (v[i][j] - mean) * rstd * g[j] + b[j];
        if (act == RSP_ACT_GELU) t = rsp_gelu(t);
        o[j] = t;
      }
      if (y) *reinterpret_cast<f32x4*>(yr + c) = o;
      if (yhi) {
        const int64_t po = ((int64_t)(c >> 5) * rows + row) * 32 + (c & 31);   // KB32 layout
        rsp_store_planes4(yhi, ylo, po, f32x4{o[0] * pscale, o[1] * pscale, o[2] * pscale, o[3] * pscale}, f8);
      }
    }
  }
}

// Plane-producing LayerNorms of the encoder (C % 64 == 0, C >= 256): FOUR consecutive rows per wave, 16 lanes per row.  A
// KB32 plane holds a row's 32-column block as 64 contiguous bytes and consecutive rows next to each other, so with one row
// per wave a store instruction wrote 8 separate 64-byte pieces; with four rows it writes two runs of 256 bytes (and a load
// reads 4 x 256 bytes).  Round 3's form ran at 3.8 TB/s (61 % of what a copy gets).
template <int NCH>  // float4 chunks per lane: C = 64 * NCH
__global__ __launch_bounds__(256) void layernorm_rows4_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ y,
                                                              int64_t rows, float eps, int act, half_t* __restrict__ yhi,
                                                              half_t* __restrict__ ylo, float pscale, bool f8) {
  constexpr int C = 64 * NCH;
  const int lane = threadIdx.x & 63, c16 = lane & 15;
  const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
  const bool rv = row < rows;
  const float* xr = x + (rv ? row : 0) * C;
  f32x4 v[NCH];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    v[i] = *reinterpret_cast<const f32x4*>(xr + (c16 + 16 * i) * 4);
    sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float mean = sum / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float dlt = v[i][j] - mean;
      sq += dlt * dlt;
    }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
  const float rstd = 1.0f / sqrtf(sq / (float)C + eps);
  if (!rv) return;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (c16 + 16 * i) * 4;
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
    const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float t = (v[i][j] - mean) * rstd * g[j] + b[j];
      if (act == RSP_ACT_GELU) t = rsp_gelu(t);
      o[j] = t;
    }
    if (y) *reinterpret_cast<f32x4*>(y + row * C + c) = o;
    if (yhi) {
      const int64_t po = ((int64_t)(c >> 5) * rows + row) * 32 + (c & 31);   // KB32 layout
      rsp_store_planes4(yhi, ylo, po, f32x4{o[0] * pscale, o[1] * pscale, o[2] * pscale, o[3] * pscale}, f8);
    }
  }
}

// ... and with EIGHT consecutive columns per lane (C % 128 == 0, fp16 planes): a lane's plane stores are 16 bytes instead of 8
// (8-byte accesses run at 0.54-0.70 of the 16-byte rate, MI355X_MICROARCH.md), a store instruction writes four runs of 256
// contiguous bytes, and the kernel issues half as many of them (round 5).
template <int NC8>  // 8-column chunks per lane: C = 128 * NC8
__global__ __launch_bounds__(256) void layernorm_rows4x8_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ y,
                                                                int64_t rows, float eps, int act, half_t* __restrict__ yhi,
                                                                half_t* __restrict__ ylo, float pscale) {
  constexpr int C = 128 * NC8;
  const int lane = threadIdx.x & 63, c16 = lane & 15;
  const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
  const bool rv = row < rows;
  const float* xr = x + (rv ? row : 0) * C;
  f32x4 v[NC8][2];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NC8; ++i) {
    v[i][0] = *reinterpret_cast<const f32x4*>(xr + (c16 + 16 * i) * 8);
    v[i][1] = *reinterpret_cast<const f32x4*>(xr + (c16 + 16 * i) * 8 + 4);
    sum += ((v[i][0][0] + v[i][0][1]) + (v[i][0][2] + v[i][0][3])) + ((v[i][1][0] + v[i][1][1]) + (v[i][1][2] + v[i][1][3]));
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float mean = sum / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NC8; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float dlt = v[i][h][j] - mean;
        sq += dlt * dlt;
      }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
  const float rstd = 1.0f / sqrtf(sq / (float)C + eps);
  if (!rv) return;
#pragma unroll
  for (int i = 0; i < NC8; ++i) {
    const int c = (c16 + 16 * i) * 8;
    half8_t h8, l8;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c + 4 * h);
      const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c + 4 * h);
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float t = (v[i][h][j] - mean) * rstd * g[j] + b[j];
        if (act == RSP_ACT_GELU) t = rsp_gelu(t);
        o[j] = t;
        half_t a, bb;
        rsp_split1(t * pscale, a, bb);
        h8[4 * h + j] = a; l8[4 * h + j] = bb;
      }
      if (y) *reinterpret_cast<f32x4*>(y + row * C + c + 4 * h) = o;
    }
    const int64_t po = ((int64_t)(c >> 5) * rows + row) * 32 + (c & 31);   // KB32 layout
    *reinterpret_cast<half8_t*>(yhi + po) = h8;
    *reinterpret_cast<half8_t*>(ylo + po) = l8;
  }
}

// C <= 64: a 16-lane group per row (4 rows per wave) so that no lane idles
__global__ __launch_bounds__(256) void layernorm_small_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta,
                                                              float* __restrict__ y, int64_t rows, int C,
                                                              float eps, int act, half_t* __restrict__ yhi,
                                                              half_t* __restrict__ ylo, float pscale) {
  const int sub = threadIdx.x & 15;
  const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const bool rv = row < rows;
  const int c = sub * 4;
  const bool ok = rv && c < C;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (ok) v = *reinterpret_cast<const f32x4*>(x + row * C + c);
  float sum = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float mean = sum / (float)C;
  float sq = 0.f;
  if (ok) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float dlt = v[j] - mean; sq += dlt * dlt; }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
  const float rstd = 1.0f / sqrtf(sq / (float)C + eps);
  if (!ok) return;
  const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
  const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c);
  f32x4 o4;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float t = (v[j] - mean) * rstd * g[j] + b[j];
    if (act == RSP_ACT_GELU) t = rsp_gelu(t);
    o4[j] = t;
  }
  if (y) *reinterpret_cast<f32x4*>(y + row * C + c) = o4;
  if (yhi) {
    half4_t h4, l4;
#pragma unroll
    for (int j = 0; j < 4; ++j) { half_t a, bb; rsp_split1(o4[j] * pscale, a, bb); h4[j] = a; l4[j] = bb; }
    const int64_t po = ((int64_t)(c >> 5) * rows + row) * 32 + (c & 31);   // KB32 layout
    *reinterpret_cast<half4_t*>(yhi + po) = h4;
    *reinterpret_cast<half4_t*>(ylo + po) = l4;
  }
}

}  // namespace

extern "C" int rsp_layernorm_ex(const float* x, const float* gamma, const float* beta, float* y,
                                uint16_t* yhi, uint16_t* ylo, int32_t scale_log2, int64_t rows, int32_t C,
                                float eps, int32_t act, rsp_stream_t stream) {
  if (!x || !gamma || !beta || rows < 0 || C <= 0 || (C & 3) || C > 2048) return RSP_EINVAL;
  if (!y && !(yhi && ylo)) return RSP_EINVAL;
  if (yhi && (C & 31)) return RSP_EINVAL;
  if ((yhi == nullptr) != (ylo == nullptr)) return RSP_EINVAL;
  if (act != RSP_ACT_NONE && act != RSP_ACT_GELU) return RSP_EINVAL;
  if (rows == 0) return RSP_OK;
  hipStream_t s = (hipStream_t)stream;
  half_t* hi = reinterpret_cast<half_t*>(yhi);
  half_t* lo = reinterpret_cast<half_t*>(ylo);
  const float ps = ldexpf(1.0f, RSP_PLANE_EXP(scale_log2));
  const bool f8 = yhi && RSP_PLANE_IS_F8(scale_log2);
  if (f8 && C <= 64) return RSP_EINVAL;   // cat8 planes feed the wide encoder GEMMs only
  if (C <= 64) {
    const int64_t blocks = (rows + 15) / 16;
    if (blocks > 0x7fffffffLL) return RSP_EINVAL;
    hipLaunchKernelGGL(layernorm_small_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, gamma, beta, y, rows, C, eps, act, hi, lo, ps);
    RSP_CHECK_LAUNCH();
    return RSP_OK;
  }
  if (yhi && (C & 63) == 0 && C >= 256 && C <= 1280) {          // the encoder's plane-producing LayerNorms
    const int64_t blocks16 = (rows + 15) / 16;
    if (blocks16 > 0x7fffffffLL) return RSP_EINVAL;
    // (ViT-H rows: 4.07 -> 4.36 TB/s; profiles/r5_layernorm_x8_vs_x4.txt).  One instantiation per width in use; every
    // other multiple of 128 (384, 640, 896, 1152) goes on to the x4 switch / the one-row-per-wave kernel below (round 5
    // sent them to the C = 1280 instantiation: wrong row stride)
    if (!f8 && (C & 127) == 0 && (C == 256 || C == 512 || C == 768 || C == 1024 || C == 1280)) {
#define RSP_LN8_LAUNCH(NC) hipLaunchKernelGGL((layernorm_rows4x8_kernel<NC>), dim3((unsigned)blocks16), dim3(256), 0, s, x, gamma, beta, y, rows, eps, act, hi, lo, ps)
      switch (C / 128) {
        case 2: RSP_LN8_LAUNCH(2); break;
        case 4: RSP_LN8_LAUNCH(4); break;
        case 6: RSP_LN8_LAUNCH(6); break;
        case 8: RSP_LN8_LAUNCH(8); break;
        default: RSP_LN8_LAUNCH(10); break;      // C == 1280
      }
#undef RSP_LN8_LAUNCH
      RSP_CHECK_LAUNCH();
      return RSP_OK;
    }
#define RSP_LN4_LAUNCH(NC) hipLaunchKernelGGL((layernorm_rows4_kernel<NC>), dim3((unsigned)blocks16), dim3(256), 0, s, x, gamma, beta, y, rows, eps, act, hi, lo, ps, f8)
    switch (C / 64) {
      case 4: RSP_LN4_LAUNCH(4); break;
      case 8: RSP_LN4_LAUNCH(8); break;
      case 12: RSP_LN4_LAUNCH(12); break;
      case 16: RSP_LN4_LAUNCH(16); break;
      case 20: RSP_LN4_LAUNCH(20); break;
      default: goto one_row_per_wave;
    }
#undef RSP_LN4_LAUNCH
    RSP_CHECK_LAUNCH();
    return RSP_OK;
  }
one_row_per_wave:
  const int64_t blocks = (rows + 3) / 4;
  if (blocks > 0x7fffffffLL) return RSP_EINVAL;
  // one instantiation per 256-channel step actually used (ViT widths 768 / 1024 / 1280): no idle chunk iterations
#define RSP_LN_LAUNCH(NC) hipLaunchKernelGGL((layernorm_kernel<NC>), dim3((unsigned)blocks), dim3(256), 0, s, x, gamma, beta, y, rows, C, eps, act, hi, lo, ps, f8)
  if (C <= 256) RSP_LN_LAUNCH(1);
  else if (C <= 512) RSP_LN_LAUNCH(2);
  else if (C <= 768) RSP_LN_LAUNCH(3);
  else if (C <= 1024) RSP_LN_LAUNCH(4);
  else if (C <= 1280) RSP_LN_LAUNCH(5);
  else RSP_LN_LAUNCH(8);
#undef RSP_LN_LAUNCH
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

extern "C" int rsp_layernorm(const float* x, const float* gamma, const float* beta, float* y,
                             int64_t rows, int32_t C, float eps, int32_t act, rsp_stream_t stream) {
  if (!y) return RSP_EINVAL;
  return rsp_layernorm_ex(x, gamma, beta, y, nullptr, nullptr, 0, rows, C, eps, act, stream);
}
