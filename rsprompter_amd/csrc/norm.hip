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
                                                        float eps, int act) {
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
      *reinterpret_cast<f32x4*>(yr + c) = o;
    }
  }
}

}  // namespace

extern "C" int rsp_layernorm(const float* x, const float* gamma, const float* beta, float* y,
                             int64_t rows, int32_t C, float eps, int32_t act,
                             rsp_stream_t stream) {
  if (!x || !gamma || !beta || !y || rows < 0 || C <= 0 || (C & 3) || C > 2048) return RSP_EINVAL;
  if (act != RSP_ACT_NONE && act != RSP_ACT_GELU) return RSP_EINVAL;
  if (rows == 0) return RSP_OK;
  const int64_t blocks = (rows + 3) / 4;
  if (blocks > 0x7fffffffLL) return RSP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (C <= 256) {
    hipLaunchKernelGGL((layernorm_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, s, x, gamma, beta, y, rows, C, eps, act);
  } else if (C <= 1024) {
    hipLaunchKernelGGL((layernorm_kernel<4>), dim3((unsigned)blocks), dim3(256), 0, s, x, gamma, beta, y, rows, C, eps, act);
  } else {
    hipLaunchKernelGGL((layernorm_kernel<8>), dim3((unsigned)blocks), dim3(256), 0, s, x, gamma, beta, y, rows, C, eps, act);
  }
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}
