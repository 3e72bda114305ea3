// Shared device helpers for the gfx950 kernels of librsp_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rsp_hip.h"

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define RSP_WAVE 64

#define RSP_CHECK_LAUNCH()                         \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return RSP_ELAUNCH;     \
  } while (0)

// x = hi + lo with hi, lo fp16 (round-to-nearest-even both times).  The fp32
// remainder x - hi is exact, so hi+lo carries ~22 significant bits of x.
__device__ __forceinline__ void rsp_split1(float x, half_t& hi, half_t& lo) {
  hi = (half_t)x;
  lo = (half_t)(x - (float)hi);
}

__device__ __forceinline__ float rsp_gelu(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

__device__ __forceinline__ float rsp_act(float v, int act) {
  switch (act) {
    case RSP_ACT_RELU: return v > 0.f ? v : 0.f;
    case RSP_ACT_GELU: return rsp_gelu(v);
    case RSP_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    default: return v;
  }
}

__device__ __forceinline__ float rsp_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float rsp_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
