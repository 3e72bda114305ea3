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

// Marks a point where lanes of ONE wave exchange data through LDS without a block barrier: the DS operations of a wave
// execute in order and its 64 lanes in lockstep, so the reads behind this point see the writes in front of it.  Expands
// to nothing on the device (no instruction, no scheduling effect); the lane-level emulator the CPU tests compile these
// sources against (tests/wave_emu/emu_hip.h) runs lanes one at a time and turns it into a wave rendezvous.
#ifndef RSP_WAVE_LOCKSTEP
#define RSP_WAVE_LOCKSTEP() ((void)0)
#endif

#define RSP_CHECK_LAUNCH()                         \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return RSP_ELAUNCH;     \
  } while (0)

// x = hi + lo with hi, lo fp16 (round-to-nearest-even both times).  The fp32
// remainder x - hi is exact, so hi+lo carries ~22 significant bits of x.
// SATURATING: |x| beyond the fp16 range does not produce inf / -inf planes (whose MFMA products are NaN for the whole
// output row): hi clamps at +-65504 and lo carries the remainder (11 bits of it) up to |x| = 131008, beyond that the
// pair saturates.  One v_med3_f32 per plane; callers pick the pre-scale so that this is the outlier path only.
constexpr float RSP_F16_MAX = 65504.0f;
__device__ __forceinline__ void rsp_split1(float x, half_t& hi, half_t& lo) {
  const float xh = __builtin_fminf(__builtin_fmaxf(x, -RSP_F16_MAX), RSP_F16_MAX);
  hi = (half_t)xh;
  const float r = x - (float)hi;
  lo = (half_t)__builtin_fminf(__builtin_fmaxf(r, -RSP_F16_MAX), RSP_F16_MAX);
}

// ---- plane stores (format word: include/rsp_hip.h "Plane format word") -------------------------------------------
// four fp32 -> four OCP e4m3 bytes (round to nearest even; clamped to +-448 first: e4m3 has no inf)
__device__ __forceinline__ uint32_t rsp_pack4_e4m3(float a, float b, float c, float d) {
  auto cl = [](float x) { return __builtin_fminf(__builtin_fmaxf(x, -448.0f), 448.0f); };
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(cl(a), cl(b), 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(cl(c), cl(d), w, true);
  return (uint32_t)w;
}
// store 4 consecutive k (k % 4 == 0) of one row: `o` = element offset inside the KB32 plane ([K/32][rows][32]);
// v already carries the plane scale 2^e.  f8 = the second plane is the cat8 plane.
__device__ __forceinline__ void rsp_store_planes4(half_t* hi, half_t* lo, int64_t o, const f32x4 v, bool f8) {
  half4_t h4, l4;
  float r[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float xh = __builtin_fminf(__builtin_fmaxf(v[e], -RSP_F16_MAX), RSP_F16_MAX);
    h4[e] = (half_t)xh;
    r[e] = v[e] - (float)h4[e];
    l4[e] = (half_t)__builtin_fminf(__builtin_fmaxf(r[e], -RSP_F16_MAX), RSP_F16_MAX);
  }
  *reinterpret_cast<half4_t*>(hi + o) = h4;
  if (!f8) {
    *reinterpret_cast<half4_t*>(lo + o) = l4;
  } else {
    constexpr float LS = (float)(1 << RSP_F8_LO_EXP), HS = 1.0f / (float)(1 << RSP_F8_HI_EXP);
    unsigned char* cb = reinterpret_cast<unsigned char*>(lo) + (o & ~(int64_t)31) * 2 + (o & 31);
    *reinterpret_cast<uint32_t*>(cb) = rsp_pack4_e4m3(r[0] * LS, r[1] * LS, r[2] * LS, r[3] * LS);
    *reinterpret_cast<uint32_t*>(cb + 32) = rsp_pack4_e4m3((float)h4[0] * HS, (float)h4[1] * HS, (float)h4[2] * HS,
                                                          (float)h4[3] * HS);
  }
}

// exact-erf GELU (nn.GELU default, HF "gelu").  erf through Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7 absolute,
// i.e. <= 1e-7 * |x| on the GELU value): branch-free, 2 transcendentals + ~12 VALU ops per element instead of the
// two-branch libm erff (~40 with divergence).  GELU sits in GEMM epilogues (encoder lin1, SAM upscaler), where
// VALU time is not hidden behind matrix work.
__device__ __forceinline__ float rsp_gelu(float x) {
  const float ax = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float pl = fmaf(1.061405429f, t, -1.453152027f);
  pl = fmaf(pl, t, 1.421413741f);
  pl = fmaf(pl, t, -0.284496736f);
  pl = fmaf(pl, t, 0.254829592f);
  pl *= t;
  const float e = __builtin_amdgcn_exp2f(ax * ax * -1.4426950408889634f);
  const float erf_abs = fmaf(-pl, e, 1.0f);               // erf(|x| / sqrt 2)
  return 0.5f * x + 0.5f * fabsf(x) * erf_abs;            // 0.5 x (1 + sign(x) erf_abs)
}

__device__ __forceinline__ float rsp_act(float v, int act) {
  switch (act) {
    case RSP_ACT_RELU: return v > 0.f ? v : 0.f;
    case RSP_ACT_GELU: return rsp_gelu(v);
    case RSP_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    default: return v;
  }
}

// RSP_ACT_RELU_POST acts after the residual has been added (rsp_act leaves the value alone for it)
__device__ __forceinline__ float rsp_act_post(float v, int act) {
  return (act == RSP_ACT_RELU_POST && v < 0.f) ? 0.f : v;
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

// unsigned division by a run-time constant prepared on the host (Granlund-Montgomery, branch-free form):
// q = (t + ((x - t) >> sh1)) >> sh2 with t = mulhi(m, x); exact for all 32-bit x.
struct FastDiv {
  uint32_t m, sh1, sh2;
  __device__ __forceinline__ int div(int x) const {
    const uint32_t t = __umulhi(m, (uint32_t)x);
    return (int)((t + (((uint32_t)x - t) >> sh1)) >> sh2);
  }
};
static inline FastDiv make_fastdiv(int dv) {
  FastDiv f{0u, 0u, 0u};
  if (dv <= 1) return f;              // q = x
  uint32_t d = (uint32_t)dv, l = 0;
  while ((1ull << l) < d) ++l;        // ceil(log2 d)
  f.m = (uint32_t)((((1ull << l) - d) << 32) / d + 1);
  f.sh1 = 1; f.sh2 = l - 1;
  return f;
}
