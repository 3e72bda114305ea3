// Shared device helpers for the gfx950 kernels of librsp_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rsp_hip.h"

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define RSP_WAVE 64

// Marks a point where lanes of ONE wave exchange data through LDS without a block barrier: the DS operations of a wave
// execute in order and its 64 lanes in lockstep, so the reads behind this point see the writes in front of it.  On the
// device it is __builtin_amdgcn_wave_barrier(): no instruction, but the compiler may not move an LDS read of one lane's
// address above the LDS write another lane makes to it (legal for single-thread semantics otherwise; round 5 -- the
// instruction counts of the five files that use it are unchanged, the schedules of two epilogues shift slightly).  The
// lane-level emulator the CPU tests compile these sources against (tests/wave_emu/emu_hip.h) runs lanes one at a time
// and turns it into a wave rendezvous.
#ifndef RSP_WAVE_LOCKSTEP
#define RSP_WAVE_LOCKSTEP() __builtin_amdgcn_wave_barrier()
#endif

// A 16-byte global load the COMPILER does not know to be a load (inline assembly): no automatic s_waitcnt for it, no drain of
// the outstanding operations at loop heads -- the kernel orders it with its own counted s_waitcnt (csrc/upscale.hip).  `dst` is
// any 16-byte register vector, `ptr` a generic / global pointer.  The lane emulator replaces it by a plain load + a place in
// its vmcnt order (tests/wave_emu/emu_hip.h).
#ifndef RSP_GLOBAL_LOAD_B128
#define RSP_GLOBAL_LOAD_B128(dst, ptr) \
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"((const __attribute__((address_space(1))) void*)(ptr)) : "memory")
#endif

// The LDS-DMA twin: 16 bytes per lane from `gptr` (per lane) to LDS, as inline assembly, so that hipcc -- which models the builtin
// as a write to LDS and protects later ds_reads of anything that may alias with s_waitcnt vmcnt(0) -- leaves the ordering to the
// kernel's own counted waits.  `lds` = the WAVE-UNIFORM LDS byte address as an integer (rsp_lds_addr() of an address_space(3)
// pointer, + uniform offsets): lane l lands at lds + 16 l.  M0 carries it (one wait state between the SALU write of M0 and the
// DMA instruction that reads it: the hazard hipcc pads for its own builtin).
#ifndef RSP_GLOBAL_LOAD_LDS_B128
#define RSP_GLOBAL_LOAD_LDS_B128(gptr, lds)                                                     \
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"                \
               :: "v"((const __attribute__((address_space(1))) void*)(gptr)), "s"((int)(lds)) : "memory")
#endif
// ... and its raw-buffer form (`rsrc` = __amdgpu_buffer_rsrc_t, `voff` per lane, `soff` wave-uniform, bounds-checked by the resource)
#ifndef RSP_BUFFER_LOAD_LDS_B128
#define RSP_BUFFER_LOAD_LDS_B128(rsrc, lds, voff, soff)                                          \
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"        \
               :: "v"((int)(voff)), "s"(rsrc), "s"((int)(soff)), "s"((int)(lds)) : "memory")
#endif
// the wave-uniform integer LDS address of an address_space(3) pointer (rsp_lds_addr_t; the emulator keeps pointers: tests/wave_emu/emu_hip.h)
#ifndef RSP_HAVE_LDS_ADDR
#define RSP_HAVE_LDS_ADDR 1
typedef int rsp_lds_addr_t;
__device__ __forceinline__ rsp_lds_addr_t rsp_lds_addr(const __attribute__((address_space(3))) void* p) {
  return __builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)p);
}
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

// the same split on four values, written so that the conversions come out packed (v_cvt_pk_f16_f32 for hi and lo, the
// remainder from the packed hi by v_fma_mix_f32): 4 instead of 6 instructions per value; results identical to rsp_split1
__device__ __forceinline__ void rsp_split4(const f32x4 y, half4_t& h4, half4_t& l4, f32x4& r) {
  f32x4 yh;
#pragma unroll
  for (int e = 0; e < 4; ++e) yh[e] = __builtin_fminf(__builtin_fmaxf(y[e], -RSP_F16_MAX), RSP_F16_MAX);
  h4 = __builtin_convertvector(yh, half4_t);
  // y - hi (exact) as ONE v_fma_mix_f32 per value: the fp16 half is a source of the fp32 fma, no v_cvt_f32_f16 in front
  // (the multiplier sits in a register the optimiser cannot see through; written as a subtraction it converts first)
  float m1 = -1.0f;
  asm volatile("" : "+v"(m1));
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = __builtin_fmaf((float)h4[e], m1, y[e]);
  f32x4 rc;
#pragma unroll
  for (int e = 0; e < 4; ++e) rc[e] = __builtin_fminf(__builtin_fmaxf(r[e], -RSP_F16_MAX), RSP_F16_MAX);
  l4 = __builtin_convertvector(rc, half4_t);
}

// Split of values known to be inside the fp16 range (probabilities * 2^14, scaled q): no saturation needed, and the hi
// part may be TRUNCATED -- the remainder is then non-negative and lo = rtz(x - hi) still carries the next 11 bits, so
// hi + lo holds ~21 bits, the same class as the round-to-nearest pair.  v_cvt_pkrtz_f16_f32 converts two values per
// instruction, the remainder is one v_fma_mix_f32: 2 instructions per value instead of 6.
__device__ __forceinline__ void rsp_split8_trunc(const float* x, half8_t& hi, half8_t& lo) {
  float m1 = -1.0f;
  asm volatile("" : "+v"(m1));
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    const half2_t h2 = __builtin_bit_cast(half2_t, __builtin_amdgcn_cvt_pkrtz(x[i], x[i + 1]));
    const float r0 = __builtin_fmaf((float)h2[0], m1, x[i]), r1 = __builtin_fmaf((float)h2[1], m1, x[i + 1]);
    const half2_t l2 = __builtin_bit_cast(half2_t, __builtin_amdgcn_cvt_pkrtz(r0, r1));
    hi[i] = h2[0]; hi[i + 1] = h2[1]; lo[i] = l2[0]; lo[i + 1] = l2[1];
  }
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
  f32x4 r;
  rsp_split4(v, h4, l4, r);
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

// exact-erf GELU (nn.GELU default, HF "gelu"):  gelu(x) = x Phi(x) = max(x, 0) - 0.5 |x| erfc(|x| / sqrt 2), and
//   erfc(u / sqrt 2) = 2^-P(u),  P(u) = u Q(u), Q a degree-7 polynomial fitted (weighted minimax on [0, 13.5], weight =
//   the GELU's sensitivity 0.5 u erfc ln 2) to -log2 erfc: monotone on the interval, P(13.5) = 163 so that the tail
//   underflows to exactly 0 (so the clamped u can stand in for |x| in the last product).  Branch-free: 1 transcendental + 11 VALU operations per element (round 5; rounds 1-4 used
//   Abramowitz-Stegun 7.1.26: 2 transcendentals + 15 operations), and every operation but min / max / exp2 exists as a
//   packed fp32 instruction (v_pk_fma_f32, v_pk_mul_f32), which the four-wide form below compiles to.  Measured in fp32
//   against the fp64 definition on 5 M points of [-16, 16] (tests/test_f32_gelu_cpu.py restates it in numpy): max
//   absolute error 2.7e-7 (at |x| = 4.2: half an ulp of the result; A-S: 4.7e-7), 1.5e-7 for |x| < 3 (A-S: 4.3e-7).
//   GELU sits in GEMM epilogues (encoder lin1: 128 values per lane and tile) and in the SAM upscaler, where VALU time
//   is not hidden behind matrix work: the lin1 epilogue of the ping-pong GEMM was VALU bound by it.
#define RSP_GELU_U_MAX 13.5f
#define RSP_GELU_C1 1.1511219356e+00f
#define RSP_GELU_C2 4.5908040493e-01f
#define RSP_GELU_C3 5.2851349953e-02f
#define RSP_GELU_C4 -7.5443004478e-03f
#define RSP_GELU_C5 4.8712664241e-04f
#define RSP_GELU_C6 4.6808155241e-05f
#define RSP_GELU_C7 -1.0968042093e-05f
#define RSP_GELU_C8 5.2522374980e-07f
__device__ __forceinline__ float rsp_gelu(float x) {
  const float u = __builtin_fminf(__builtin_fabsf(x), RSP_GELU_U_MAX);
  float q = __builtin_fmaf(RSP_GELU_C8, u, RSP_GELU_C7);
  q = __builtin_fmaf(q, u, RSP_GELU_C6);
  q = __builtin_fmaf(q, u, RSP_GELU_C5);
  q = __builtin_fmaf(q, u, RSP_GELU_C4);
  q = __builtin_fmaf(q, u, RSP_GELU_C3);
  q = __builtin_fmaf(q, u, RSP_GELU_C2);
  q = __builtin_fmaf(q, u, RSP_GELU_C1);
  const float e = __builtin_amdgcn_exp2f(-(q * u));       // erfc(|x| / sqrt 2)
  // (u instead of |x|: where they differ e is exactly 0.)  max(x, 0) is written 0.5 x + 0.5 |x|: the same bits for every
  // finite normal x, but a NaN input stays NaN as in torch / HF (fmaxf(NaN, 0) = 0 silently zeroed a bad activation: ADVICE r5)
  return __builtin_fmaf(-0.5f * u, e, __builtin_fmaf(0.5f, x, 0.5f * __builtin_fabsf(x)));
}
// the same arithmetic on four values (identical results element by element: fma / mul are the same operations packed)
__device__ __forceinline__ f32x4 rsp_gelu4(const f32x4 x) {
  f32x4 u;
#pragma unroll
  for (int e = 0; e < 4; ++e) u[e] = __builtin_fminf(__builtin_fabsf(x[e]), RSP_GELU_U_MAX);   // |x|: a source modifier
  const auto k = [](float c) { return f32x4{c, c, c, c}; };
  f32x4 q = __builtin_elementwise_fma(k(RSP_GELU_C8), u, k(RSP_GELU_C7));
  q = __builtin_elementwise_fma(q, u, k(RSP_GELU_C6));
  q = __builtin_elementwise_fma(q, u, k(RSP_GELU_C5));
  q = __builtin_elementwise_fma(q, u, k(RSP_GELU_C4));
  q = __builtin_elementwise_fma(q, u, k(RSP_GELU_C3));
  q = __builtin_elementwise_fma(q, u, k(RSP_GELU_C2));
  q = __builtin_elementwise_fma(q, u, k(RSP_GELU_C1));
  const f32x4 pu = q * u;
  const f32x4 hx = u * -0.5f;
  f32x4 ex, rl;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    ex[e] = __builtin_amdgcn_exp2f(-pu[e]);
    rl[e] = __builtin_fmaf(0.5f, x[e], 0.5f * __builtin_fabsf(x[e]));      // max(x, 0), NaN-propagating (see rsp_gelu)
  }
  return __builtin_elementwise_fma(hx, ex, rl);
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
