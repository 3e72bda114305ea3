// SAM two-way-transformer cross attentions (HF:243-288 SamAttention inside HF:306-348, 396-404), the two shapes
// that matter in the RSPrompter decoders:
//   token -> image : T <= 12 prompt tokens attend over the N = h*w image positions   (many keys, few queries)
//   image -> token : the N image positions attend over the T prompt tokens          (many queries, few keys)
// with internal width 128 = 8 heads x 16.  Both are HBM-bound streams of the per-RoI image tensors (4 MB per RoI
// and call); the MFMA flash kernel in attn.hip spends its time on 64-key tiles and barriers here, so these two are
// plain fp32 VALU kernels: exact fp32 products (no fp16 split at all), one online-softmax state per lane, merged
// across lanes once at the end.
#include "rsp_common.h"

namespace {

constexpr int NH = 8, DH = 16, W = NH * DH;   // heads, head dim, internal width
constexpr float LOG2E = 1.4426950408889634f;

// ---------------------------------------------------------------------------------------------------------------
// token -> image.  One block per RoI, wave h <-> head h, lane = 2 * key slot + dim half: every lane owns 8 of the 16
// head dims of its key (the two halves of a score meet through one DPP exchange), keys slot, slot+32, ...
//   q  [R, T, 128]            tokens (already projected)
//   kv [Rkv * N, 256]         image rows, K in columns [0,128), V in [128,256)  (one fused projection GEMM)
//   kv_map[r]                 RoI -> image row block (NULL: r)
//   out[R, T, 128]
template <int TMAX>
__global__ __launch_bounds__(512) void sam_t2i_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                      const int32_t* __restrict__ kv_map, float* __restrict__ out,
                                                      int T, int N, float scale) {
  constexpr int HD = DH / 2;
  __shared__ __attribute__((aligned(16))) float sQ[TMAX * W];
  const int r = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int h = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane & 1;
  for (int i = tid; i < TMAX * W; i += 512)
    sQ[i] = (i < T * W) ? q[(int64_t)r * T * W + i] * (scale * LOG2E) : 0.f;   // softmax in base 2
  __syncthreads();
  const int64_t rb = kv_map ? kv_map[r] : r;
  const float* kbase = kv + rb * (int64_t)N * (2 * W) + h * DH + half * HD;
  const float* qbase = &sQ[h * DH + half * HD];

  float m[TMAX], l[TMAX], acc[TMAX][HD];
#pragma unroll
  for (int t = 0; t < TMAX; ++t) {
    m[t] = -INFINITY; l[t] = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[t][d] = 0.f;
  }
  for (int key = lane >> 1; key < N; key += 32) {
    // the loop-invariant q values fit in registers next to the accumulators up to 10 tokens; beyond that they are
    // re-read from LDS (wave-uniform addresses) every key instead of being hoisted into spills
    if constexpr (TMAX > 10) asm volatile("" ::: "memory");
    const float* kr = kbase + (int64_t)key * (2 * W);
    const f32x4 k0 = *reinterpret_cast<const f32x4*>(kr), k1 = *reinterpret_cast<const f32x4*>(kr + 4);
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(kr + W), v1 = *reinterpret_cast<const f32x4*>(kr + W + 4);
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
      const f32x4 q0 = *reinterpret_cast<const f32x4*>(qbase + t * W);
      const f32x4 q1 = *reinterpret_cast<const f32x4*>(qbase + t * W + 4);
      float s = q0[0] * k0[0] + q0[1] * k0[1] + q0[2] * k0[2] + q0[3] * k0[3] +
                q1[0] * k1[0] + q1[1] * k1[1] + q1[2] * k1[2] + q1[3] * k1[3];
      s += __shfl_xor(s, 1, 64);                               // the other 8 dims of the same key
      const float mn = fmaxf(m[t], s);
      const float a = __builtin_amdgcn_exp2f(m[t] - mn);      // 0 on the first key (m = -inf)
      const float p = __builtin_amdgcn_exp2f(s - mn);
      m[t] = mn;
      l[t] = l[t] * a + p;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[t][e] = acc[t][e] * a + p * v0[e];
        acc[t][4 + e] = acc[t][4 + e] * a + p * v1[e];
      }
    }
  }
  // merge the 32 per-key-slot states of this (head, dim half) with a butterfly, then slot 0 normalises and stores
  for (int o = 32; o > 1; o >>= 1) {
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
      const float mo = __shfl_xor(m[t], o, 64);
      const float lo = __shfl_xor(l[t], o, 64);
      const float mn = fmaxf(m[t], mo);
      const float a = (m[t] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m[t] - mn);
      const float b = (mo == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mo - mn);
      l[t] = l[t] * a + lo * b;
#pragma unroll
      for (int d = 0; d < HD; ++d) {
        const float ao = __shfl_xor(acc[t][d], o, 64);
        acc[t][d] = acc[t][d] * a + ao * b;
      }
      m[t] = mn;
    }
  }
  if (lane < 2) {
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
      if (t < T) {
        const float inv = 1.0f / l[t];
        float* dst = out + ((int64_t)r * T + t) * W + h * DH + half * HD;
        f32x4 o0, o1;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o0[e] = acc[t][e] * inv; o1[e] = acc[t][4 + e] * inv; }
        *reinterpret_cast<f32x4*>(dst) = o0;
        *reinterpret_cast<f32x4*>(dst + 4) = o1;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// image -> token.  Block = 256 threads = 32 image positions x 8 heads per pass (lane&7 = head: a wave reads 8 whole
// rows = 4 KiB contiguous), PIX_PER_BLOCK positions per block so that the RoI's K/V (T x 128 each) is staged once.
//   q   [Rq * N, 128]   image-side queries; q_map[r] -> row block (NULL: r)
//   k,v [R, T, 128]     tokens
//   out [R * N, 128] fp32 (optional) and/or fp16 planes (KB32 [4][R*N][32], hi/lo, value * 2^e)
constexpr int PIX_PER_BLOCK = 128;
template <int TMAX>
__global__ __launch_bounds__(256) void sam_i2t_kernel(const float* __restrict__ q, const int32_t* __restrict__ q_map,
                                                      const float* __restrict__ k, const float* __restrict__ v,
                                                      float* __restrict__ out, half_t* __restrict__ ohi,
                                                      half_t* __restrict__ olo, float pscale, int T, int N,
                                                      int64_t out_rows, float scale) {
  __shared__ __attribute__((aligned(16))) float sK[TMAX * W];
  __shared__ __attribute__((aligned(16))) float sV[TMAX * W];
  const int r = blockIdx.y;
  const int tid = threadIdx.x;
  for (int i = tid; i < TMAX * W; i += 256) {
    const bool ok = i < T * W;
    sK[i] = ok ? k[(int64_t)r * T * W + i] * (scale * LOG2E) : 0.f;
    sV[i] = ok ? v[(int64_t)r * T * W + i] : 0.f;
  }
  __syncthreads();
  const int h = tid & 7;
  const int64_t qb = q_map ? q_map[r] : r;
  const int p0 = blockIdx.x * PIX_PER_BLOCK;
  for (int pp = tid >> 3; pp < PIX_PER_BLOCK; pp += 32) {
    const int pix = p0 + pp;
    if (pix >= N) break;
    const float* qr = q + (qb * N + pix) * (int64_t)W + h * DH;
    f32x4 qv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) qv[c] = *reinterpret_cast<const f32x4*>(qr + 4 * c);
    float s[TMAX];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
      float a = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 kk = *reinterpret_cast<const f32x4*>(&sK[t * W + h * DH + 4 * c]);
        a += qv[c][0] * kk[0] + qv[c][1] * kk[1] + qv[c][2] * kk[2] + qv[c][3] * kk[3];
      }
      s[t] = (t < T) ? a : -INFINITY;
      mx = fmaxf(mx, s[t]);
    }
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) { s[t] = __builtin_amdgcn_exp2f(s[t] - mx); sum += s[t]; }
    const float inv = 1.0f / sum;
    f32x4 o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 vv = *reinterpret_cast<const f32x4*>(&sV[t * W + h * DH + 4 * c]);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[c][e] += s[t] * vv[e];
      }
    const int64_t orow = (int64_t)r * N + pix;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[c][e] *= inv;
      if (out) *reinterpret_cast<f32x4*>(out + orow * W + h * DH + 4 * c) = o[c];
      if (ohi) {
        half4_t h4, l4;
#pragma unroll
        for (int e = 0; e < 4; ++e) { half_t a, b; rsp_split1(o[c][e] * pscale, a, b); h4[e] = a; l4[e] = b; }
        const int col = h * DH + 4 * c;
        const int64_t po = ((int64_t)(col >> 5) * out_rows + orow) * 32 + (col & 31);   // KB32
        *reinterpret_cast<half4_t*>(ohi + po) = h4;
        *reinterpret_cast<half4_t*>(olo + po) = l4;
      }
    }
  }
}

}  // namespace

extern "C" int rsp_sam_t2i_attention(const float* q, const float* kv, const int32_t* kv_map, float* out, int32_t R,
                                     int32_t T, int32_t N, float scale, rsp_stream_t stream) {
  if (!q || !kv || !out || R < 0 || T <= 0 || T > 12 || N <= 0) return RSP_EINVAL;   // T > 12: rsp_attention
  if (R == 0) return RSP_OK;
  hipStream_t s = (hipStream_t)stream;
  if (T <= 8) hipLaunchKernelGGL((sam_t2i_kernel<8>), dim3(R), dim3(512), 0, s, q, kv, kv_map, out, T, N, scale);
  else if (T <= 10) hipLaunchKernelGGL((sam_t2i_kernel<10>), dim3(R), dim3(512), 0, s, q, kv, kv_map, out, T, N, scale);
  else hipLaunchKernelGGL((sam_t2i_kernel<12>), dim3(R), dim3(512), 0, s, q, kv, kv_map, out, T, N, scale);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

extern "C" int rsp_sam_i2t_attention(const float* q, const int32_t* q_map, const float* k, const float* v, float* out,
                                     uint16_t* out_hi, uint16_t* out_lo, int32_t out_scale_log2, int32_t R, int32_t T,
                                     int32_t N, float scale, rsp_stream_t stream) {
  if (!q || !k || !v || (!out && !(out_hi && out_lo)) || R < 0 || T <= 0 || T > 16 || N <= 0) return RSP_EINVAL;
  if ((out_hi == nullptr) != (out_lo == nullptr)) return RSP_EINVAL;
  if (R == 0) return RSP_OK;
  if (R > 65535) return RSP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((N + PIX_PER_BLOCK - 1) / PIX_PER_BLOCK, R);
  half_t* hi = reinterpret_cast<half_t*>(out_hi);
  half_t* lo = reinterpret_cast<half_t*>(out_lo);
  const float ps = ldexpf(1.0f, out_scale_log2);
  const int64_t rows = (int64_t)R * N;
  if (T <= 8) hipLaunchKernelGGL((sam_i2t_kernel<8>), grid, dim3(256), 0, s, q, q_map, k, v, out, hi, lo, ps, T, N, rows, scale);
  else if (T <= 12) hipLaunchKernelGGL((sam_i2t_kernel<12>), grid, dim3(256), 0, s, q, q_map, k, v, out, hi, lo, ps, T, N, rows, scale);
  else hipLaunchKernelGGL((sam_i2t_kernel<16>), grid, dim3(256), 0, s, q, q_map, k, v, out, hi, lo, ps, T, N, rows, scale);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}
