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


// ---------------------------------------------------------------------------------------------------------------
// image -> token attention FUSED with its out_proj, the residual and the LayerNorm that follow it
// (HF:340-348: keys = layer_norm4(keys + out_proj(attention(q = keys + pe, k = queries + pe, v = queries)))).
// out_proj is folded into the values: with p[h][t] the softmax of head h,
//   out_proj(attn)[c] = sum_h sum_t p[h][t] * Vp[t][h][c] + b[c],   Vp[t][h][c] = sum_{d in head h} v[t][d] * Wo[c][d]
// (exact algebra, fp32 throughout), so the [R*N, 128] attention output, the K = 128 GEMM over 3.3 M rows that consumed
// it, the fp32 [R*N, 256] result and the separate LayerNorm pass all disappear: per image position the kernel reads its
// 128-wide query row and its 256-wide residual row and writes the normalised 256-wide row as fp16 planes.
// Block = 1024 threads = 128 positions x 8 lanes; lane l8 of a position first plays head l8 (scores, softmax), then
// owns output channels {32 j + 4 l8 + e}: the 8 lanes of a position read 128 contiguous bytes of a Vp row (LDS, no bank
// conflict, the other 7 positions of the wave get the same addresses broadcast) and write one 64-byte plane row per
// 32-channel block.  Vp (T x 8 x 256 floats) is built once per block in LDS.
struct I2tFusedP {
  const float* q; const int32_t* q_map; const float* k; const float* v;
  const float* wo; const float* bo;
  const float* res; const int32_t* res_map;
  const half_t* res_hi; const half_t* res_lo; float res_inv_scale; int64_t res_rows;
  const float* gamma; const float* beta; float eps;
  float* out; half_t* ohi; half_t* olo; float pscale; int64_t out_rows;
  int T, N; float scale;
};
constexpr int FUSED_PIX_PER_BLOCK = 512, CO = 256;

template <int TMAX>
__global__ __launch_bounds__(1024) void sam_i2t_fused_kernel(const I2tFusedP p) {
  constexpr int PST = TMAX * NH + 8;               // probability row pitch: 8 positions of a wave land on distinct banks
  __shared__ __attribute__((aligned(16))) float sK[TMAX * W];
  __shared__ __attribute__((aligned(16))) float sV[TMAX * W];
  __shared__ __attribute__((aligned(16))) float sVp[TMAX * NH * CO];
  __shared__ __attribute__((aligned(16))) float sP[128 * PST];            // softmax probabilities [position][t][head]
  __shared__ __attribute__((aligned(16))) float sBo[CO], sG[CO], sB[CO];   // out_proj bias, LayerNorm gamma / beta
  const int r = blockIdx.y;
  const int tid = threadIdx.x;
  const int T = p.T, N = p.N;
  if (tid < CO) { sBo[tid] = p.bo[tid]; sG[tid] = p.gamma[tid]; sB[tid] = p.beta[tid]; }
  for (int i = tid; i < TMAX * W; i += 1024) {
    const bool ok = i < T * W;
    sK[i] = ok ? p.k[(int64_t)r * T * W + i] * (p.scale * LOG2E) : 0.f;
    sV[i] = ok ? p.v[(int64_t)r * T * W + i] : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < TMAX * NH * CO; i += 1024) {
    const int c = i & (CO - 1), th = i >> 8, h = th & 7, t = th >> 3;
    const float* wr = p.wo + (int64_t)c * W + h * DH;
    const float* vr = &sV[t * W + h * DH];
    float a = 0.f;
#pragma unroll
    for (int d4 = 0; d4 < 4; ++d4) {
      const f32x4 w4 = *reinterpret_cast<const f32x4*>(wr + 4 * d4);
      const f32x4 v4 = *reinterpret_cast<const f32x4*>(vr + 4 * d4);
      a += w4[0] * v4[0] + w4[1] * v4[1] + w4[2] * v4[2] + w4[3] * v4[3];
    }
    sVp[i] = a;                                   // rows t >= T are zero (sV is)
  }
  __syncthreads();
  const int l8 = tid & 7;
  const int64_t qb = p.q_map ? p.q_map[r] : r;
  const int64_t rb = p.res_map ? p.res_map[r] : r;
  const int p0 = blockIdx.x * FUSED_PIX_PER_BLOCK;
  float* myP = &sP[(tid >> 3) * PST];
  for (int pp = tid >> 3; pp < FUSED_PIX_PER_BLOCK; pp += 128) {
    const bool live = p0 + pp < N;                 // whole 8-lane groups are live or idle together
    const int pix = live ? p0 + pp : N - 1;
    // ---- scores and softmax of head l8 (as sam_i2t_kernel) ----
    {
      const float* qr = p.q + (qb * N + pix) * (int64_t)W + l8 * DH;
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
          const f32x4 kk = *reinterpret_cast<const f32x4*>(&sK[t * W + l8 * DH + 4 * c]);
          a += qv[c][0] * kk[0] + qv[c][1] * kk[1] + qv[c][2] * kk[2] + qv[c][3] * kk[3];
        }
        s[t] = (t < T) ? a : -INFINITY;
        mx = fmaxf(mx, s[t]);
      }
      float sum = 0.f;
#pragma unroll
      for (int t = 0; t < TMAX; ++t) { s[t] = __builtin_amdgcn_exp2f(s[t] - mx); sum += s[t]; }
      const float inv = 1.0f / sum;
#pragma unroll
      for (int t = 0; t < TMAX; ++t) myP[t * NH + l8] = s[t] * inv;
    }
    // the 8 lanes of a position sit in one wave: the exchange through LDS needs no block barrier
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- 256 output channels of the position: lane l8 accumulates channels 32 j + 4 l8 + e over all (token, head).
    //      A real loop (no unrolling): unrolled, the compiler loads every Vp row up front and spills them. ----
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = *reinterpret_cast<const f32x4*>(&sBo[32 * j + 4 * l8]);
#pragma unroll 1
    for (int th = 0; th < TMAX * NH; ++th) {
      const float pv = myP[th];                                   // same address for the 8 lanes: broadcast
      const float* vp = &sVp[th * CO + 4 * l8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const f32x4 vv = *reinterpret_cast<const f32x4*>(vp + 32 * j);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[j][e] = __builtin_fmaf(pv, vv[e], acc[j][e]);   // (file builds with -ffp-contract=off)
      }
    }
    __builtin_amdgcn_wave_barrier();              // everybody has read myP before the next pass overwrites it
    // ---- residual ----
    const int64_t orow = (int64_t)r * N + pix;
    if (p.res) {
      const float* rr = p.res + (rb * N + pix) * (int64_t)CO + 4 * l8;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const f32x4 rv = *reinterpret_cast<const f32x4*>(rr + 32 * j);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[j][e] += rv[e];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int64_t ro = ((int64_t)j * p.res_rows + orow) * 32 + 4 * l8;
        const half4_t rh = *reinterpret_cast<const half4_t*>(p.res_hi + ro);
        const half4_t rl = *reinterpret_cast<const half4_t*>(p.res_lo + ro);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[j][e] += ((float)rh[e] + (float)rl[e]) * p.res_inv_scale;
      }
    }
    // ---- LayerNorm over the 256 channels (two-pass, like layernorm_kernel) ----
    float sm = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) sm += (acc[j][0] + acc[j][1]) + (acc[j][2] + acc[j][3]);
    sm += __shfl_xor(sm, 1, 64); sm += __shfl_xor(sm, 2, 64); sm += __shfl_xor(sm, 4, 64);
    const float mean = sm * (1.0f / CO);
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float dl = acc[j][e] - mean; sq += dl * dl; }
    sq += __shfl_xor(sq, 1, 64); sq += __shfl_xor(sq, 2, 64); sq += __shfl_xor(sq, 4, 64);
    const float rstd = 1.0f / sqrtf(sq * (1.0f / CO) + p.eps);
    if (!live) continue;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      f32x4 o;
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(&sG[32 * j + 4 * l8]);
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(&sB[32 * j + 4 * l8]);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (acc[j][e] - mean) * rstd * g4[e] + b4[e];
      if (p.out) *reinterpret_cast<f32x4*>(p.out + orow * CO + 32 * j + 4 * l8) = o;
      if (p.ohi) rsp_store_planes4(p.ohi, p.olo, ((int64_t)j * p.out_rows + orow) * 32 + 4 * l8, o * p.pscale, false);
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

extern "C" int rsp_sam_i2t_fused(const RspI2tFusedDesc* d, rsp_stream_t stream) {
  if (!d || !d->q || !d->k || !d->v || !d->wo || !d->bo || !d->gamma || !d->beta) return RSP_EINVAL;
  if (!d->out && !(d->out_hi && d->out_lo)) return RSP_EINVAL;
  if ((d->out_hi == nullptr) != (d->out_lo == nullptr)) return RSP_EINVAL;
  if ((d->res != nullptr) == (d->res_hi != nullptr) || (d->res_hi && !d->res_lo)) return RSP_EINVAL;   // exactly one residual form
  if (d->R < 0 || d->T <= 0 || d->T > 10 || d->N <= 0 || d->R > 65535) return RSP_EINVAL;   // T <= 10: the LDS budget
  if (RSP_PLANE_IS_F8(d->out_scale_log2) || RSP_PLANE_IS_F8(d->res_scale_log2)) return RSP_EINVAL;
  if (d->R == 0) return RSP_OK;
  I2tFusedP p;
  p.q = d->q; p.q_map = d->q_map; p.k = d->k; p.v = d->v; p.wo = d->wo; p.bo = d->bo;
  p.res = d->res; p.res_map = d->res_map;
  p.res_hi = reinterpret_cast<const half_t*>(d->res_hi); p.res_lo = reinterpret_cast<const half_t*>(d->res_lo);
  p.res_inv_scale = ldexpf(1.0f, -RSP_PLANE_EXP(d->res_scale_log2)); p.res_rows = (int64_t)d->R * d->N;
  p.gamma = d->gamma; p.beta = d->beta; p.eps = d->eps;
  p.out = d->out; p.ohi = reinterpret_cast<half_t*>(d->out_hi); p.olo = reinterpret_cast<half_t*>(d->out_lo);
  p.pscale = ldexpf(1.0f, RSP_PLANE_EXP(d->out_scale_log2)); p.out_rows = (int64_t)d->R * d->N;
  p.T = d->T; p.N = d->N; p.scale = d->scale;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((d->N + FUSED_PIX_PER_BLOCK - 1) / FUSED_PIX_PER_BLOCK, d->R);
  if (d->T <= 8) hipLaunchKernelGGL((sam_i2t_fused_kernel<8>), grid, dim3(1024), 0, s, p);
  else hipLaunchKernelGGL((sam_i2t_fused_kernel<10>), grid, dim3(1024), 0, s, p);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}
