// SAM two-way-transformer cross attentions (HF:243-288 SamAttention inside HF:306-348, 396-404), the two shapes
// that matter in the RSPrompter decoders:
//   token -> image : T <= 12 prompt tokens attend over the N = h*w image positions   (many keys, few queries)
//   image -> token : the N image positions attend over the T prompt tokens          (many queries, few keys)
// with internal width 128 = 8 heads x 16.  Both are HBM-bound streams of the per-RoI image tensors (4 MB per RoI
// and call); the MFMA flash kernel in attn.hip spends its time on 64-key tiles and barriers here, so these two are
// plain fp32 VALU kernels: exact fp32 products (no fp16 split at all), one online-softmax state per lane, merged
// across lanes once at the end.
#include <stdlib.h>
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


// ---------------------------------------------------------------------------------------------------------------
// Round 3: the same fused block with the out_proj-folded product ON THE MATRIX CORES.
//   out[pos, c] = sum_k P[pos, k] * Vp[k, c],  k = (token t, head h) = 8 t + h  (K = 8 T <= 80), c = 256 channels
// is a [positions x 80] x [80 x 256] matrix product; the VALU form above spends 80 x (9 LDS reads + 16 packed FMAs) per
// position on it (measured: 4.2 ms per call at R = 800, VALU / LDS-issue bound, 5x its HBM time).  Here:
//   * block = 8 waves, Vp as fp16 (hi, lo) planes in LDS in MFMA A-fragment order [k step][32-channel block][32][16]
//     (32-byte rows, chunk swizzled with bit 3 of the row like the GEMM ring), scale chosen per block from max |Vp|;
//   * a wave takes 32 consecutive positions; its lanes are (position, token parity): lane (p, hh) owns the scores of
//     tokens 2 tt + hh for all 8 heads, so the softmax needs ONE exchange with lane p + 32 (max and sum), and the
//     probabilities of token 2 s + hh, heads 0..7 ARE the B fragment of k step s (k = 16 s + 8 hh + h): P never moves;
//   * O^T = Vp^T P^T in fp16x3 (v_mfma_f32_32x32x16_f16, 8 channel blocks x K/16 steps x 3), fp32 accumulate;
//   * global memory is only touched in ROW layout (16 lanes x 16 B per position row: q, residual, result): the q rows
//     and the 256-channel results pass through a per-wave 8 KB LDS piece (64 columns at a time, XOR swizzle) on their
//     way to / from the position-per-lane layout the matrix product wants -- same trick as the GEMM epilogue;
//   * LayerNorm statistics in row layout: 16 values per lane and row, 4 DPP-class exchanges over the row's 16 lanes.
constexpr int F2_THREADS = 512, F2_POS = 1024;            // positions of one RoI per block
typedef float f32x16_t __attribute__((ext_vector_type(16)));

template <int NKS>     // K steps of 16 = ceil(8 T / 16): T <= 8 -> 4, T <= 10 -> 5
__global__ __launch_bounds__(F2_THREADS) void sam_i2t_fused_mfma_kernel(const I2tFusedP p) {
  constexpr int TT = 2 * NKS;                               // token slots (T <= TT)
  constexpr int VP_PL = NKS * 8 * 32 * 32;                  // bytes of one Vp plane
  // transposition piece of a wave: 32 rows of 64 floats, rows PADDED to 272 bytes -- conflict free for the 16-byte
  // writes (8 rows per LDS cycle land 4 banks apart) and the position-per-lane reads, one 2-way conflict in 16 for the
  // row reads; every address is base + row * 272 + constant (an XOR swizzle gave dozens of distinct lane-dependent
  // addresses, all hoisted out of the position loop and spilled)
  constexpr int TBR = 272, TBW = 32 * TBR;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * VP_PL + 8 * TBW + TT * W * 4 + 3 * CO * 4 + 64];
  unsigned char* const sVh = smem;
  unsigned char* const sVl = smem + VP_PL;
  unsigned char* const tb_all = smem + 2 * VP_PL;
  float* const sK = reinterpret_cast<float*>(smem + 2 * VP_PL + 8 * TBW);
  float* const sBo = sK + TT * W;
  float* const sG = sBo + CO;
  float* const sB = sG + CO;
  float* const sRed = sB + CO;
  const int r = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = p.T, N = p.N;
  if (tid < CO) { sBo[tid] = p.bo[tid]; sG[tid] = p.gamma[tid]; sB[tid] = p.beta[tid]; }
  for (int i = tid; i < TT * W; i += F2_THREADS)
    sK[i] = i < T * W ? p.k[(int64_t)r * T * W + i] * (p.scale * LOG2E) : 0.f;
  // ---- Vp[k][c] = sum_{d in head h} v[t][d] Wo[c][d]: 40 (32) values per thread, block maximum, planes ----
  constexpr int NVP = NKS * 16 * CO / F2_THREADS;
  float vp[NVP];
  float vmax = 0.f;
#pragma unroll
  for (int n = 0; n < NVP; ++n) {
    const int i = tid + n * F2_THREADS;
    const int c = i & (CO - 1), k = i >> 8, h = k & 7, t = k >> 3;
    float a = 0.f;
    if (t < T) {
      const float* wr = p.wo + (int64_t)c * W + h * DH;
      const float* vr = p.v + ((int64_t)r * T + t) * W + h * DH;
#pragma unroll
      for (int d4 = 0; d4 < 4; ++d4) {
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(wr + 4 * d4);
        const f32x4 v4 = *reinterpret_cast<const f32x4*>(vr + 4 * d4);
        a += w4[0] * v4[0] + w4[1] * v4[1] + w4[2] * v4[2] + w4[3] * v4[3];
      }
    }
    vp[n] = a;
    vmax = fmaxf(vmax, fabsf(a));
  }
  vmax = rsp_wave_max(vmax);
  if (lane == 0) sRed[wave] = vmax;
  __syncthreads();
  float bm = sRed[0];
#pragma unroll
  for (int w = 1; w < 8; ++w) bm = fmaxf(bm, sRed[w]);
  // power-of-two scale that puts max |Vp| near 2^13 (hi stays far inside the fp16 range, lo stays normal)
  int ve = 0;
  if (bm > 0.f) { int ex; frexpf(bm, &ex); ve = 13 - ex; }
  ve = max(-24, min(24, ve));
  const float vs = ldexpf(1.0f, ve);
#pragma unroll
  for (int n = 0; n < NVP; ++n) {
    const int i = tid + n * F2_THREADS;
    const int c = i & (CO - 1), k = i >> 8;
    const int s_ = k >> 4, kk = k & 15, j = c >> 5, cc = c & 31;
    const int off = ((s_ * 8 + j) * 32 + cc) * 32 + ((((kk >> 3) ^ ((cc >> 3) & 1))) << 4) + (kk & 7) * 2;
    half_t hi, lo;
    rsp_split1(vp[n] * vs, hi, lo);
    *reinterpret_cast<half_t*>(sVh + off) = hi;
    *reinterpret_cast<half_t*>(sVl + off) = lo;
  }
  __syncthreads();
  const float unscale = ldexpf(1.0f, -(14 + ve));
  const int64_t qb = p.q_map ? p.q_map[r] : r;
  const int64_t rb = p.res_map ? p.res_map[r] : r;
  unsigned char* const tb = tb_all + wave * TBW;            // this wave's transposition piece
  const int p_blk = blockIdx.x * F2_POS;

  for (int grp = wave; grp < F2_POS / 32; grp += 8) {
    const int p_base = p_blk + grp * 32;
    if (p_base >= N) break;                                 // wave-uniform
    asm volatile("" ::: "memory");                          // K / Vp stay in LDS: no hoisting of their reads out of the loop
    // lane-constant addresses are RE-derived in every iteration from an opaque copy of the lane id: hoisted to the
    // kernel entry, the dozens of them are spilled around this loop (cdna_hip_programming.md, persistent-attention note)
    int lane_ = lane;
    asm volatile("" : "+v"(lane_));
    const int hh = lane_ >> 5, l31 = lane_ & 31;
    const int lr0 = lane_ >> 4, c4 = (lane_ & 15) * 4;      // row layout: row g * 4 + lr0, columns c4 .. c4 + 3
    unsigned char* const tb_row = tb + lr0 * TBR + c4 * 4;  // row layout: + g * 4 * TBR
    unsigned char* const tb_pos = tb + l31 * TBR;           // position-per-lane layout: + unit * 16
    const int a_off = l31 * 32 + ((hh ^ ((l31 >> 3) & 1)) << 4);
    // ---- scores: s[tt][h] = q[pos, head h] . K[token 2 tt + hh, head h]  (log2 domain, scale folded into K) ----
    float sc[NKS][8];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int g = 0; g < 8; ++g) {                         // q rows, coalesced: 4 rows x 256 B per instruction
        const int lr = g * 4 + lr0, pos = p_base + lr;
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (pos < N) x = *reinterpret_cast<const f32x4*>(p.q + (qb * N + pos) * (int64_t)W + half * 64 + c4);
        *reinterpret_cast<f32x4*>(tb_row + g * 4 * TBR) = x;
      }
#pragma unroll
      for (int hl = 0; hl < 4; ++hl) {
        const int h = half * 4 + hl;
        f32x4 qv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) qv[u] = *reinterpret_cast<const f32x4*>(tb_pos + (hl * 4 + u) * 16);
#pragma unroll
        for (int tt = 0; tt < NKS; ++tt) {
          const float* kr = sK + (2 * tt + hh) * W + h * DH;
          float a = 0.f;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const f32x4 kk = *reinterpret_cast<const f32x4*>(kr + 4 * u);
            a += qv[u][0] * kk[0] + qv[u][1] * kk[1] + qv[u][2] * kk[2] + qv[u][3] * kk[3];
          }
          sc[tt][h] = (2 * tt + hh < T) ? a : -INFINITY;
        }
        __builtin_amdgcn_sched_barrier(0);                  // one head's K reads in flight at a time (register pressure)
      }
    }
    // ---- softmax over the 2 NKS token slots of every head: 5 here, 5 in lane + 32; probabilities * 2^14, split ----
    half8_t ph[NKS], pl[NKS];
    {
      float inv[8], mx[8];
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        float m = sc[0][h];
#pragma unroll
        for (int tt = 1; tt < NKS; ++tt) m = fmaxf(m, sc[tt][h]);
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sm = 0.f;
#pragma unroll
        for (int tt = 0; tt < NKS; ++tt) { sc[tt][h] = __builtin_amdgcn_exp2f(sc[tt][h] - m); sm += sc[tt][h]; }
        sm += __shfl_xor(sm, 32, 64);
        mx[h] = m;
        inv[h] = 16384.0f / sm;
      }
#pragma unroll
      for (int tt = 0; tt < NKS; ++tt)
#pragma unroll
        for (int h = 0; h < 8; ++h) {
          half_t a, b;
          rsp_split1(sc[tt][h] * inv[h], a, b);
          ph[tt][h] = a; pl[tt][h] = b;
        }
      (void)mx;
    }
    // ---- O^T = Vp^T P^T (8 channel blocks x NKS steps x 3 passes), 64 channels at a time: the two accumulators of
    //      a chunk go through the transposition piece to the row layout x[ch][g] (4 channels 64 ch + c4.. of row
    //      g * 4 + lr0), where the out_proj bias and the residual are added.  One chunk in flight keeps the register
    //      count at x (128) + 2 accumulators (32) + the P fragments (40). ----
    f32x16_t acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
      for (int s_ = 0; s_ < NKS; ++s_) {
        const half8_t ah = *reinterpret_cast<const half8_t*>(sVh + (s_ * 8 + j) * 1024 + a_off);
        const half8_t al = *reinterpret_cast<const half8_t*>(sVl + (s_ * 8 + j) * 1024 + a_off);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, ph[s_], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, pl[s_], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ph[s_], acc[j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    f32x4 x[4][8];
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = {acc[2 * ch + jj][4 * q], acc[2 * ch + jj][4 * q + 1], acc[2 * ch + jj][4 * q + 2], acc[2 * ch + jj][4 * q + 3]};
          *reinterpret_cast<f32x4*>(tb_pos + hh * 16 + (jj * 8 + 2 * q) * 16) = v * unscale;
        }
      const f32x4 bo4 = *reinterpret_cast<const f32x4*>(sBo + 64 * ch + c4);
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const int lr = g * 4 + lr0;
        const int pos = min(p_base + lr, N - 1);
        f32x4 rsd;
        if (p.res) {
          rsd = *reinterpret_cast<const f32x4*>(p.res + (rb * N + pos) * (int64_t)CO + 64 * ch + c4);
        } else {
          const int col = 64 * ch + c4;
          const int64_t ro = ((int64_t)(col >> 5) * p.res_rows + (int64_t)r * N + pos) * 32 + (col & 31);
          const half4_t rh = *reinterpret_cast<const half4_t*>(p.res_hi + ro);
          const half4_t rl = *reinterpret_cast<const half4_t*>(p.res_lo + ro);
#pragma unroll
          for (int e = 0; e < 4; ++e) rsd[e] = ((float)rh[e] + (float)rl[e]) * p.res_inv_scale;
        }
        x[ch][g] = *reinterpret_cast<const f32x4*>(tb_row + g * 4 * TBR) + bo4 + rsd;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- LayerNorm over the 256 channels of a row: 16 values here, the rest in the row's other 15 lanes ----
    asm volatile("" ::: "memory");
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      float sm = 0.f;
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) sm += (x[ch][g][0] + x[ch][g][1]) + (x[ch][g][2] + x[ch][g][3]);
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) sm += __shfl_xor(sm, o, 64);
      const float mean = sm * (1.0f / CO);
      float sq = 0.f;
#pragma unroll
      for (int ch = 0; ch < 4; ++ch)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float dl = x[ch][g][e] - mean; sq += dl * dl; }
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) sq += __shfl_xor(sq, o, 64);
      const float rstd = 1.0f / sqrtf(sq * (1.0f / CO) + p.eps);
      const int pos = p_base + g * 4 + lr0;
      if (pos < N) {
        const int64_t orow = (int64_t)r * N + pos;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
          const f32x4 g4 = *reinterpret_cast<const f32x4*>(sG + 64 * ch + c4);
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(sB + 64 * ch + c4);
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (x[ch][g][e] - mean) * rstd * g4[e] + b4[e];
          const int col = 64 * ch + c4;
          if (p.out) *reinterpret_cast<f32x4*>(p.out + orow * CO + col) = o;
          if (p.ohi) rsp_store_planes4(p.ohi, p.olo, ((int64_t)(col >> 5) * p.out_rows + orow) * 32 + (col & 31), o * p.pscale, false);
        }
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
  // The matrix-core form is correct (same unit tests) but not yet the faster one: 4.9 / 6.8 ms against 3.8 / 4.2 ms of
  // the VALU form at R = 800 (profiles/r3_i2t_mfma_vs_valu.txt) -- hipcc spills ~250 registers around its position
  // loop (800 B of scratch per lane), and with 156 KB of LDS one block per CU hides no latency.  Opt-in until that is
  // fixed: RSP_I2T_MFMA=1.
  const bool mfma_form = getenv("RSP_I2T_MFMA") != nullptr;      // read per call: the tests run both forms
  if (mfma_form) {
    dim3 grid2((d->N + F2_POS - 1) / F2_POS, d->R);
    if (d->T <= 8) hipLaunchKernelGGL((sam_i2t_fused_mfma_kernel<4>), grid2, dim3(F2_THREADS), 0, s, p);
    else hipLaunchKernelGGL((sam_i2t_fused_mfma_kernel<5>), grid2, dim3(F2_THREADS), 0, s, p);
    RSP_CHECK_LAUNCH();
    return RSP_OK;
  }
  dim3 grid((d->N + FUSED_PIX_PER_BLOCK - 1) / FUSED_PIX_PER_BLOCK, d->R);
  if (d->T <= 8) hipLaunchKernelGGL((sam_i2t_fused_kernel<8>), grid, dim3(1024), 0, s, p);
  else hipLaunchKernelGGL((sam_i2t_fused_kernel<10>), grid, dim3(1024), 0, s, p);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}
