// hipcc-flags: -ffp-contract=fast -fno-slp-vectorize
// (round 6, profiles/r6_slp_ab/: without hipcc's SLP packing of independent fp32 operations the image -> token block spills less and
// runs 4.28 instead of 4.92-5.00 ms per ViT-H step, sam_t2i_kernel 0.67 instead of 0.77; every other file of the library is faster or
// equal WITH it -- upscale.hip 3.46 vs 3.98, t2i_fold.hip 4.43 vs 4.60 -- so the flag is per file, tools/r6_slp_ab.sh)
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
  // the rows of the NEXT key are requested before the current key's 10 x 26 VALU operations: with two waves per SIMD
  // and four loads per ~2 us of latency the kernel ran at 2.9 TB/s of its K | V stream (round 3); the wave's last
  // request reads the row of its last key again (clamped: no out-of-range access)
  int key = lane >> 1;
  f32x4 k0 = {0.f, 0.f, 0.f, 0.f}, k1 = k0, v0 = k0, v1 = k0;
  if (key < N) {
    const float* kr = kbase + (int64_t)key * (2 * W);
    k0 = *reinterpret_cast<const f32x4*>(kr); k1 = *reinterpret_cast<const f32x4*>(kr + 4);
    v0 = *reinterpret_cast<const f32x4*>(kr + W); v1 = *reinterpret_cast<const f32x4*>(kr + W + 4);
  }
  for (; key < N; key += 32) {
    // the loop-invariant q values fit in registers next to the accumulators up to 10 tokens; beyond that they are
    // re-read from LDS (wave-uniform addresses) every key instead of being hoisted into spills
    if constexpr (TMAX > 10) asm volatile("" ::: "memory");
    const float* kn = kbase + (int64_t)min(key + 32, N - 1) * (2 * W);
    const f32x4 nk0 = *reinterpret_cast<const f32x4*>(kn), nk1 = *reinterpret_cast<const f32x4*>(kn + 4);
    const f32x4 nv0 = *reinterpret_cast<const f32x4*>(kn + W), nv1 = *reinterpret_cast<const f32x4*>(kn + W + 4);
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
    k0 = nk0; k1 = nk1; v0 = nv0; v1 = nv1;
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
// position on it (measured: 4.2 ms per call at R = 800, VALU / LDS-issue bound, 5x its HBM time).  On the matrix cores
// it is 0.4 TFLOP of fp16x3 per call -- a quarter of a millisecond -- and the kernel becomes the HBM stream it should be.
// Block = 8 waves, one RoI, rounds of 256 positions; the waves change roles inside a round:
//   A  wave w = POSITION GROUP w (32 positions): q rows are read in row layout (16 lanes x 16 B per row) and turned to
//      the position-per-lane layout through the wave's LDS piece; lane (p, hh) owns the scores of tokens 2 tt + hh for
//      all 8 heads, so the softmax needs one exchange with lane p + 32, and the probabilities of token 2 s + hh, heads
//      0..7 ARE the B fragment of k step s (k = 16 s + 8 hh + h).  They go to LDS as fp16 (hi, lo) fragments, 16 bytes
//      per lane -- over the same piece that turned the q rows.
//   B  wave w = CHANNEL BLOCK w (32 of the 256 output channels): its Vp^T fragments (the A operand: 5 k steps x (hi, lo)
//      = 40 registers) are built ONCE per block, straight into registers -- lane (c, hh) needs Vp[t = 2 s + hh][h][c] =
//      v[t, head h] . Wo[c, head h], i.e. its own row of Wo and the RoI's T value rows; no Vp table in LDS.  Every wave
//      multiplies ALL 8 position groups: O^T[g] = Vp^T P[g]^T, 8 x 5 x 3 MFMAs, B fragments by ds_read_b128.
//      (The first version gave each wave all 256 channels of its own positions: 128 accumulators + 128 row-layout
//      values + fragments streamed from LDS, ~250 spilled registers, slower than the VALU form.)
//   C  in the accumulator layout (lane = position, 16 of the wave's 32 channels per half wave): + out_proj bias +
//      residual, LayerNorm statistics as partial sums over the wave's 32 channels, combined across the 8 waves through
//      LDS (two passes like rsp_layernorm: mean, then centred squares), normalise, fp16 planes out: 8 bytes per lane,
//      the 4 stores of a lane complete its 64-byte plane row.
constexpr int F2_THREADS = 512, F2_RPOS = 256, F2_POS = 1024;   // threads, positions per round, positions per block
typedef float f32x16_t __attribute__((ext_vector_type(16)));

// NKS: K steps of 16 = ceil(8 T / 16): T <= 8 -> 4, T <= 10 -> 5.  RES_F32: the residual is an fp32 [rows, 256] tensor
// (layer 0: the per-image source rows), else fp16 planes (layer 1: the previous layer's output).  Planes out only.
template <int NKS, bool RES_F32>
__global__ __launch_bounds__(F2_THREADS) void sam_i2t_fused_mfma_kernel(const I2tFusedP p) {
  constexpr int TT = 2 * NKS;                               // token slots (T <= TT)
  // a wave's LDS piece: first the 32 x 64-float transposition rows (padded to 272 bytes: conflict free for the 16-byte
  // row writes and the position-per-lane reads), then the P fragments of its position group ([k step][plane][lane] x 16 B)
  constexpr int TBR = 272, TBW = 32 * TBR, PGB = NKS * 2 * 1024;
  constexpr int REG = PGB > TBW ? PGB : TBW;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[8 * REG + (2 * TT * W + 3 * CO + 2 * 8 * 8 * 32 + 16) * 4];
  float* const sK = reinterpret_cast<float*>(smem + 8 * REG);
  float* const sV = sK + TT * W;
  float* const sBo = sV + TT * W;
  float* const sG = sBo + CO;
  float* const sB = sG + CO;
  float* const sSum = sB + CO;                              // [group][wave][position]
  float* const sSq = sSum + 8 * 8 * 32;
  float* const sRed = sSq + 8 * 8 * 32;
  const int r = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, l31 = lane & 31;
  const int T = p.T, N = p.N;
  if (tid < CO) { sBo[tid] = p.bo[tid]; sG[tid] = p.gamma[tid]; sB[tid] = p.beta[tid]; }
  for (int i = tid; i < TT * W; i += F2_THREADS) {
    const bool ok = i < T * W;
    sK[i] = ok ? p.k[(int64_t)r * T * W + i] * (p.scale * LOG2E) : 0.f;
    sV[i] = ok ? p.v[(int64_t)r * T * W + i] : 0.f;
  }
  __syncthreads();
  // ---- A fragments: Vp[t][h][c] = sum_{d in head h} v[t][d] Wo[c][d] for c = 32 wave + l31, t = 2 s + hh ----
  half8_t ah[NKS], al[NKS];
  float unscale;
  {
    float vp[NKS][8];
    const float* wr = p.wo + (int64_t)(wave * 32 + l31) * W;
    float vmax = 0.f;
#pragma unroll
    for (int h = 0; h < 8; ++h) {
      f32x4 w4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) w4[u] = *reinterpret_cast<const f32x4*>(wr + h * DH + 4 * u);
#pragma unroll
      for (int s_ = 0; s_ < NKS; ++s_) {
        const float* vr = sV + (2 * s_ + hh) * W + h * DH;
        float a = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const f32x4 v4 = *reinterpret_cast<const f32x4*>(vr + 4 * u);
          a += w4[u][0] * v4[0] + w4[u][1] * v4[1] + w4[u][2] * v4[2] + w4[u][3] * v4[3];
        }
        vp[s_][h] = a;
        vmax = fmaxf(vmax, fabsf(a));
      }
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
    unscale = ldexpf(1.0f, -(14 + ve));
#pragma unroll
    for (int s_ = 0; s_ < NKS; ++s_)
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        half_t a, b;
        rsp_split1(vp[s_][h] * vs, a, b);
        ah[s_][h] = a; al[s_][h] = b;
      }
  }
  const int64_t qb = p.q_map ? p.q_map[r] : r;
  const int64_t rb = p.res_map ? p.res_map[r] : r;
  unsigned char* const piece = smem + wave * REG;           // this wave's piece (phase A)
  const int p_blk = blockIdx.x * F2_POS;
  const int lr0 = lane >> 4, c4 = (lane & 15) * 4;          // row layout: row g * 4 + lr0, columns c4 .. c4 + 3
  const int ch0 = wave * 32 + 4 * hh;                       // phase C: this lane's channels are ch0 + 8 a + e

  for (int rd = 0; rd < F2_POS / F2_RPOS; ++rd) {
    const int p_round = p_blk + rd * F2_RPOS;
    if (p_round >= N) break;                                // block-uniform
    asm volatile("" ::: "memory");                          // K stays in LDS: no hoisting of its reads out of the round loop
    // ================= phase A: scores + softmax of position group `wave`, P fragments -> LDS =================
    {
      const int p_base = p_round + wave * 32;
      unsigned char* const tb_row = piece + lr0 * TBR + c4 * 4;     // row layout: + g * 4 * TBR
      unsigned char* const tb_pos = piece + l31 * TBR;              // position-per-lane layout: + unit * 16
      float sc[NKS][8];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {                       // q rows, coalesced: 4 rows x 256 B per instruction
          const int pos = min(p_base + g * 4 + lr0, N - 1);     // rows past N: a valid row, results never stored
          const f32x4 x = *reinterpret_cast<const f32x4*>(p.q + (qb * N + pos) * (int64_t)W + half * 64 + c4);
          *reinterpret_cast<f32x4*>(tb_row + g * 4 * TBR) = x;
        }
        RSP_WAVE_LOCKSTEP();                                // row layout in, position-per-lane layout out: one wave's piece
#pragma unroll
        for (int hl = 0; hl < 4; ++hl) {
          const int h = half * 4 + hl;
          f32x4 qv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) qv[u] = *reinterpret_cast<const f32x4*>(tb_pos + (hl * 4 + u) * 16);
#pragma unroll
          for (int tt = 0; tt < NKS; ++tt) {
            const float* kr = sK + (2 * tt + hh) * W + h * DH;
            // 16 products as 8 packed fmas on an (even, odd) pair of partial sums (v_pk_fma_f32) + one add
            f32x2 a2 = {0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const f32x4 kk = *reinterpret_cast<const f32x4*>(kr + 4 * u);
              a2 = __builtin_elementwise_fma(f32x2{qv[u][0], qv[u][1]}, f32x2{kk[0], kk[1]}, a2);
              a2 = __builtin_elementwise_fma(f32x2{qv[u][2], qv[u][3]}, f32x2{kk[2], kk[3]}, a2);
            }
            sc[tt][h] = (2 * tt + hh < T) ? a2[0] + a2[1] : -INFINITY;
          }
          __builtin_amdgcn_sched_barrier(0);                // one head's K reads in flight at a time (register pressure)
        }
        RSP_WAVE_LOCKSTEP();                                // (the piece is overwritten: second half, then the P fragments)
      }
      // softmax over the 2 NKS token slots of every head: NKS here, NKS in lane + 32; probabilities * 2^14, split
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
        const float inv = 16384.0f / sm;
#pragma unroll
        for (int tt = 0; tt < NKS; ++tt) sc[tt][h] *= inv;
      }
#pragma unroll
      for (int tt = 0; tt < NKS; ++tt) {
        half8_t ph, pl;                                   // probabilities * 2^14 in [0, 16384]: the truncating split
        rsp_split8_trunc(sc[tt], ph, pl);
        *reinterpret_cast<half8_t*>(piece + (tt * 2 + 0) * 1024 + lane * 16) = ph;
        *reinterpret_cast<half8_t*>(piece + (tt * 2 + 1) * 1024 + lane * 16) = pl;
      }
    }
    __syncthreads();
    // Phases B and C run twice per round, over four position groups each (round 6): with all eight groups' accumulators (128
    // registers) beside the 40 fragment registers and the residual prefetch the kernel spilled 70 dwords, the Vp fragments
    // among them -- reloaded from scratch in front of the MFMAs of every round.  Two more block barriers per round buy that back.
#pragma unroll 1
    for (int g0 = 0; g0 < 8; g0 += 4) {
    // ================= phase B: O^T[g] = Vp^T (channel block `wave`) x P[g]^T for four position groups =================
    f32x16_t acc[4];
#pragma unroll
    for (int gl = 0; gl < 4; ++gl) {
      const int g = g0 + gl;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[gl][e] = 0.f;
#pragma unroll
      for (int s_ = 0; s_ < NKS; ++s_) {
        const half8_t bh = *reinterpret_cast<const half8_t*>(smem + g * REG + (s_ * 2 + 0) * 1024 + lane * 16);
        const half8_t bl = *reinterpret_cast<const half8_t*>(smem + g * REG + (s_ * 2 + 1) * 1024 + lane * 16);
        acc[gl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s_], bh, acc[gl], 0, 0, 0);
        acc[gl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s_], bl, acc[gl], 0, 0, 0);
        acc[gl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s_], bh, acc[gl], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);                    // one group's fragments in registers at a time
    }
    // ================= phase C: bias + residual, LayerNorm over the 256 channels (8 waves), planes out =================
    // residual of group g + 1 requested before group g is processed (straight-line code: no branch, counted waits)
    f32x4 rf[2][4];                                         // (two groups ahead -- a ring of three -- measured: 4.14 vs 4.06 ms, not kept)
    half4_t rh[2][4], rl[2][4];
    auto res_load = [&](int g, int b) {
      const int pos = min(p_round + g * 32 + l31, N - 1);
      if constexpr (RES_F32) {
        const float* rp = p.res + (rb * N + pos) * (int64_t)CO + ch0;            // + 8 a: immediate offsets
#pragma unroll
        for (int a = 0; a < 4; ++a) rf[b][a] = *reinterpret_cast<const f32x4*>(rp + 8 * a);
      } else {
        const int64_t ro = ((int64_t)wave * p.res_rows + (int64_t)r * N + pos) * 32 + 4 * hh;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          rh[b][a] = *reinterpret_cast<const half4_t*>(p.res_hi + ro + 8 * a);
          rl[b][a] = *reinterpret_cast<const half4_t*>(p.res_lo + ro + 8 * a);
        }
      }
    };
    res_load(g0, 0);
#pragma unroll
    for (int gl = 0; gl < 4; ++gl) {
      const int g = g0 + gl;
      if (gl + 1 < 4) res_load(g + 1, (gl + 1) & 1);
      float sm = 0.f;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const f32x4 bo4 = *reinterpret_cast<const f32x4*>(sBo + ch0 + 8 * a);
        f32x4 rsd;
        if constexpr (RES_F32) {
          rsd = rf[gl & 1][a];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) rsd[e] = ((float)rh[gl & 1][a][e] + (float)rl[gl & 1][a][e]) * p.res_inv_scale;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float y = acc[gl][4 * a + e] * unscale + bo4[e] + rsd[e];
          acc[gl][4 * a + e] = y;
          sm += y;
        }
      }
      sm += __shfl_xor(sm, 32, 64);
      if (hh == 0) sSum[(g * 8 + wave) * 32 + l31] = sm;
      __builtin_amdgcn_sched_barrier(0);                    // (the residual loads of all groups at once would spill)
    }
    __syncthreads();
    float mean[4];
#pragma unroll
    for (int gl = 0; gl < 4; ++gl) {
      const int g = g0 + gl;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += sSum[(g * 8 + w) * 32 + l31];
      mean[gl] = t * (1.0f / CO);
      float sq = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) { const float dl = acc[gl][e] - mean[gl]; sq += dl * dl; }
      sq += __shfl_xor(sq, 32, 64);
      if (hh == 0) sSq[(g * 8 + wave) * 32 + l31] = sq;
    }
    __syncthreads();
#pragma unroll
    for (int gl = 0; gl < 4; ++gl) {
      const int g = g0 + gl;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += sSq[(g * 8 + w) * 32 + l31];
      const float rstd = 1.0f / sqrtf(t * (1.0f / CO) + p.eps);
      const int pos = p_round + g * 32 + l31;
      if (pos < N) {
        const int64_t orow = (int64_t)r * N + pos;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int ch = ch0 + 8 * a;
          const f32x4 g4 = *reinterpret_cast<const f32x4*>(sG + ch);
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(sB + ch);
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (acc[gl][4 * a + e] - mean[gl]) * rstd * g4[e] + b4[e];
          rsp_store_planes4(p.ohi, p.olo, ((int64_t)wave * p.out_rows + orow) * 32 + (ch & 31), o * p.pscale, false);
        }
      }
    }
    }   // g0: the two halves of the round
    // (the next round's phase A writes the pieces, phase C the statistics: both behind this round's barriers)
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
  // The matrix-core form is the product path (2.7 / 2.4 ms against 3.8 / 4.0 ms of the VALU form at R = 800, layer-0 /
  // layer-1 arguments, profiles/r3_i2t_mfma_vs_valu.txt); it writes planes only, so a request for the fp32 copy of the
  // result (tests, tools/i2t_micro.py) is served by the VALU form.
  const bool mfma_form = !d->out;
  if (mfma_form) {
    dim3 grid2((d->N + F2_POS - 1) / F2_POS, d->R);
    const bool f32res = d->res != nullptr;
    if (d->T <= 8) {
      if (f32res) hipLaunchKernelGGL((sam_i2t_fused_mfma_kernel<4, true>), grid2, dim3(F2_THREADS), 0, s, p);
      else hipLaunchKernelGGL((sam_i2t_fused_mfma_kernel<4, false>), grid2, dim3(F2_THREADS), 0, s, p);
    } else {
      if (f32res) hipLaunchKernelGGL((sam_i2t_fused_mfma_kernel<5, true>), grid2, dim3(F2_THREADS), 0, s, p);
      else hipLaunchKernelGGL((sam_i2t_fused_mfma_kernel<5, false>), grid2, dim3(F2_THREADS), 0, s, p);
    }
    RSP_CHECK_LAUNCH();
    return RSP_OK;
  }
  dim3 grid((d->N + FUSED_PIX_PER_BLOCK - 1) / FUSED_PIX_PER_BLOCK, d->R);
  if (d->T <= 8) hipLaunchKernelGGL((sam_i2t_fused_kernel<8>), grid, dim3(1024), 0, s, p);
  else hipLaunchKernelGGL((sam_i2t_fused_kernel<10>), grid, dim3(1024), 0, s, p);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}
