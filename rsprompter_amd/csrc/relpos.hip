// hipcc-flags: -fno-slp-vectorize
// Decomposed relative-position terms of the SAM ViT attention (HF:761-801; vit_sam.py:202-221):
//   rel[bh, t, 0:S] = q . Rh[qh - kh + S-1],  rel[bh, t, S:2S] = q . Rw[qw - kw + S-1]     (UNSCALED q)
// Own translation unit because it wants the SLP vectorizer off: packing the FMAs of two table rows into
// v_pk_fma_f32 makes the compiler keep {q, q} pairs of the whole query (hundreds of registers, spills).
#include <stdlib.h>
#include "rsp_common.h"

namespace {

// rel[bh, t, 0:S] = q . Rh[qh - kh + S-1], rel[bh, t, S:2S] = q . Rw[qw - kw + S-1]
// plain fp32 FMA dot products (same arithmetic class as the reference einsum).
// Block = 64 queries of one (window, head); thread = (query, quarter of the 2S outputs).  The query lives in
// registers, both tables in LDS (rows padded to DH+4 floats: 16-byte aligned, conflict-free float4 reads), so one
// output costs DH FMAs + DH/4 LDS reads instead of 2*DH LDS reads.
template <int DH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8))) void vit_relpos_kernel(const float* __restrict__ qkv,
                                                         const float* __restrict__ rph,
                                                         const float* __restrict__ rpw,
                                                         float* __restrict__ rel, int T, int S,
                                                         int nh, int64_t rows_total, int64_t tok_stride) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int LD = DH + 4;
  const int nrow = 2 * S - 1;
  float* sH = smem;                 // [nrow][LD]
  float* sW = sH + nrow * LD;       // [nrow][LD]
  const int tid = threadIdx.x;
  const int h = blockIdx.y;
  // 64 consecutive rows of the [Bp*T] token axis: windows of T = 196 tokens leave no ragged last block
  const int64_t g = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  for (int idx = tid; idx < nrow * (DH / 4); idx += 256) {
    const int r = idx / (DH / 4), c = idx - r * (DH / 4);
    *reinterpret_cast<f32x4*>(sH + r * LD + 4 * c) = *reinterpret_cast<const f32x4*>(rph + (int64_t)r * DH + 4 * c);
    *reinterpret_cast<f32x4*>(sW + r * LD + 4 * c) = *reinterpret_cast<const f32x4*>(rpw + (int64_t)r * DH + 4 * c);
  }
  const bool qok = g < rows_total;
  const int bp = qok ? (int)(g / T) : 0;
  const int q = qok ? (int)(g - (int64_t)bp * T) : 0;
  const int part = tid >> 6;                      // wave-uniform: 0,1 -> rel_h halves, 2,3 -> rel_w halves
  f32x4 qv[DH / 4];
  {
    const float* src = qkv + ((int64_t)bp * T + q) * tok_stride + (int64_t)h * DH;
#pragma unroll
    for (int c = 0; c < DH / 4; ++c) qv[c] = *reinterpret_cast<const f32x4*>(src + 4 * c);
  }
  __syncthreads();
  if (!qok) return;
  const int qy = q / S, qx = q - qy * S;
  const int half = (((S + 1) / 2) + 1) & ~1;        // even split point: 8-byte stores stay aligned
  const int jb = (part & 1) * half, je = (part & 1) ? S : half;
  const int tab0 = (part < 2) ? 0 : nrow * LD;      // float index of the table inside smem
  const int pos = (part < 2) ? qy : qx;
  float* dst = rel + (((int64_t)bp * nh + h) * T + q) * (2 * S) + (part < 2 ? 0 : S);
  if ((S & 7) == 0) {     // global layers (S = 64): 16-byte stores, a thread fills whole 128-byte lines
#pragma unroll 1
    for (int j = jb; j < je; j += 4) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ro = tab0 + (pos - (j + e) + S - 1) * LD;
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < DH / 4; ++c) {
          const f32x4 t4 = *reinterpret_cast<const f32x4*>(&smem[ro + 4 * c]);
          acc = fmaf(qv[c][0], t4[0], acc);
          acc = fmaf(qv[c][1], t4[1], acc);
          acc = fmaf(qv[c][2], t4[2], acc);
          acc = fmaf(qv[c][3], t4[3], acc);
        }
        o[e] = acc;
        if (e == 1) __builtin_amdgcn_sched_barrier(0);   // two rows' LDS reads in flight at a time, not four
      }
      *reinterpret_cast<f32x4*>(dst + j) = o;
    }
  } else if ((S & 1) == 0) {   // windows (S = 14): 8-byte stores
#pragma unroll 1
    for (int j = jb; j < je; j += 2) {
      float o[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int ro = tab0 + (pos - (j + e) + S - 1) * LD;
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < DH / 4; ++c) {
          const f32x4 t4 = *reinterpret_cast<const f32x4*>(&smem[ro + 4 * c]);
          acc = fmaf(qv[c][0], t4[0], acc);
          acc = fmaf(qv[c][1], t4[1], acc);
          acc = fmaf(qv[c][2], t4[2], acc);
          acc = fmaf(qv[c][3], t4[3], acc);
        }
        o[e] = acc;
      }
      *reinterpret_cast<float2*>(dst + j) = make_float2(o[0], o[1]);
    }
  } else {
#pragma unroll 1
    for (int j = jb; j < je; ++j) {
      const int ro = tab0 + (pos - j + S - 1) * LD;
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < DH / 4; ++c) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(&smem[ro + 4 * c]);
        acc = fmaf(qv[c][0], t4[0], acc);
        acc = fmaf(qv[c][1], t4[1], acc);
        acc = fmaf(qv[c][2], t4[2], acc);
        acc = fmaf(qv[c][3], t4[3], acc);
      }
      dst[j] = acc;
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Windowed layers (S <= 16, tables of 2S-1 <= 32 rows): the same terms as ONE small matrix product per table,
//   G[idx, q] = R[idx, :] . q        (fp16x3 MFMA, 32 queries x 32 table rows per wave, K = dh)
// followed by the Toeplitz gather rel_h[q, j] = G_h[qy - j + S-1, q], rel_w[q, j] = G_w[qx - j + S-1, q] through a
// per-wave LDS transpose.  28 of the 32 ViT-H layers are windowed; the FMA kernel above is LDS-read bound there.
constexpr int RQ = 6, RT = 6;          // power-of-two operand scales (fp16 range), as in the attention kernels
constexpr int WIN_ITERS = 4;           // groups of 128 queries per block of the window kernel

template <int DH>
__global__ __launch_bounds__(256) void vit_relpos_win_kernel(const float* __restrict__ qkv,
                                                             const float* __restrict__ rph,
                                                             const float* __restrict__ rpw,
                                                             float* __restrict__ rel, int T, int S, int nh,
                                                             int64_t rows_total, int64_t tok_stride,
                                                             const int32_t* __restrict__ rows_map) {
  // rows_map (optional): the rows to process, e.g. the real tokens of padded windows (the rel rows of padded queries
  // are never read once the attention skips them); rows_total then counts the map's entries.
  // A block stages the two tables once and walks WIN_ITERS groups of 128 queries (the staging was 40 % of its work).
  constexpr int DSTEPS = DH / 16;
  constexpr int LDT = DH + 8;                                  // halves per table row (conflict-free b128 reads)
  __shared__ __attribute__((aligned(16))) half_t sT[2][2][32 * LDT];   // [table][hi/lo][idx][d]
  __shared__ float sG[4][32][33];                              // per wave: [query][idx]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, l31 = lane & 31;
  const int h = blockIdx.y;
  const int nrow = 2 * S - 1;
  const float ts = ldexpf(1.0f, RT);
  for (int idx = tid; idx < 2 * 32 * (DH / 4); idx += 256) {
    const int tb = idx / (32 * (DH / 4)), rem = idx - tb * 32 * (DH / 4);
    const int r = rem / (DH / 4), c = rem - r * (DH / 4);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < nrow) v = *reinterpret_cast<const f32x4*>((tb ? rpw : rph) + (int64_t)r * DH + 4 * c);
    half4_t hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) { half_t a, b; rsp_split1(v[e] * ts, a, b); hi[e] = a; lo[e] = b; }
    *reinterpret_cast<half4_t*>(&sT[tb][0][r * LDT + 4 * c]) = hi;
    *reinterpret_cast<half4_t*>(&sT[tb][1][r * LDT + 4 * c]) = lo;
  }
  __syncthreads();
  for (int it = 0; it < WIN_ITERS; ++it) {
  const int64_t g = (((int64_t)blockIdx.x * WIN_ITERS + it) * 4 + wave) * 32 + l31;
  if (((int64_t)blockIdx.x * WIN_ITERS + it) * 128 >= rows_total) break;          // block-uniform
  const bool ok = g < rows_total;
  const int64_t gg = ok ? (rows_map ? (int64_t)rows_map[g] : g) : 0;
  const int q = (int)(gg % T);
  const int qy = q / S, qx = q - qy * S;
  // Q fragments (B operand): lane = query column, k slots 8hh..8hh+7 of every 16
  half8_t qh[DSTEPS], ql[DSTEPS];
  {
    const float qs = ldexpf(1.0f, RQ);
    const float* src = qkv + gg * tok_stride + (int64_t)h * DH;
#pragma unroll
    for (int st = 0; st < DSTEPS; ++st) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(src + st * 16 + hh * 8);
      const f32x4 b = *reinterpret_cast<const f32x4*>(src + st * 16 + hh * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        half_t x, y;
        rsp_split1((ok ? a[e] : 0.f) * qs, x, y); qh[st][e] = x; ql[st][e] = y;
        rsp_split1((ok ? b[e] : 0.f) * qs, x, y); qh[st][4 + e] = x; ql[st][4 + e] = y;
      }
    }
  }
  const float unscale = ldexpf(1.0f, -(RQ + RT));
  const int64_t bpw = gg / T;                                  // window index: rel is [Bp*nh, T, 2S]
  float* dst = rel + ((bpw * nh + h) * T + q) * (2 * S);
  // the half waves split the S outputs of a table into [0, S0) and [S0, S) with S0 even (8-byte stores stay aligned)
  const int S0 = ((S / 2) + 1) & ~1;
  const int jb = hh ? S0 : 0, je = hh ? S : S0;
#pragma unroll
  for (int tb = 0; tb < 2; ++tb) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int st = 0; st < DSTEPS; ++st) {
      const int off = l31 * LDT + st * 16 + hh * 8;
      const half8_t th = *reinterpret_cast<const half8_t*>(&sT[tb][0][off]);
      const half8_t tl = *reinterpret_cast<const half8_t*>(&sT[tb][1][off]);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(tl, qh[st], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(th, ql[st], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(th, qh[st], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) sG[wave][l31][(r & 3) + 8 * (r >> 2) + 4 * hh] = acc[r] * unscale;
    __syncthreads();
    if (ok) {
      const int pos = (tb ? qx : qy) + S - 1;
      float* d2 = dst + (tb ? S : 0);
      if ((S & 1) == 0) {
        for (int j = jb; j < je; j += 2)
          *reinterpret_cast<float2*>(d2 + j) = make_float2(sG[wave][l31][pos - j], sG[wave][l31][pos - j - 1]);
      } else {
        for (int j = jb; j < je; ++j) d2[j] = sG[wave][l31][pos - j];
      }
    }
    __syncthreads();
  }
  }   // it
}

// ---------------------------------------------------------------------------------------------------------------
// Global layers (S = 32 | 64, tables of 2S-1 rows), round 3: the same matrix-core form.  The FMA kernel above spends
// 2S x dh FMAs per query and head (10 GFLOP per ViT-H layer on the VALU: 0.65 ms, 6x the traffic floor of the q rows it
// reads and the rel rows it writes).  A wave's 32 consecutive queries lie in ONE image row (32 | S), so
//   rel_h[q, j] = Rh[qy + S-1 - j] . q   needs exactly the S table rows qy .. qy + S-1, wave-uniform: with the A rows taken
//                                        in reverse order the accumulator IS rel_h^T (no Toeplitz gather),
//   rel_w[q, j] = Rw[qx + S-1 - j] . q   needs rows qx0 .. qx0 + S + 30 for the wave (3 blocks of 32 at S = 64): G_w^T for
//                                        those rows, then the per-query band through the wave's LDS piece.
// Both results leave through the piece in row layout (S/4 lanes x 16 B per rel row and table: whole 256-byte runs).
template <int DH>
__global__ __launch_bounds__(256) void vit_relpos_glob_kernel(const float* __restrict__ qkv, const float* __restrict__ rph,
                                                              const float* __restrict__ rpw, float* __restrict__ rel, int T,
                                                              int S, int nh, int64_t tok_stride) {
  constexpr int DSTEPS = DH / 16;
  constexpr int LDT = DH + 8;                                  // halves per table row (conflict-free b128 reads)
  constexpr int GS = 97;                                       // floats per query row of the piece (odd: conflict free)
  __shared__ __attribute__((aligned(16))) half_t sTh[2][68 * LDT];    // Rh rows qyb .. qyb + S + 2 (hi, lo)
  __shared__ __attribute__((aligned(16))) half_t sTw[2][128 * LDT];   // Rw, all 2S-1 rows (hi, lo), row 127 = 0
  __shared__ float sG[4][32 * GS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, l31 = lane & 31;
  const int h = blockIdx.y;
  const int nrow = 2 * S - 1;
  const int64_t g0 = (int64_t)blockIdx.x * 128;                // first query row of the block (T % 128 == 0)
  const int bp = (int)(g0 / T);
  const int qb = (int)(g0 - (int64_t)bp * T);
  const int qyb = qb / S;                                      // first image row of the block
  const int nh_rows = S + 128 / S - 1;                         // Rh rows the block's waves need
  const float ts = ldexpf(1.0f, RT);
  for (int idx = tid; idx < (nh_rows + 128) * (DH / 4); idx += 256) {
    const int r = idx / (DH / 4), c = idx - r * (DH / 4);
    const bool is_w = r >= nh_rows;
    const int tr = is_w ? r - nh_rows : qyb + r;               // table row
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (tr < nrow) v = *reinterpret_cast<const f32x4*>((is_w ? rpw : rph) + (int64_t)tr * DH + 4 * c);
    half4_t hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) { half_t a, b; rsp_split1(v[e] * ts, a, b); hi[e] = a; lo[e] = b; }
    half_t* d0 = is_w ? &sTw[0][(r - nh_rows) * LDT + 4 * c] : &sTh[0][r * LDT + 4 * c];
    half_t* d1 = is_w ? &sTw[1][(r - nh_rows) * LDT + 4 * c] : &sTh[1][r * LDT + 4 * c];
    *reinterpret_cast<half4_t*>(d0) = hi;
    *reinterpret_cast<half4_t*>(d1) = lo;
  }
  const int q0 = qb + wave * 32;                               // the wave's queries q0 .. q0 + 31: one image row
  const int qy = q0 / S, qx0 = q0 - qy * S;
  const int64_t row = g0 + wave * 32 + l31;
  half8_t qh[DSTEPS], ql[DSTEPS];
  {
    const float qs = ldexpf(1.0f, RQ);
    const float* src = qkv + row * tok_stride + (int64_t)h * DH;
#pragma unroll
    for (int st = 0; st < DSTEPS; ++st) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(src + st * 16 + hh * 8);
      const f32x4 b = *reinterpret_cast<const f32x4*>(src + st * 16 + hh * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        half_t x, y;
        rsp_split1(a[e] * qs, x, y); qh[st][e] = x; ql[st][e] = y;
        rsp_split1(b[e] * qs, x, y); qh[st][4 + e] = x; ql[st][4 + e] = y;
      }
    }
  }
  __syncthreads();
  const float unscale = ldexpf(1.0f, -(RQ + RT));
  float* const piece = sG[wave];
  float* const dst0 = rel + (((int64_t)bp * nh + h) * T + q0) * (2 * S);    // rel row of query q0 (row pitch 2S floats)
  const int lpr = S >> 2;                                      // lanes per rel row and table (16 B each)
  const int rpi = 64 / lpr;                                    // rows per store instruction
  const int lr = lane / lpr, c4 = (lane - lr * lpr) * 4;
  auto product = [&](const half_t* th, const half_t* tl, int row_of_l31, int nblk, int col0) {
    // piece[q][col0 + 32 blk + m] = T[row(m)] . q for the nblk blocks of 32 table rows (row(m) given per lane by the caller)
    for (int blk = 0; blk < nblk; ++blk) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const int trow = row_of_l31 + (col0 < 0 ? -32 * blk : 32 * blk);
#pragma unroll
      for (int st = 0; st < DSTEPS; ++st) {
        const int off = trow * LDT + st * 16 + hh * 8;
        const half8_t a_h = *reinterpret_cast<const half8_t*>(th + off);
        const half8_t a_l = *reinterpret_cast<const half8_t*>(tl + off);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_l, qh[st], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h, ql[st], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h, qh[st], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) piece[l31 * GS + 32 * blk + (r & 3) + 8 * (r >> 2) + 4 * hh] = acc[r] * unscale;
    }
  };
  // ---- rel_h: A row m of block blk = Rh[qy + S-1 - (32 blk + m)] -> piece[q][j] = rel_h[q, j] ----
  product(sTh[0], sTh[1], (qy - qyb) + S - 1 - l31, S / 32, -1);
  RSP_WAVE_LOCKSTEP();                                         // the piece is private to the wave: no block barrier
  for (int g = 0; g < 32; g += rpi) {
    const int r_ = g + lr;
    const float* pr = piece + r_ * GS + c4;
    const f32x4 o = {pr[0], pr[1], pr[2], pr[3]};
    *reinterpret_cast<f32x4*>(dst0 + (int64_t)r_ * (2 * S) + c4) = o;
  }
  RSP_WAVE_LOCKSTEP();
  // ---- rel_w: G_w^T for table rows qx0 .. qx0 + 32 nbw - 1, then rel_w[q, j] = piece[q][(q - q0) + S-1 - j] ----
  product(sTw[0], sTw[1], qx0 + l31, (S + 62) / 32, 0);
  RSP_WAVE_LOCKSTEP();
  for (int g = 0; g < 32; g += rpi) {
    const int r_ = g + lr;
    const float* pr = piece + r_ * GS + r_ + S - 1 - c4;
    const f32x4 o = {pr[0], pr[-1], pr[-2], pr[-3]};
    *reinterpret_cast<f32x4*>(dst0 + (int64_t)r_ * (2 * S) + S + c4) = o;
  }
}

}  // namespace

extern "C" int rsp_vit_relpos_rows(const float* qkv, int64_t q_ld, const float* rel_pos_h, const float* rel_pos_w,
                                   float* rel, int32_t Bp, int32_t S, int32_t nh, int32_t dh, const int32_t* rows_map,
                                   int64_t n_rows, rsp_stream_t stream) {
  const int64_t tok_stride = q_ld;
  if (rows_map && (n_rows < 0 || n_rows > (int64_t)Bp * S * S || S > 16)) return RSP_EINVAL;   // row lists: windowed layers
  if ((q_ld & 3) || q_ld < (int64_t)nh * dh) return RSP_EINVAL;
  if (!qkv || !rel_pos_h || !rel_pos_w || !rel || Bp <= 0 || S <= 0 || S > 64 || nh <= 0)
    return RSP_EINVAL;
  const int T = S * S;
  const int64_t rows_total = rows_map ? n_rows : (int64_t)Bp * T;
  if (rows_total == 0) return RSP_OK;
  if (S <= 16 && (dh == 64 || dh == 80)) {   // windowed layers: MFMA form
    dim3 g2((unsigned)((rows_total + 128 * WIN_ITERS - 1) / (128 * WIN_ITERS)), nh, 1);
    if (dh == 64)
      hipLaunchKernelGGL((vit_relpos_win_kernel<64>), g2, dim3(256), 0, (hipStream_t)stream, qkv, rel_pos_h, rel_pos_w, rel, T, S, nh, rows_total, tok_stride, rows_map);
    else
      hipLaunchKernelGGL((vit_relpos_win_kernel<80>), g2, dim3(256), 0, (hipStream_t)stream, qkv, rel_pos_h, rel_pos_w, rel, T, S, nh, rows_total, tok_stride, rows_map);
    RSP_CHECK_LAUNCH();
    return RSP_OK;
  }
  if (rows_map) return RSP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  // global layers: matrix-core form (round 3); other grid sizes: the fp32 FMA kernel below
  if ((S == 32 || S == 64) && (dh == 64 || dh == 80)) {
    dim3 g3((unsigned)(rows_total / 128), nh, 1);              // T = S^2 is a multiple of 128
    if (dh == 64)
      hipLaunchKernelGGL((vit_relpos_glob_kernel<64>), g3, dim3(256), 0, s, qkv, rel_pos_h, rel_pos_w, rel, T, S, nh, tok_stride);
    else
      hipLaunchKernelGGL((vit_relpos_glob_kernel<80>), g3, dim3(256), 0, s, qkv, rel_pos_h, rel_pos_w, rel, T, S, nh, tok_stride);
    RSP_CHECK_LAUNCH();
    return RSP_OK;
  }
  dim3 grid((unsigned)((rows_total + 63) / 64), nh, 1);
  const size_t smem = (size_t)(2 * (2 * S - 1)) * (dh + 4) * sizeof(float);
  if (dh == 64) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&vit_relpos_kernel<64>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((vit_relpos_kernel<64>), grid, dim3(256), smem, s, qkv, rel_pos_h, rel_pos_w, rel, T, S, nh, rows_total, tok_stride);
  } else if (dh == 80) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&vit_relpos_kernel<80>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((vit_relpos_kernel<80>), grid, dim3(256), smem, s, qkv, rel_pos_h, rel_pos_w, rel, T, S, nh, rows_total, tok_stride);
  } else {
    return RSP_EINVAL;
  }
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}


extern "C" int rsp_vit_relpos_q(const float* qkv, int64_t q_ld, const float* rel_pos_h, const float* rel_pos_w,
                                float* rel, int32_t Bp, int32_t S, int32_t nh, int32_t dh, rsp_stream_t stream) {
  return rsp_vit_relpos_rows(qkv, q_ld, rel_pos_h, rel_pos_w, rel, Bp, S, nh, dh, nullptr, 0, stream);
}

extern "C" int rsp_vit_relpos(const float* qkv, const float* rel_pos_h, const float* rel_pos_w,
                              float* rel, int32_t Bp, int32_t S, int32_t nh, int32_t dh,
                              rsp_stream_t stream) {
  return rsp_vit_relpos_q(qkv, (int64_t)3 * nh * dh, rel_pos_h, rel_pos_w, rel, Bp, S, nh, dh, stream);
}
