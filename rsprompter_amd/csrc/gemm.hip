// fp16x3 split-precision MFMA GEMM with fused prologue (row gather / implicit
// im2col) and epilogue (scale, bias, activation, residual, row scatter).
//
//   C[crow(m), n] = act(alpha * sum_k A(m,k) * W[n,k] + bias[n]) + res[crow(m), n]
//
// Numerics ("fp16x3"): every fp32 operand x is represented as
// hi + lo with hi = f16(x*2^e), lo = f16(x*2^e - hi) (~22 significant bits).
// The product is accumulated in fp32 on the matrix cores as
//   a_lo*b_hi + a_hi*b_lo + a_hi*b_hi      (the 2^-22 lo*lo term is dropped)
// using v_mfma_f32_32x32x16_f16 -- 3 MFMA passes, i.e. an effective peak of
// 2.5 PF / 3 = 833 TFLOP/s versus 157 TFLOP/s for the f32-input MFMA, at
// fp32-class accuracy (needed for the 1e-3 mask-logit parity bound).
//
// Weights arrive pre-split (rsp_split_f16, done once at load); activations are
// split on the fly while the tile is staged global -> registers -> LDS.
//
// Tiling: 256 threads = 4 waves, block tile BM x BN (128x128 / 128x64 /
// 128x32), BK = 32, LDS double-buffered with one barrier per K step; the next
// tile's global loads are issued before the MFMA block and written to LDS
// after it (issue-early / write-late).
#include "rsp_common.h"

namespace {

constexpr int BK = 32;
constexpr int LDS_LD = 40;  // halves per LDS row (32 + 8 pad => 80 B rows, conflict-free b128 reads)

struct GemmP {
  RspGemmDesc d;
};

template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(256) void gemm_f16x3_kernel(const GemmP p) {
  static_assert(WGM * WGN == 4, "4 waves per block");
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int A_PER_THREAD = BM / 32;                  // float4 loads per thread per K tile
  constexpr int B_CHUNKS = BN * 4;                       // 16-byte chunks per plane per K tile
  constexpr int B_PER_THREAD = (B_CHUNKS + 255) / 256;   // per plane

  __shared__ __attribute__((aligned(16))) half_t sA[2][2][BM * LDS_LD];
  __shared__ __attribute__((aligned(16))) half_t sB[2][2][BN * LDS_LD];

  const RspGemmDesc& d = p.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int hh = lane >> 5, l31 = lane & 31;
  const int M = d.M, N = d.N, K = d.K;
  const int nbn = (N + BN - 1) / BN;          // 1-D grid, N-blocks fastest: neighbours share the A panel
  const int m0 = (int)(blockIdx.x / nbn) * BM, n0 = (int)(blockIdx.x % nbn) * BN;
  const float a_scale = ldexpf(1.0f, RSP_PLANE_EXP(d.a_scale_log2));

  // ---- per-thread A row bookkeeping (fixed across K tiles) ----
  const int a_kc = (tid & 7) << 2;
  int64_t a_base[A_PER_THREAD];   // plain: element offset of the row; conv: batch pixel base
  int a_y[A_PER_THREAD], a_x[A_PER_THREAD];
  bool a_ok[A_PER_THREAD];
#pragma unroll
  for (int i = 0; i < A_PER_THREAD; ++i) {
    const int r = (tid >> 3) + 32 * i;
    const int gm = m0 + r;
    a_ok[i] = gm < M;
    a_base[i] = 0; a_y[i] = 0; a_x[i] = 0;
    if (a_ok[i]) {
      if (d.conv_k == 0) {
        int srow = d.a_rowmap ? d.a_rowmap[gm] : gm;
        if (srow < 0) a_ok[i] = false;
        a_base[i] = (int64_t)srow * d.lda;
      } else {
        const int hw = d.conv_Ho * d.conv_Wo;
        const int b = gm / hw;
        const int rem = gm - b * hw;
        const int yo = rem / d.conv_Wo;
        const int xo = rem - yo * d.conv_Wo;
        a_base[i] = (int64_t)b * d.conv_H * d.conv_W;
        a_y[i] = yo * d.conv_stride - d.conv_pad;
        a_x[i] = xo * d.conv_stride - d.conv_pad;
      }
    }
  }
  // ---- per-thread B chunk bookkeeping ----
  int b_row[B_PER_THREAD], b_kc[B_PER_THREAD];
  bool b_ok[B_PER_THREAD], b_in[B_PER_THREAD];
#pragma unroll
  for (int i = 0; i < B_PER_THREAD; ++i) {
    const int idx = tid + 256 * i;
    b_in[i] = idx < B_CHUNKS;
    b_row[i] = idx >> 2;
    b_kc[i] = (idx & 3) << 3;
    b_ok[i] = b_in[i] && (n0 + b_row[i] < N);
  }

  f32x4 a_reg[A_PER_THREAD];
  uint4 bh_reg[B_PER_THREAD], bl_reg[B_PER_THREAD];

  auto load_tile = [&](int k0) {
    int ky = 0, kx = 0, c0 = 0;
    if (d.conv_k != 0) {
      const int tap = k0 / d.conv_C;
      c0 = k0 - tap * d.conv_C;
      ky = tap / d.conv_k;
      kx = tap - ky * d.conv_k;
    }
#pragma unroll
    for (int i = 0; i < A_PER_THREAD; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (a_ok[i]) {
        if (d.conv_k == 0) {
          v = *reinterpret_cast<const f32x4*>(d.A + a_base[i] + k0 + a_kc);
        } else {
          const int y = a_y[i] + ky, x = a_x[i] + kx;
          if (y >= 0 && y < d.conv_H && x >= 0 && x < d.conv_W) {
            v = *reinterpret_cast<const f32x4*>(
                d.A + (a_base[i] + (int64_t)y * d.conv_W + x) * d.conv_C + c0 + a_kc);
          }
        }
      }
      a_reg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < B_PER_THREAD; ++i) {
      uint4 zh = {0u, 0u, 0u, 0u}, zl = {0u, 0u, 0u, 0u};
      if (b_ok[i]) {
        const int64_t off = ((int64_t)(k0 / BK) * (d.b_rows > 0 ? d.b_rows : N) + n0 + b_row[i]) * BK + b_kc[i];   // KB32 weight layout
        zh = *reinterpret_cast<const uint4*>(d.Bhi + off);
        zl = *reinterpret_cast<const uint4*>(d.Blo + off);
      }
      bh_reg[i] = zh; bl_reg[i] = zl;
    }
  };

  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_PER_THREAD; ++i) {
      const int r = (tid >> 3) + 32 * i;
      half4_t hi, lo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        half_t h, l;
        rsp_split1(a_reg[i][j] * a_scale, h, l);
        hi[j] = h; lo[j] = l;
      }
      *reinterpret_cast<half4_t*>(&sA[buf][0][r * LDS_LD + a_kc]) = hi;
      *reinterpret_cast<half4_t*>(&sA[buf][1][r * LDS_LD + a_kc]) = lo;
    }
#pragma unroll
    for (int i = 0; i < B_PER_THREAD; ++i) {
      if (b_in[i]) {
        *reinterpret_cast<uint4*>(&sB[buf][0][b_row[i] * LDS_LD + b_kc[i]]) = bh_reg[i];
        *reinterpret_cast<uint4*>(&sB[buf][1][b_row[i] * LDS_LD + b_kc[i]]) = bl_reg[i];
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = K / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile((kt + 1) * BK);

#pragma unroll
    for (int s = 0; s < 2; ++s) {
      half8_t ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int off = (wm * WTM + i * 32 + l31) * LDS_LD + s * 16 + hh * 8;
        ah[i] = *reinterpret_cast<const half8_t*>(&sA[buf][0][off]);
        al[i] = *reinterpret_cast<const half8_t*>(&sA[buf][1][off]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int off = (wn * WTN + j * 32 + l31) * LDS_LD + s * 16 + hh * 8;
        bh[j] = *reinterpret_cast<const half8_t*>(&sB[buf][0][off]);
        bl[j] = *reinterpret_cast<const half8_t*>(&sB[buf][1][off]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }

    if (kt + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const float alpha = d.alpha;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      if (row >= M) continue;
      int crow = d.c_rowmap ? d.c_rowmap[row] : row;
      if (crow < 0) continue;
      if (d.ct_W > 0) {  // ConvTranspose2d(k=2, s=2): pixel (y*W + x) -> output row ((y*2 + dy)*W + x)
        const int yy = crow / d.ct_W;
        crow = (yy * 2 + d.ct_dy) * d.ct_W + (crow - yy * d.ct_W);
      }
      int64_t rrow = d.res_mod > 0 ? crow % d.res_mod : crow;
      if (d.res_bmap) {
        const int rb = crow / d.res_brows;
        rrow = (int64_t)d.res_bmap[rb] * d.res_brows + (crow - rb * d.res_brows);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * WTN + j * 32 + l31;
        if (col >= N) continue;
        float v = acc[i][j][r] * alpha;
        if (d.bias) v += d.bias[col];
        v = rsp_act(v, d.act);
        if (d.res) v += d.res[rrow * d.ldr + col];
        v = rsp_act_post(v, d.act);
        if (d.C) d.C[(int64_t)crow * d.ldc + col] = v;
        if (d.Chi) {
          half_t h, l;
          rsp_split1(v * ldexpf(1.0f, RSP_PLANE_EXP(d.c_scale_log2)), h, l);
          const int64_t po = ((int64_t)(col >> 5) * d.c_rows + crow) * 32 + (col & 31);   // KB32 layout
          reinterpret_cast<half_t*>(d.Chi)[po] = h;
          reinterpret_cast<half_t*>(d.Clo)[po] = l;
        }
      }
    }
  }
}

__global__ void split_f16_kernel(const float* __restrict__ w, half_t* __restrict__ hi,
                                 half_t* __restrict__ lo, int64_t n, float scale) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    half_t h, l;
    rsp_split1(w[i] * scale, h, l);
    hi[i] = h; lo[i] = l;
  }
}

// row-major [rows, K] fp32 -> KB32 planes [K/32][rows][32]; a thread handles 4 consecutive k
__global__ void split_f16_kb32_kernel(const float* __restrict__ w, half_t* __restrict__ hi,
                                      half_t* __restrict__ lo, int64_t rows, int K, float scale, bool f8) {
  const int k4n = K / 4;
  const int64_t total = rows * k4n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / k4n;
    const int k = (int)(i - r * k4n) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(w + r * K + k);
    const int64_t o = ((int64_t)(k >> 5) * rows + r) * 32 + (k & 31);
    rsp_store_planes4(hi, lo, o, f32x4{v[0] * scale, v[1] * scale, v[2] * scale, v[3] * scale}, f8);
  }
}

template <int BM, int BN, int WGM, int WGN>
int launch_gemm(const RspGemmDesc& d, hipStream_t s) {
  GemmP p; p.d = d;
  const long long nblk = (long long)((d.N + BN - 1) / BN) * ((d.M + BM - 1) / BM);
  if (nblk > 0x7fffffffLL) return RSP_EINVAL;
  dim3 grid((unsigned)nblk);
  hipLaunchKernelGGL((gemm_f16x3_kernel<BM, BN, WGM, WGN>), grid, dim3(256), 0, s, p);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

}  // namespace

extern "C" int rsp_split_f16(const float* w, uint16_t* hi, uint16_t* lo, int64_t n,
                             int scale_log2, rsp_stream_t stream) {
  if (!w || !hi || !lo || n < 0) return RSP_EINVAL;
  if (n == 0) return RSP_OK;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(split_f16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w,
                     reinterpret_cast<half_t*>(hi), reinterpret_cast<half_t*>(lo), n,
                     ldexpf(1.0f, scale_log2));
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

extern "C" int rsp_split_f16_kb32(const float* w, uint16_t* hi, uint16_t* lo, int64_t rows, int32_t K,
                                  int scale_log2, rsp_stream_t stream) {
  if (!w || !hi || !lo || rows < 0 || K <= 0 || (K & 31)) return RSP_EINVAL;
  if (rows == 0) return RSP_OK;
  int64_t blocks = (rows * (K / 4) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(split_f16_kb32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w,
                     reinterpret_cast<half_t*>(hi), reinterpret_cast<half_t*>(lo), rows, K,
                     ldexpf(1.0f, RSP_PLANE_EXP(scale_log2)), RSP_PLANE_IS_F8(scale_log2));
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

int rsp_gemm_dma_dispatch(const RspGemmDesc& d, hipStream_t s);  // gemm_dma.hip

extern "C" int rsp_gemm(const RspGemmDesc* desc, rsp_stream_t stream) {
  if (!desc) return RSP_EINVAL;
  const RspGemmDesc& d = *desc;
  if (!d.Bhi || !d.Blo) return RSP_EINVAL;
  if (!d.A && !(d.Ahi && d.Alo)) return RSP_EINVAL;
  if (!d.C && !(d.Chi && d.Clo) && !d.hd_out) return RSP_EINVAL;
  if ((d.Chi == nullptr) != (d.Clo == nullptr)) return RSP_EINVAL;
  if (d.Chi && (d.c_rows <= 0 || (d.N & 31))) return RSP_EINVAL;
  if ((d.hd_out || d.ln_gamma || d.res_hi || (d.ct_W > 0 && d.ct_dy < 0)) && !(d.Ahi && d.Alo)) return RSP_EINVAL;   // the fused hyper-network epilogue lives in the plane path
  if (d.M < 0 || d.N <= 0 || d.K <= 0 || (d.K % BK) != 0) return RSP_EINVAL;
  // the fp8-corrected product lives in the plane path: with an fp32 A the cat8 plane of W would be read as a lo plane
  if (RSP_PLANE_IS_F8(d.a_scale_log2) && !(d.Ahi && d.Alo)) return RSP_EINVAL;
  // ... and so do the column-range outputs and the cat8 output planes: the fp32-A kernel writes all N columns of C and
  // plain (hi, lo) planes from column 0
  if (!(d.Ahi && d.Alo) && (d.c_ncols != 0 || d.pl_col0 != 0 || (d.Chi && RSP_PLANE_IS_F8(d.c_scale_log2)))) return RSP_EINVAL;
  if (!RSP_PLANE_WORD_VALID(d.a_scale_log2) || (d.Chi && !RSP_PLANE_WORD_VALID(d.c_scale_log2)) ||
      (d.res_hi && !RSP_PLANE_WORD_VALID(d.res_scale_log2)))
    return RSP_EINVAL;
  if (d.M == 0) return RSP_OK;
  if (d.conv_k != 0) {
    if (d.conv_C % BK != 0) return RSP_EINVAL;
    if (d.K != d.conv_k * d.conv_k * d.conv_C) return RSP_EINVAL;
    if (d.a_rowmap) return RSP_EINVAL;
  } else {
    if ((d.lda & 3) != 0) return RSP_EINVAL;
  }
  if (d.M == 0) return RSP_OK;
  if (d.res && d.ldr <= 0) return RSP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (d.Ahi && d.Alo) return rsp_gemm_dma_dispatch(d, s);
  if (d.N > 64) return launch_gemm<128, 128, 2, 2>(d, s);
  if (d.N > 32) return launch_gemm<128, 64, 2, 2>(d, s);
  return launch_gemm<128, 32, 4, 1>(d, s);
}
