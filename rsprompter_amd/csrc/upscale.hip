// Last stage of the SAM mask decoder's upscaler (HF:521-531), streaming form:
//   masks[r, 2y+dy, 2x+dx] = sum_c GELU( ConvTranspose2d(64 -> 32, k2, s2)(u)[r, c, 2y+dy, 2x+dx] ) * hyper_in[r, c]
// As a GEMM this is M = R*(2h)(2w) rows (13 M at R = 800), K = 64, N = 128 = (dy, dx, c): two K tiles only, so the
// tiled GEMM kernel spends its time in prologues and epilogues (1 TB/s).  Here the whole weight (32 KB as fp16 hi/lo)
// stays in LDS, waves stream 32-row tiles of the fp16-plane input straight from HBM into MFMA B operands (no LDS, no
// barrier in the loop, next tile's loads in flight during the current tile's math), and the transposed accumulator
// (lane = pixel, registers = channels) makes GELU + the hyper-network dot an in-lane reduction.  The [R, 4h, 4w, 32]
// tensor never exists.  Same fp16x3 arithmetic as gemm_dma.hip.
#include "rsp_common.h"

namespace {

struct Up2P {
  const half_t* Ahi; const half_t* Alo;     // KB32 planes [2][rows][32]
  const half_t* Whi; const half_t* Wlo;     // packed weight planes [2][128][32], rows (dy, dx, c)
  const float* bias;                         // [128] (bias tiled over the 4 sub-pixels)
  const float* hyper;                        // [R, 32]
  float* out;                                // [R, 2*(rows_per/ctW), 2*ctW]
  int64_t rows;
  int rows_per, ctW;                         // rows per RoI (= H2*W2), W2
  float alpha;                               // 2^-(a_scale + w_scale)
  int ntiles;
};

__global__ __launch_bounds__(256) void sam_upscale2_kernel(const Up2P p) {
  // weight image in LDS: [plane][kb][128 rows][64 B], 16-byte chunks XOR-swizzled with (row >> 2) & 3 (the fragment
  // reads of 32 consecutive rows would otherwise be 4-way bank conflicts, as in gemm_dma.hip)
  __shared__ __attribute__((aligned(16))) unsigned char sW[2 * 2 * 128 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, l31 = lane & 31;
  for (int u = tid; u < 2 * 2 * 128 * 4; u += 256) {             // 16-byte units
    const int c = u & 3, row = (u >> 2) & 127, kb = (u >> 9) & 1, pl = u >> 10;
    const half_t* src = (pl == 0 ? p.Whi : p.Wlo) + ((int64_t)(kb * 128 + row) * 32 + c * 8);
    *reinterpret_cast<uint4*>(sW + (((pl * 2 + kb) * 128 + row) * 4 + (c ^ ((row >> 2) & 3))) * 16) =
        *reinterpret_cast<const uint4*>(src);
  }
  __syncthreads();

  const int wstride = gridDim.x * 4;
  int tile = blockIdx.x * 4 + wave;
  half8_t ah[4], al[4];                                           // B operands of the current tile (k16 step s)
  auto load_a = [&](int t, half8_t (&h8)[4], half8_t (&l8)[4]) {
    int64_t row = (int64_t)t * 32 + l31;
    if (row >= p.rows) row = p.rows - 1;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int64_t off = ((int64_t)(s >> 1) * p.rows + row) * 32 + ((s & 1) * 2 + hh) * 8;
      h8[s] = *reinterpret_cast<const half8_t*>(p.Ahi + off);
      l8[s] = *reinterpret_cast<const half8_t*>(p.Alo + off);
    }
  };
  if (tile < p.ntiles) load_a(tile, ah, al);
  for (; tile < p.ntiles; tile += wstride) {
    // keep the weight fragments in LDS (re-read per tile): hoisted into registers they would cost 128 VGPRs and
    // leave one wave per SIMD, and this kernel lives on memory-level parallelism
    asm volatile("" ::: "memory");
    half8_t nh[4], nl[4];
    const int nxt = tile + wstride;
    if (nxt < p.ntiles) load_a(nxt, nh, nl);                      // in flight during this tile's math
    const int64_t row = (int64_t)tile * 32 + l31;
    const bool rok = row < p.rows;
    const int roi = rok ? (int)(row / p.rows_per) : 0;
    const int pix = rok ? (int)(row - (int64_t)roi * p.rows_per) : 0;
    const int y = pix / p.ctW, x = pix - y * p.ctW;
    f32x4 hy[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) hy[g] = *reinterpret_cast<const f32x4*>(p.hyper + (int64_t)roi * 32 + 8 * g + 4 * hh);
    float res[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {                                  // sub-pixel (dy, dx) = (j >> 1, j & 1)
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const int wrow = j * 32 + l31;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int off = (((s >> 1) * 128 + wrow) * 4 + ((((s & 1) * 2 + hh)) ^ ((wrow >> 2) & 3))) * 16;
        const half8_t wh = *reinterpret_cast<const half8_t*>(sW + off);
        const half8_t wl = *reinterpret_cast<const half8_t*>(sW + 2 * 128 * 64 + off);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, ah[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, al[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, ah[s], acc, 0, 0, 0);
      }
      float sum = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + j * 32 + 8 * g + 4 * hh);
#pragma unroll
        for (int e = 0; e < 4; ++e) sum += rsp_gelu(acc[4 * g + e] * p.alpha + b4[e]) * hy[g][e];
      }
      sum += __shfl_xor(sum, 32, 64);
      res[j] = sum;
    }
    if (rok) {
      // the half waves split the four stores: hh == 0 writes the dy = 0 row pair, hh == 1 the dy = 1 one
      float* o = p.out + (int64_t)roi * (4 * (int64_t)p.rows_per) + (int64_t)(2 * y + hh) * (2 * p.ctW) + 2 * x;
      *reinterpret_cast<float2*>(o) = make_float2(hh ? res[2] : res[0], hh ? res[3] : res[1]);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) { ah[s] = nh[s]; al[s] = nl[s]; }
  }
}

}  // namespace

extern "C" int rsp_sam_upscale2(const uint16_t* a_hi, const uint16_t* a_lo, int64_t rows, int32_t a_scale_log2,
                                const uint16_t* w_hi, const uint16_t* w_lo, int32_t w_scale_log2, const float* bias,
                                const float* hyper, float* out, int32_t rows_per_roi, int32_t ct_W,
                                rsp_stream_t stream) {
  if (!a_hi || !a_lo || !w_hi || !w_lo || !bias || !hyper || !out || rows <= 0 || rows_per_roi <= 0 || ct_W <= 0 ||
      (rows_per_roi % ct_W) != 0 || (rows % rows_per_roi) != 0)
    return RSP_EINVAL;
  Up2P p;
  p.Ahi = reinterpret_cast<const half_t*>(a_hi); p.Alo = reinterpret_cast<const half_t*>(a_lo);
  p.Whi = reinterpret_cast<const half_t*>(w_hi); p.Wlo = reinterpret_cast<const half_t*>(w_lo);
  p.bias = bias; p.hyper = hyper; p.out = out; p.rows = rows; p.rows_per = rows_per_roi; p.ctW = ct_W;
  p.alpha = ldexpf(1.0f, -(a_scale_log2 + w_scale_log2));
  const int64_t nt = (rows + 31) / 32;
  if (nt > 0x7fffffffLL) return RSP_EINVAL;
  p.ntiles = (int)nt;
  int64_t blocks = (nt + 3) / 4;
  if (blocks > 256 * 8) blocks = 256 * 8;          // persistent: up to 8 blocks (32 waves) per CU
  hipLaunchKernelGGL(sam_upscale2_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}
