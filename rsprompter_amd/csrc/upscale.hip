// hipcc-flags: -fslp-vectorize
// Last stage of the SAM mask decoder's upscaler (HF:521-531), streaming form:
//   masks[r, 2y+dy, 2x+dx] = sum_c GELU( ConvTranspose2d(64 -> 32, k2, s2)(u)[r, c, 2y+dy, 2x+dx] ) * hyper_in[r, c]
// As a GEMM this is M = R*(2h)(2w) rows (13 M at R = 800), K = 64, N = 128 = (dy, dx, c): two K tiles only, so the
// tiled GEMM kernel spends its time in prologues and epilogues (1 TB/s).  Here the whole weight (32 KB as fp16 hi/lo)
// stays in LDS, waves stream 32-row tiles of the fp16-plane input straight from HBM into MFMA B operands (no LDS, no
// barrier in the loop, next tile's loads in flight during the current tile's math), and the transposed accumulator
// (lane = pixel, registers = channels) makes GELU + the hyper-network dot an in-lane reduction.  The [R, 4h, 4w, 32]
// tensor never exists.  Same fp16x3 arithmetic as gemm_dma.hip.
#include <type_traits>
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

// Round 6 ("the box-dependent multimask answer"): this kernel as shipped in rounds 4-5 produced, about once
// in 5 ... 1500 launches depending on the box, 16 wrong outputs -- pixels 16..31 of ONE wave tile, ONE sub-pixel, off by about one
// addend of its 32-channel sum -- but only (a) with several blocks co-resident on a CU (2 waves per SIMD; one block per CU:
// 0 of 6000 launches) and (b) in the builds whose channel sums hipcc SLP-packed ACROSS sub-pixel pairs (v_mov register
// shuffles + v_pk_mul_f32 / v_pk_add_f32 with op_sel, 204-244 VGPRs); builds that finish each sub-pixel's sum with scalar
// adds before the next one starts (156-192 VGPRs): 0 of 7500.  Excluded by experiment: uninitialised HBM / LDS reads (poisoned
// allocations on the GPU, poisoned LDS on the lane emulator), the copy to the host, the cross-half exchange instruction
// (ds_bpermute_b32 and v_permlane32_swap both fail), GELU / transcendentals, the bias loads (through LDS: still fails), the
// MFMA -> accumulator-read distance (32 more wait states: still fails), a write-after-read on MFMA sources
// (tools/probes/mfma_war_probe.hip: 0 of 10^11).  Both conditions are removed here: the sums are pinned scalar per sub-pixel,
// and the grid is one persistent block per CU.  Found afterwards by patching the failing build's assembly
// (tools/probes/up2_isa_bisect.py): every wrong value is the exact sum minus the ONE addend hipcc forms as
//     v_pk_mul_f32 vD, vA, vB op_sel:[0,1]            (low result = A.lo x B.HI)
// in lanes 48..63; with those eight instructions replaced by two v_mul_f32 each the failing build never fails (0 of 3000, and
// 0 of 200 under the s_nop stress that makes every launch of the unpatched build wrong).  The library is therefore built
// with -fno-slp-vectorize and tests/test_isa_guard_cpu.py keeps packed-fp32 instructions with a set op_sel bit out of every
// kernel; this FILE keeps the vectoriser (first line: the fused kernel below gains 0.5 ms per step from it), which the pinned
// sums of this kernel and the pinned dot of the fused one keep from forming that instruction.
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
        f32x4 t4;
#pragma unroll
        for (int e = 0; e < 4; ++e) t4[e] = acc[4 * g + e] * p.alpha + b4[e];
        t4 = rsp_gelu4(t4);
#pragma unroll
        for (int e = 0; e < 4; ++e) sum += t4[e] * hy[g][e];
      }
      // pinned: this sub-pixel's sum is complete (scalar adds) before the next sub-pixel's epilogue starts -- no SLP packing of
      // the four sums across sub-pixels (see the note above the kernel)
      asm volatile("" : "+v"(sum));
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


// ---------------------------------------------------------------------------------------------------------------
// The whole upscaler tail in ONE pass over the per-RoI keys (round 4; verified on the lane-level emulator, not yet on a
// GPU: opt-in):   masks[r, 4y + 2dy + dy2, 4x + 2dx + dx2] =
//     sum_c2 GELU( ConvT2(64 -> 32)( GELU( LN_64( ConvT1(256 -> 64)(keys)[r, :, 2y + dy, 2x + dx] ) ) )[c2, dy2, dx2] ) hyper[r, c2]
// (HF:513-531).  The two-kernel form writes the [R, 2h, 2w, 64] intermediate as fp16 planes (3.4 GB at R = 800) and reads it
// back; here a wave owns 32 input pixels end to end:
//   GEMM 1  acc1[(pos, co)][pixel] (8 blocks of 32 rows) = W1 (256 x 256, streamed through a 2 x 32 KB LDS ring in K chunks
//           of 32 by the DMA engine, shared by the block's waves) x the pixels' plane rows (B operand straight from global
//           memory, 16 bytes per lane, k-step and plane);
//   LN      over the 64 channels of a (pixel, pos): 32 values in the lane + 32 in lane ^ 32; GELU;
//   GEMM 2  per pos: the normalised values ARE the B fragments (accumulator registers [8 (s & 1), +8) of channel block
//           s >> 1 feed k-step s; W2's K columns are packed in that order by the host), A = W2 resident in LDS;
//   GELU, dot with hyper_in in the lane, + lane ^ 32; 16 outputs per pixel leave as four 16-byte rows.
struct UpFP {
  const half_t* Xhi; const half_t* Xlo; int64_t x_rows;     // keys planes KB32 [8][x_rows][32]
  const half_t* W1hi; const half_t* W1lo;                   // [8][256][32], rows (dy, dx, co)
  const half_t* W2hi; const half_t* W2lo;                   // [2][128][32], rows (dy2, dx2, c2), K columns permuted (see above)
  const float* bias1; const float* gamma; const float* beta; const float* bias2;   // [256] (tiled x4), [64], [64], [128]
  const float* hyper;                                       // [R, 32]
  float* out;                                               // [R, 4h, 4w]
  int64_t rows; int rows_per, W;                            // R * h * w, h * w, w
  float alpha1, alpha2, eps;
  int ntiles;                                               // tiles of 128 pixels
};
constexpr int UF_YS = 8;                                    // the normalised activations are split at scale 2^8

__global__ __launch_bounds__(256) void sam_upscale_fused_kernel(const UpFP p) {
  // W2 image as in sam_upscale2_kernel; W1 ring: 2 x [plane][256 rows][64 B], 16-byte chunks XOR-swizzled with (row >> 2) & 3
  __shared__ __attribute__((aligned(16))) unsigned char sW2[2 * 2 * 128 * 64];
  // (two OBJECTS, and a K loop unrolled by two so that every access names its buffer at compile time: with one array
  // indexed by a run-time `buf`, hipcc -- which knows that an LDS-DMA instruction writes LDS -- put an s_waitcnt vmcnt(0)
  // between the DMA issue of chunk c + 1 and the fragment reads of chunk c: rounds 4-5 shipped this kernel without any
  // overlap of the W1 stream and the matrix work, which is what "54 % of the wave cycles at s_waitcnt" in the r5 counters was)
  __shared__ __attribute__((aligned(1024))) unsigned char sW1a[2 * 256 * 64];
  __shared__ __attribute__((aligned(1024))) unsigned char sW1b[2 * 256 * 64];
  typedef const __attribute__((address_space(1))) void* gptr_u;
  typedef __attribute__((address_space(3))) void* lptr_u;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, l31 = lane & 31;
  for (int u = tid; u < 2 * 2 * 128 * 4; u += 256) {
    const int c = u & 3, row = (u >> 2) & 127, kb = (u >> 9) & 1, pl = u >> 10;
    const half_t* src = (pl == 0 ? p.W2hi : p.W2lo) + ((int64_t)(kb * 128 + row) * 32 + c * 8);
    *reinterpret_cast<uint4*>(sW2 + (((pl * 2 + kb) * 128 + row) * 4 + (c ^ ((row >> 2) & 3))) * 16) =
        *reinterpret_cast<const uint4*>(src);
  }
  __syncthreads();                                          // (the W2 image is first read a whole K loop later; made explicit)
  // DMA slots of a W1 chunk: unit u = i * 256 + tid of [plane][row][4 chunks]; the LDS side is lane-linear, so the swizzle
  // is applied to the SOURCE chunk
  int dsrc[8];                                              // byte offset inside the chunk's 16 KB plane block
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int u = (i & 3) * 256 + tid;                      // 1024 units per plane: slots 0-3 hi, 4-7 lo
    const int row = u >> 2, c = (u & 3) ^ ((row >> 2) & 3);
    dsrc[i] = (row * 4 + c) * 16;
  }
  typedef __attribute__((address_space(3))) unsigned char* lds_u8;
  const rsp_lds_addr_t sW1a_a = rsp_lds_addr((lds_u8)sW1a), sW1b_a = rsp_lds_addr((lds_u8)sW1b);
  auto issue_chunk = [&](int kc, rsp_lds_addr_t dst) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned char* base = reinterpret_cast<const unsigned char*>(i < 4 ? p.W1hi : p.W1lo) + (int64_t)kc * (256 * 64);
      RSP_GLOBAL_LOAD_LDS_B128(base + dsrc[i], dst + ((i >> 2) * 1024 + (i & 3) * 256 + wave * 64) * 16);
    }
  };
  // Pixel rows (the B operand, 1 KB per pixel straight from HBM): a ring of THREE register sets, chunk c + 2 requested while
  // chunk c is multiplied -- also across the tile boundary, so the next tile's first two chunks fly during this tile's epilogue
  // (round 6; round 5 requested chunk c + 1 only and started every tile with an exposed HBM round trip: 54 % of the wave
  // cycles at s_waitcnt / s_barrier, PMC r5).  Order of this wave's vector-memory operations: [x(0)] [W1 chunk 0] [x(1)], then
  // per K step [W1 chunk c + 1] [x(c + 2)] -- at the top of step c the four youngest are x(c + 1): vmcnt(4) = chunk c's DMA landed.
  auto tile_row = [&](int t) -> int64_t {
    const int64_t r = (int64_t)t * 128 + wave * 32 + l31;
    return r < p.rows ? r : p.rows - 1;                      // (past the end: a valid row, results never stored)
  };
  // The loads are RSP_GLOBAL_LOAD_B128 (inline assembly on the device): hipcc does not see them as loads, so it neither waits
  // for them on its own nor -- what ruled out plain C++ loads -- drains EVERY outstanding operation at the head of a loop whose
  // back edge they cross.  The s_waitcnt at the top of a K step is therefore the only thing that orders them: a register set
  // is read two steps after its request (behind two such waits, each leaving only the four youngest operations in flight) and
  // is never copied while in flight -- the ring has FOUR sets (8 chunks per tile: chunk c always lives in set c & 3).
  auto load_x = [&](int64_t rowc, int kc, half8_t (&h8)[2], half8_t (&l8)[2]) {
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
      const int64_t o = ((int64_t)kc * p.x_rows + rowc) * 32 + 16 * s_ + 8 * hh;
      RSP_GLOBAL_LOAD_B128(h8[s_], p.Xhi + o);
      RSP_GLOBAL_LOAD_B128(l8[s_], p.Xlo + o);
    }
  };
  half8_t xh[4][2], xl[4][2];
  {
    const int64_t r0 = tile_row(blockIdx.x);
    load_x(r0, 0, xh[0], xl[0]);
    __builtin_amdgcn_sched_barrier(0);
    issue_chunk(0, sW1a_a);
    __builtin_amdgcn_sched_barrier(0);
    load_x(r0, 1, xh[1], xl[1]);
    __builtin_amdgcn_sched_barrier(0);
  }

  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const int64_t row = (int64_t)tile * 128 + wave * 32 + l31;
    const bool rok = row < p.rows;
    const int64_t rowc = rok ? row : p.rows - 1;
    const int64_t rown = tile_row(tile + (int)gridDim.x);   // the next tile's rows (clamped: dummy loads behind the last tile)
    f32x16 acc1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc1[j][e] = 0.f;
    // one K step: chunk kc (pixel rows: register set kc & 3) is read from `rd` while the DMA of chunk kc + 1 fills `wr` and the
    // pixel rows of chunk kc + 2 are requested into set (kc + 2) & 3
    auto k_step = [&](auto kcc, int kc, const unsigned char* rd, rsp_lds_addr_t wr) {
      constexpr int cur = decltype(kcc)::value & 3, nxt = (decltype(kcc)::value + 2) & 3;     // kc & 3 == kcc
      // this wave's part of W1 chunk kc and the pixel rows of chunk kc have landed (the four loads of x(kc + 1) may still fly)
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      __builtin_amdgcn_s_barrier();                         // ... everybody's; `wr` has been read by everybody
      const bool last = (kc == 7) && (tile + (int)gridDim.x >= p.ntiles);
      if (!last) issue_chunk((kc + 1) & 7, wr);
      __builtin_amdgcn_sched_barrier(0);
      load_x(kc < 6 ? rowc : rown, (kc + 2) & 7, xh[nxt], xl[nxt]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int wrow = j * 32 + l31;
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
          const int off = (wrow * 4 + ((2 * s_ + hh) ^ ((wrow >> 2) & 3))) * 16;
          const half8_t wh = *reinterpret_cast<const half8_t*>(rd + off);
          const half8_t wl = *reinterpret_cast<const half8_t*>(rd + 256 * 64 + off);
          acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh[cur][s_], acc1[j], 0, 0, 0);
          acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl[cur][s_], acc1[j], 0, 0, 0);
          acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh[cur][s_], acc1[j], 0, 0, 0);
        }
      }
    };
    for (int half = 0; half < 2; ++half) {                  // 2 x 4 steps: the ring positions repeat, nothing is unrolled further
      k_step(std::integral_constant<int, 0>{}, 4 * half + 0, sW1a, sW1b_a);
      k_step(std::integral_constant<int, 1>{}, 4 * half + 1, sW1b, sW1a_a);
      k_step(std::integral_constant<int, 2>{}, 4 * half + 2, sW1a, sW1b_a);
      k_step(std::integral_constant<int, 3>{}, 4 * half + 3, sW1b, sW1a_a);
    }

    // ---- per sub-pixel (dy, dx): bias, LayerNorm over its 64 channels, GELU, second ConvTranspose, GELU, hyper dot ----
    const int roi = (int)(rowc / p.rows_per);
    const int pix = (int)(rowc - (int64_t)roi * p.rows_per);
    const int y = pix / p.W, x = pix - y * p.W;
    f32x4 hy[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) hy[g] = *reinterpret_cast<const f32x4*>(p.hyper + (int64_t)roi * 32 + 8 * g + 4 * hh);
    float res[4][4];
    const float ys = ldexpf(1.0f, UF_YS);
#pragma unroll
    for (int pos = 0; pos < 4; ++pos) {
      __builtin_amdgcn_sched_barrier(0);
      float v[2][16];
      float sum = 0.f;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias1 + pos * 64 + jj * 32 + 8 * g + 4 * hh);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[jj][4 * g + e] = acc1[2 * pos + jj][4 * g + e] * p.alpha1 + b4[e];
            sum += v[jj][4 * g + e];
          }
        }
      sum += __shfl_xor(sum, 32, 64);
      const float mean = sum * (1.0f / 64.0f);
      float sq = 0.f;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int e = 0; e < 16; ++e) { const float dl = v[jj][e] - mean; sq += dl * dl; }
      sq += __shfl_xor(sq, 32, 64);
      const float rstd = 1.0f / sqrtf(sq * (1.0f / 64.0f) + p.eps);
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int ch = jj * 32 + 8 * g + 4 * hh;
          const f32x4 g4 = *reinterpret_cast<const f32x4*>(p.gamma + ch);
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.beta + ch);
          f32x4 t4;
#pragma unroll
          for (int e = 0; e < 4; ++e) t4[e] = (v[jj][4 * g + e] - mean) * rstd * g4[e] + b4[e];
          t4 = rsp_gelu4(t4) * ys;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[jj][4 * g + e] = t4[e];
        }
      // B fragments of the second product: k-step s = registers [8 (s & 1), +8) of channel block s >> 1
      half8_t yh[4], yl[4];
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) {
#pragma unroll
        for (int t4i = 0; t4i < 2; ++t4i) {
          f32x4 in4, rem;
          half4_t a4, b4;
#pragma unroll
          for (int e = 0; e < 4; ++e) in4[e] = v[s_ >> 1][8 * (s_ & 1) + 4 * t4i + e];
          rsp_split4(in4, a4, b4, rem);
#pragma unroll
          for (int e = 0; e < 4; ++e) { yh[s_][4 * t4i + e] = a4[e]; yl[s_][4 * t4i + e] = b4[e]; }
        }
      }
#pragma unroll
      for (int jb = 0; jb < 4; ++jb) {                        // (dy2, dx2) = (jb >> 1, jb & 1)
        f32x16 acc2;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[e] = 0.f;
        const int wrow = jb * 32 + l31;
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
          const int off = (((s_ >> 1) * 128 + wrow) * 4 + ((((s_ & 1) * 2 + hh)) ^ ((wrow >> 2) & 3))) * 16;
          const half8_t wh = *reinterpret_cast<const half8_t*>(sW2 + off);
          const half8_t wl = *reinterpret_cast<const half8_t*>(sW2 + 2 * 128 * 64 + off);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, yh[s_], acc2, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, yl[s_], acc2, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, yh[s_], acc2, 0, 0, 0);
        }
        float dot = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias2 + jb * 32 + 8 * g + 4 * hh);
          f32x4 t4;
#pragma unroll
          for (int e = 0; e < 4; ++e) t4[e] = acc2[4 * g + e] * p.alpha2 + b4[e];
          t4 = rsp_gelu4(t4);
#pragma unroll
          for (int e = 0; e < 4; ++e) dot += t4[e] * hy[g][e];
        }
        // (as in sam_upscale2_kernel: without this hipcc's SLP vectoriser packs the dots of two sub-pixels and multiplies element 1
        // of every hyper group with  v_pk_mul_f32 ... op_sel:[0,1]  -- the instruction form of DESIGN 9.1; this kernel runs one
        // wave per SIMD, where the form never failed, but the library keeps it out of every kernel: tests/test_isa_guard_cpu.py)
        asm volatile("" : "+v"(dot));
        dot += __shfl_xor(dot, 32, 64);
        res[pos][jb] = dot;
        __builtin_amdgcn_sched_barrier(0);                  // (keeps the next block's loads from piling up registers)
      }
    }
    if (rok) {
      // output rows 4y + 2dy + dy2; the half waves split them: hh == 0 writes dy = 0, hh == 1 dy = 1
      float* o = p.out + (int64_t)roi * (16 * (int64_t)p.rows_per) + (int64_t)(4 * y + 2 * hh) * (4 * p.W) + 4 * x;
#pragma unroll
      for (int dy2 = 0; dy2 < 2; ++dy2) {
        const float a0 = hh ? res[2][2 * dy2] : res[0][2 * dy2], a1 = hh ? res[2][2 * dy2 + 1] : res[0][2 * dy2 + 1];
        const float a2 = hh ? res[3][2 * dy2] : res[1][2 * dy2], a3 = hh ? res[3][2 * dy2 + 1] : res[1][2 * dy2 + 1];
        *reinterpret_cast<f32x4*>(o + (int64_t)dy2 * (4 * p.W)) = f32x4{a0, a1, a2, a3};
      }
    }
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
  if (blocks > 256) blocks = 256;                  // persistent, ONE block per CU (round 6: see the note above the kernel; rounds 4-5: up to 8)
  hipLaunchKernelGGL(sam_upscale2_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

// keys: planes of [rows = R * h * w, 256]; w1: packed [(dy, dx, co), 256] with bias1 tiled x4; LayerNorm2d(64) gamma / beta /
// eps; w2: packed [(dy2, dx2, c2), 64] whose K columns are in the order k' = 16 s + 8 hh + j <- channel
// 32 (s >> 1) + 8 ((8 (s & 1) + j) >> 2) + 4 hh + (j & 3)  (see sam_upscale_fused_kernel), bias2 tiled x4; hyper [R, 32];
// out [R, 4h, 4w].
extern "C" int rsp_sam_upscale_fused(const uint16_t* x_hi, const uint16_t* x_lo, int64_t x_rows, int32_t x_scale_log2,
                                     const uint16_t* w1_hi, const uint16_t* w1_lo, int32_t w1_scale_log2, const float* bias1,
                                     const float* gamma, const float* beta, float eps, const uint16_t* w2_hi,
                                     const uint16_t* w2_lo, int32_t w2_scale_log2, const float* bias2, const float* hyper,
                                     float* out, int64_t rows, int32_t rows_per_roi, int32_t W, rsp_stream_t stream) {
  if (!x_hi || !x_lo || !w1_hi || !w1_lo || !bias1 || !gamma || !beta || !w2_hi || !w2_lo || !bias2 || !hyper || !out ||
      rows <= 0 || x_rows < rows || rows_per_roi <= 0 || W <= 0 || (rows_per_roi % W) != 0 || (rows % rows_per_roi) != 0)
    return RSP_EINVAL;
  UpFP p;
  p.Xhi = reinterpret_cast<const half_t*>(x_hi); p.Xlo = reinterpret_cast<const half_t*>(x_lo); p.x_rows = x_rows;
  p.W1hi = reinterpret_cast<const half_t*>(w1_hi); p.W1lo = reinterpret_cast<const half_t*>(w1_lo);
  p.W2hi = reinterpret_cast<const half_t*>(w2_hi); p.W2lo = reinterpret_cast<const half_t*>(w2_lo);
  p.bias1 = bias1; p.gamma = gamma; p.beta = beta; p.bias2 = bias2; p.hyper = hyper; p.out = out;
  p.rows = rows; p.rows_per = rows_per_roi; p.W = W;
  p.alpha1 = ldexpf(1.0f, -(x_scale_log2 + w1_scale_log2));
  p.alpha2 = ldexpf(1.0f, -(UF_YS + w2_scale_log2));
  p.eps = eps;
  const int64_t nt = (rows + 127) / 128;
  if (nt > 0x7fffffffLL) return RSP_EINVAL;
  p.ntiles = (int)nt;
  const int64_t blocks = nt < 256 ? nt : 256;               // persistent: one block (4 waves, 96 KB of LDS) per CU
  hipLaunchKernelGGL(sam_upscale_fused_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}
