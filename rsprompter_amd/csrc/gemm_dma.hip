// fp16x3 GEMM, "plane" fast path: BOTH operands arrive as pre-split fp16 planes (hi, lo) and are
// streamed HBM -> LDS by the DMA engine (`global_load_lds_dwordx4`, 16 B per lane), bypassing the
// register file: no staging VGPRs, no conversion VALU in the main loop.  Same arithmetic as
// gemm.hip (a_lo*b_hi + a_hi*b_lo + a_hi*b_hi, fp32 accumulate in v_mfma_f32_32x32x16_f16).
//
// HBM layout ("KB32"): a plane of a [rows, K] matrix is stored K-blocked as [K/32][rows][32 halves],
// so the 128x32 (or 256x32) slice a K step needs is ONE contiguous run of full 128-B cache lines and
// a wave-wide DMA instruction reads 1 KiB contiguously (no 64-B half-line gathers, no power-of-two
// row stride hammering a single L2 channel).
//
// LDS image (per buffer): [A_hi | A_lo | B_hi | B_lo], each plane [rows][32 halves] = 64-B rows,
// written lane-linearly by the DMA (wave-uniform base + lane*16).  A row-major 64-B-row tile would
// be a 4-way bank conflict for the ds_read_b128 fragment reads (rows r and r+4 share a 16-B slot
// column), so the 16-B chunk index is XOR-swizzled with (row >> 2) & 3 -- applied on the per-lane
// GLOBAL source address (the DMA destination must stay linear) and again on the read.
//
// Pipeline (template PIPE, see the kernel): an NBUF-deep LDS ring fed by the DMA; the product variants
// (PIPE = 1 / 2) issue the fragment reads of the next half K tile before the MFMAs of the current one and spread the
// DMA instructions of the tile after next between the MFMA groups; ONE s_barrier per K tile.  The epilogue stages the
// accumulators through the (then idle) ring and stores row-wise, 4 consecutive columns per thread.
#include <type_traits>
#include "rsp_common.h"

namespace {

constexpr int BK = 32;
constexpr int ROWB = BK * 2;  // bytes per LDS row

__device__ uint4 g_zero_page[16];  // zeros: source of padded rows / out-of-image conv taps

struct GemmP {
  RspGemmDesc d;
  FastDiv fd_ctw, fd_resmod, fd_resb, fd_hd;   // ct_W, res_mod, res_brows, hd_rows
  int group_m;                                 // > 1: grouped tile order (see the kernel)
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// XCD-aware block remap (8 XCDs, block b runs on XCD b % 8): give every XCD a contiguous range of
// the logical block order so that blocks sharing an A panel / neighbouring B panels hit the SAME L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
  const unsigned q = nblk >> 3, r = nblk & 7u;
  const unsigned xcd = bid & 7u, idx = bid >> 3;
  const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// NBUF-deep LDS ring: the DMA of tile k+NBUF-1 is issued while tile k is computed and is only waited
// for NBUF-1 iterations later with a COUNTED s_waitcnt vmcnt (never 0 in steady state) + a raw
// s_barrier, so HBM/L2 latency is covered by several K steps instead of one
// (cdna_hip_programming.md §5 "Pipelining across barriers", T3+T4).
// ABL: ablation switch for the tuning probe (0 = product kernel, 1 = skip the DMA inside the K loop,
// 2 = skip the MFMAs); results are garbage for ABL != 0.
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// EPI: reserved (0); the fused LayerNorm / hyper-network epilogues are run-time modes of the generic epilogue.
// PIPE: 0 = plain ring, 1 = register-pipelined loop, 2 = 1 + DMA instructions spread between the MFMA groups.
// CONV: implicit-GEMM 3x3 convolution loader (A addresses from (pixel, tap)).
// ORD: 0 = the three products of an accumulator back to back; 1 = pass-major (all a_lo b_hi, then all a_hi b_lo, then all
// a_hi b_hi: dependent MFMAs TM*TN issue slots apart) -- tuning variant.
// F8: fp8-corrected product (plane format word bit RSP_PLANE_F8): the "lo" planes of A and B are cat8 planes
// [lo8 x 32 | hi8 x 32] per row and K block.  Half tile 0 = a_hi b_hi for BOTH 16-wide K halves (2 fp16 MFMAs per
// accumulator), half tile 1 = ONE v_mfma_scale_f32_32x32x64_f8f6f4 per accumulator: lanes 0-31 feed a_lo8 against b_hi8,
// lanes 32-63 a_hi8 against b_lo8 (its K = 64 is the two correction products side by side), the per-lane E8M0 block
// scales undo the storage scales.  Same LDS image, same 12 ds_read_b128 per half tile, 128 instead of 192 matrix cycles
// per accumulator and K tile.
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
template <int BM, int BN, int WGM, int WGN, int NBUF, int ABL = 0, int EPI = 0, int PIPE = 0, bool CONV = false, int ORD = 0,
          bool F8 = false>
__global__ __launch_bounds__(WGM * WGN * 64) void gemm_f16x3_dma_kernel(const GemmP p) {
  static_assert(!F8 || (ORD == 0 && !CONV), "fp8-corrected product: plain GEMMs, accumulator-major order");
  constexpr int NT = WGM * WGN * 64;
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int A_UNITS = 2 * BM * 4, B_UNITS = 2 * BN * 4;   // 16-byte units per buffer
  constexpr int NA = A_UNITS / NT, NB = (B_UNITS + NT - 1) / NT;
  static_assert(A_UNITS % NT == 0 && B_UNITS % NT == 0, "tile/threads mismatch");
  constexpr int BUF_BYTES = (A_UNITS + B_UNITS) * 16;
  constexpr int OFF_ALO = BM * ROWB, OFF_BHI = 2 * BM * ROWB, OFF_BLO = 2 * BM * ROWB + BN * ROWB;

  constexpr int LPT = NA + NB;   // DMA instructions per wave per tile
  static_assert(NBUF >= 2 && NBUF <= 4 && (NBUF - 2 + (PIPE != 0)) * LPT < 64, "ring depth / vmcnt range");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NBUF][BUF_BYTES];

  const RspGemmDesc& d = p.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int hh = lane >> 5, l31 = lane & 31;
  const int M = d.M, N = d.N, K = d.K;
  const int nbn = (N + BN - 1) / BN;
  const unsigned lbid = xcd_remap(blockIdx.x, gridDim.x);
  int mb = (int)(lbid / nbn), nb_ = (int)(lbid % nbn);
  if (p.group_m > 1) {
    // grouped order inside the XCD's contiguous range: group_m M-blocks x all N-blocks, M fastest -- the ~32 tiles an
    // XCD runs at once then cover a (group_m x 32/group_m) patch of the output, i.e. group_m A panels + 32/group_m W
    // panels per K step in its 4 MB L2 instead of ~2 A panels + every W panel (PMC r2: W re-fetched 10x per XCD)
    const int nbm = (M + BM - 1) / BM;
    const int per = p.group_m * nbn;
    const int grp = (int)(lbid / (unsigned)per), rem = (int)(lbid % (unsigned)per);
    const int first = grp * p.group_m;
    const int gsz = min(nbm - first, p.group_m);
    mb = first + rem % gsz;
    nb_ = rem / gsz;
  }
  const int m0 = mb * BM, n0 = nb_ * BN;
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(g_zero_page);

  // ---- per-thread DMA slots.  A unit u = i*256 + tid: plane = u / (BM*4), row = (u % (BM*4)) / 4,
  //      physical 16-B position pos = u & 3 holds logical chunk pos ^ ((row >> 2) & 3).
  // When a plane is a whole number of slots (A_PAIR / B_PAIR: every product tile) slot i + N/2 addresses the SAME rows
  // of the second plane: only the first half of the slots keeps per-thread state (pointers are 2 VGPRs each), the
  // plane distance rides on the wave-uniform K offset.
  constexpr bool A_PAIR = (BM * 4) % NT == 0, B_PAIR = (BN * 4) % NT == 0;
  constexpr int NAH = A_PAIR ? NA / 2 : NA, NBH = B_PAIR ? NB / 2 : NB;
  const unsigned char* a_src[NAH];  // plain mode: row base (bytes) incl. chunk offset; conv: plane base
  bool a_ok[NAH];
  int a_chunkb[NAH];                // byte offset of the logical chunk inside the K tile
  int a_pix[NAH];                   // conv: batch pixel base (B * H * W < 2^31: checked by the dispatcher)
  int a_y[NAH], a_x[NAH];
  const int64_t a_plane_d = reinterpret_cast<const unsigned char*>(d.Alo) - reinterpret_cast<const unsigned char*>(d.Ahi);
  const int64_t b_plane_d = reinterpret_cast<const unsigned char*>(d.Blo) - reinterpret_cast<const unsigned char*>(d.Bhi);
#pragma unroll
  for (int i = 0; i < NAH; ++i) {
    const int u = i * NT + tid;
    const int plane = u / (BM * 4);
    const int row = (u % (BM * 4)) >> 2;
    const int chunk = (u & 3) ^ ((row >> 2) & 3);
    const uint16_t* base = plane == 0 ? d.Ahi : d.Alo;
    const int gm = m0 + row;
    a_ok[i] = gm < M;
    a_chunkb[i] = chunk * 16;
    a_src[i] = reinterpret_cast<const unsigned char*>(base);
    a_pix[i] = 0; a_y[i] = 0; a_x[i] = 0;
    if (a_ok[i]) {
      if constexpr (!CONV) {
        const int srow = d.a_rowmap ? d.a_rowmap[gm] : gm;
        if (srow < 0) a_ok[i] = false;
        a_src[i] += (int64_t)srow * ROWB + chunk * 16;
      } else {
        const int hw = d.conv_Ho * d.conv_Wo;
        const int b = gm / hw;
        const int rem = gm - b * hw;
        const int yo = rem / d.conv_Wo;
        const int xo = rem - yo * d.conv_Wo;
        a_pix[i] = b * d.conv_H * d.conv_W;
        a_y[i] = yo * d.conv_stride - d.conv_pad;
        a_x[i] = xo * d.conv_stride - d.conv_pad;
      }
    }
  }
  const unsigned char* b_src[NBH];
  bool b_ok[NBH], b_in[NBH];
#pragma unroll
  for (int i = 0; i < NBH; ++i) {
    const int u = i * NT + tid;
    b_in[i] = u < B_UNITS;
    const int plane = u / (BN * 4);
    const int row = (u % (BN * 4)) >> 2;
    const int chunk = (u & 3) ^ ((row >> 2) & 3);
    const uint16_t* base = plane == 0 ? d.Bhi : d.Blo;
    b_ok[i] = b_in[i] && (n0 + row < N);
    b_src[i] = reinterpret_cast<const unsigned char*>(base) + (int64_t)(n0 + row) * ROWB + chunk * 16;
  }

  const int64_t a_kstride = (int64_t)d.a_rows * ROWB;   // bytes between consecutive K blocks of an A plane
  const int64_t b_kstride = (int64_t)(d.b_rows > 0 ? d.b_rows : N) * ROWB;
  // branch-free: every lane always issues its DMA; out-of-range sources read the zero page.
  // TileK = the wave-uniform part of a K tile's addresses; issue_slot<I> = ONE 1-KiB DMA instruction (slots
  // [0, NA) are A, [NA, LPT) are B), so that the pipelined loop can spread them between its MFMAs.
  struct TileK { int64_t koff, koffb; int ky, kx; };
  auto tile_k = [&](int k0) {
    TileK t;
    const int kb = k0 / BK;
    t.koffb = (int64_t)kb * b_kstride;
    t.ky = 0; t.kx = 0;
    if constexpr (!CONV) {
      t.koff = (int64_t)kb * a_kstride;
    } else {
      const int tap = k0 / d.conv_C;
      const int c0 = k0 - tap * d.conv_C;
      t.ky = tap / d.conv_k;
      t.kx = tap - t.ky * d.conv_k;
      t.koff = (int64_t)(c0 / BK) * a_kstride;
    }
    return t;
  };
  auto issue_slot = [&](auto ic, const TileK& t, unsigned char* lbase) {
    constexpr int I = decltype(ic)::value;
    if constexpr (I < NA) {
      constexpr int H = I % NAH;                                   // per-thread state of this slot
      const int64_t koff = (I >= NAH) ? t.koff + a_plane_d : t.koff;   // wave-uniform (scalar) part
      const unsigned char* src;
      if constexpr (!CONV) {
        src = a_ok[H] ? a_src[H] + koff : zero;
      } else {
        const int y = a_y[H] + t.ky, x = a_x[H] + t.kx;
        const bool inb = a_ok[H] && (unsigned)y < (unsigned)d.conv_H && (unsigned)x < (unsigned)d.conv_W;
        src = a_src[H] + koff + (int64_t)(a_pix[H] + y * d.conv_W + x) * ROWB + a_chunkb[H];
        src = inb ? src : zero;
      }
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lbase + (I * NT + wave * 64) * 16), 16, 0, 0);
    } else {
      constexpr int J = I - NA, H = J % NBH;
      const int64_t koffb = (J >= NBH) ? t.koffb + b_plane_d : t.koffb;
      const unsigned char* src = b_ok[H] ? b_src[H] + koffb : zero;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lbase + (A_UNITS + J * NT + wave * 64) * 16), 16, 0, 0);
    }
  };
  auto issue_tile = [&](int k0, int buf) {
    const TileK t = tile_k(k0);
    unsigned char* lbase = &smem[buf][0];
    static_for<0, LPT>([&](auto ic) { issue_slot(ic, t, lbase); });
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // per-lane fragment byte offsets (row * 64 + swizzled chunk), fixed across K tiles
  int a_off[TM][2], b_off[TN][2];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int r = wm * WTM + i * 32 + l31;
#pragma unroll
    for (int s = 0; s < 2; ++s) a_off[i][s] = r * ROWB + (((s * 2 + hh) ^ ((r >> 2) & 3)) << 4);
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int r = wn * WTN + j * 32 + l31;
#pragma unroll
    for (int s = 0; s < 2; ++s) b_off[j][s] = r * ROWB + (((s * 2 + hh) ^ ((r >> 2) & 3)) << 4);
  }

  // F8: cat8 fragment = 32 bytes per lane = chunks (2h, 2h+1) of the row; A: h = lane half (0 -> lo8, 1 -> hi8),
  // B: h = the OTHER half (lanes 0-31 read hi8, lanes 32-63 lo8)
  int a8_off[F8 ? TM : 1][2], b8_off[F8 ? TN : 1][2];
  if constexpr (F8) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int r = wm * WTM + i * 32 + l31;
#pragma unroll
      for (int t = 0; t < 2; ++t) a8_off[i][t] = OFF_ALO + r * ROWB + (((hh * 2 + t) ^ ((r >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int r = wn * WTN + j * 32 + l31;
#pragma unroll
      for (int t = 0; t < 2; ++t) b8_off[j][t] = OFF_BLO + r * ROWB + ((((1 - hh) * 2 + t) ^ ((r >> 2) & 3)) << 4);
    }
  }
  // E8M0 block scales (one per lane = per 32 k of the K = 64 instruction)
  // (byte 0 = A's scale, byte 1 = B's: one register, selected with op_sel)
  const int sc_ab = hh ? (127 + RSP_F8_HI_EXP) | ((127 - RSP_F8_LO_EXP) << 8) : (127 - RSP_F8_LO_EXP) | ((127 + RSP_F8_HI_EXP) << 8);

  const int nk = K / BK;

  // F8: half tile 0 holds {ah = a_hi(k 0..15), al = a_hi(k 16..31), bh / bl likewise}, half tile 1 the two 16-byte
  // pieces of the cat8 fragments
  struct Frags { half8_t ah[TM], al[TM], bh[TN], bl[TN]; };
  auto read_frags = [&](const unsigned char* sb, int s, Frags& f) {
    if constexpr (F8) {
      if (s == 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          f.ah[i] = *reinterpret_cast<const half8_t*>(sb + a_off[i][0]);
          f.al[i] = *reinterpret_cast<const half8_t*>(sb + a_off[i][1]);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          f.bh[j] = *reinterpret_cast<const half8_t*>(sb + OFF_BHI + b_off[j][0]);
          f.bl[j] = *reinterpret_cast<const half8_t*>(sb + OFF_BHI + b_off[j][1]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          f.ah[i] = *reinterpret_cast<const half8_t*>(sb + a8_off[i][0]);
          f.al[i] = *reinterpret_cast<const half8_t*>(sb + a8_off[i][1]);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          f.bh[j] = *reinterpret_cast<const half8_t*>(sb + b8_off[j][0]);
          f.bl[j] = *reinterpret_cast<const half8_t*>(sb + b8_off[j][1]);
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      f.ah[i] = *reinterpret_cast<const half8_t*>(sb + a_off[i][s]);
      f.al[i] = *reinterpret_cast<const half8_t*>(sb + OFF_ALO + a_off[i][s]);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      f.bh[j] = *reinterpret_cast<const half8_t*>(sb + OFF_BHI + b_off[j][s]);
      f.bl[j] = *reinterpret_cast<const half8_t*>(sb + OFF_BLO + b_off[j][s]);
    }
  };
  // the matrix work of ONE accumulator for half tile s
  auto mfma_one = [&](const Frags& f, int s, int i, int j) {
    if constexpr (F8) {
      if (s == 0) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[i], f.bl[j], acc[i][j], 0, 0, 0);
      } else {
        const i32x8 a8 = __builtin_shufflevector(__builtin_bit_cast(i32x4, f.ah[i]), __builtin_bit_cast(i32x4, f.al[i]),
                                                 0, 1, 2, 3, 4, 5, 6, 7);
        const i32x8 b8 = __builtin_shufflevector(__builtin_bit_cast(i32x4, f.bh[j]), __builtin_bit_cast(i32x4, f.bl[j]),
                                                 0, 1, 2, 3, 4, 5, 6, 7);
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[i][j], 0, 0, 0, sc_ab, 1, sc_ab);
      }
    } else {
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[i], f.bh[j], acc[i][j], 0, 0, 0);
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bl[j], acc[i][j], 0, 0, 0);
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bh[j], acc[i][j], 0, 0, 0);
    }
  };
  auto mfma_frags = [&](const Frags& f, int s) {
    if constexpr (ORD == 1 && ABL != 2) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[i], f.bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bh[j], acc[i][j], 0, 0, 0);
      return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (ABL == 2) {   // keep the fragments live without the matrix work
          asm volatile("" ::"v"(f.al[i]), "v"(f.ah[i]), "v"(f.bl[j]), "v"(f.bh[j]));
        } else {
          mfma_one(f, s, i, j);
        }
      }
  };

  // the same matrix work with the LPT DMA instructions of one K tile spread between the (i, j) groups: a DMA costs
  // its wave 60-185 issue cycles (MI355X_MICROARCH.md, per-instruction constants); issued as one burst after the
  // barrier both waves of a SIMD stall together and the MFMA pipe idles, spread out they cover each other
  auto mfma_frags_dma = [&](const Frags& f, int s, int k0, int buf, bool do_issue) {
    const TileK t = tile_k(k0);
    unsigned char* lbase = &smem[buf][0];
    constexpr int G = TM * TN;
    if constexpr (ORD == 1) {
      // pass-major: 3 G MFMAs in (pass, i, j) order, the DMA slots spread over the 3 G positions
      static_for<0, 3 * G>([&](auto gc) {
        constexpr int q = decltype(gc)::value, ps = q / G, g = q % G, i = g / TN, j = g % TN;
        if constexpr (ps == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[i], f.bh[j], acc[i][j], 0, 0, 0);
        else if constexpr (ps == 1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bl[j], acc[i][j], 0, 0, 0);
        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bh[j], acc[i][j], 0, 0, 0);
        if constexpr ((q + 1) * LPT / (3 * G) > q * LPT / (3 * G)) {
          __builtin_amdgcn_sched_barrier(0);
          if (do_issue) static_for<q * LPT / (3 * G), (q + 1) * LPT / (3 * G)>([&](auto sc) { issue_slot(sc, t, lbase); });
          __builtin_amdgcn_sched_barrier(0);
        }
      });
      return;
    }
    static_for<0, G>([&](auto gc) {
      constexpr int g = decltype(gc)::value, i = g / TN, j = g % TN;
      mfma_one(f, s, i, j);
      // F8: without this pin the optimizer sinks the (side-effect free) fp8 MFMAs below all the `if (do_issue)` branches:
      // eight DMA instructions in one burst, then eight MFMAs -- the order this function exists to avoid
      if constexpr (F8) asm volatile("" : "+v"(acc[i][j]));
      __builtin_amdgcn_sched_barrier(0);
      if (do_issue) static_for<g * LPT / G, (g + 1) * LPT / G>([&](auto sc) { issue_slot(sc, t, lbase); });
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  if constexpr (PIPE == 0) {
    // ---- plain ring: [wait tile kt | barrier | issue tile kt+NBUF-1 | read+MFMA s0 | read+MFMA s1] ----
#pragma unroll
    for (int t = 0; t < NBUF - 1; ++t)
      if (t < nk) issue_tile(t * BK, t);
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
      // tile kt must have landed (this wave's part), newer tiles may stay in flight
      const int ahead = min(NBUF - 2, nk - 1 - kt);
      if (ahead >= 2) wait_vmcnt<2 * LPT>();
      else if (ahead == 1) wait_vmcnt<LPT>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();          // ... and everybody else's part; also: all reads of tile kt-1 are done
      if (ABL != 1 && kt + NBUF - 1 < nk) {
        int nb = buf + NBUF - 1;
        if (nb >= NBUF) nb -= NBUF;
        issue_tile((kt + NBUF - 1) * BK, nb);
      }
      const unsigned char* sb = &smem[buf][0];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        Frags f;
        read_frags(sb, s, f);
        mfma_frags(f, s);
      }
      if (++buf == NBUF) buf = 0;
    }
  } else {
    // ---- register-pipelined ring: the LDS reads of the NEXT half tile are issued before the MFMAs of the current
    // one, also across the tile boundary, so LDS latency and the barrier hide under matrix work:
    //   step kt:  F1 = read(kt, s1) ; MFMA(F0)
    //             wait DMA(kt+1), lgkmcnt(0) ; barrier   (tile kt+1 complete everywhere, buffer kt drained)
    //             issue DMA(kt+NBUF) -> buffer kt ; F0 = read(kt+1, s0) ; MFMA(F1)
    // All NBUF buffers are filled up front; a buffer is refilled half a step after its last read.
#pragma unroll
    for (int t = 0; t < NBUF; ++t)
      if (t < nk) issue_tile(t * BK, t);
    {
      const int ahead = min(NBUF - 1, nk - 1);
      if (ahead >= 3) wait_vmcnt<3 * LPT>();
      else if (ahead == 2) wait_vmcnt<2 * LPT>();
      else if (ahead == 1) wait_vmcnt<LPT>();
      else wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    // s_waitcnt through the builtin (the compiler's own waitcnt insertion understands it, inline asm it does not):
    // gfx9 encoding vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14]
    constexpr int WC_LGKM0 = 0xC07F;                                                     // lgkmcnt(0) only
    constexpr auto wc_vm_lgkm0 = [](int n) { return (n & 15) | ((n >> 4) << 14) | 0x70; };   // vmcnt(n) lgkmcnt(0)
    Frags f0, f1;
    read_frags(&smem[0][0], 0, f0);
    int buf = 0;
    for (int kt = 0; kt + 1 < nk; ++kt) {        // every step but the last: straight-line, no joins
      __builtin_amdgcn_s_waitcnt(WC_LGKM0);       // f0 has landed (issued half a step ago: free)
      read_frags(&smem[buf][0], 1, f1);
      __builtin_amdgcn_sched_barrier(0);          // keep the LDS reads AHEAD of the matrix work they overlap with
      mfma_frags(f0, 0);
      __builtin_amdgcn_sched_barrier(0);          // ... and the matrix work ahead of the wait + barrier it hides
      int nb = buf + 1;
      if (nb == NBUF) nb = 0;
      const int ahead = min(NBUF - 2, nk - 2 - kt);   // tiles beyond kt+1 that may stay in flight
      if (NBUF >= 4 && ahead >= 2) __builtin_amdgcn_s_waitcnt(wc_vm_lgkm0(NBUF >= 4 ? 2 * LPT : 0));
      else if (NBUF >= 3 && ahead == 1) __builtin_amdgcn_s_waitcnt(wc_vm_lgkm0(NBUF >= 3 ? LPT : 0));
      else __builtin_amdgcn_s_waitcnt(wc_vm_lgkm0(0));
      __builtin_amdgcn_s_barrier();
      read_frags(&smem[nb][0], 0, f0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (PIPE == 2) {
        mfma_frags_dma(f1, 1, (kt + NBUF) * BK, buf, ABL == 0 && kt + NBUF < nk);   // DMA spread between the MFMAs
      } else {
        if (ABL == 0 && kt + NBUF < nk) issue_tile((kt + NBUF) * BK, buf);
        __builtin_amdgcn_sched_barrier(0);
        mfma_frags(f1, 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      buf = nb;
    }
    __builtin_amdgcn_s_waitcnt(WC_LGKM0);
    read_frags(&smem[buf][0], 1, f1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_frags(f0, 0);
    mfma_frags(f1, 1);
  }

  // ---- epilogue (same contract as gemm.hip) + optional fp16-plane output for the next GEMM ----
  // The (i, j) loops are expanded by template recursion: a #pragma unroll the optimizer declines here turns the
  // accumulator indices dynamic and sends all of acc[][] to scratch.
  const float alpha = d.alpha;
  const float cs = d.Chi ? ldexpf(1.0f, RSP_PLANE_EXP(d.c_scale_log2)) : 1.0f;
  const bool c_f8 = RSP_PLANE_IS_F8(d.c_scale_log2);   // output planes in the cat8 format (N % 4 == 0 checked by the host)
  half_t* const chi = reinterpret_cast<half_t*>(d.Chi);
  half_t* const clo = reinterpret_cast<half_t*>(d.Clo);

  {
    // generic epilogue, staged through LDS: the accumulators (lane <-> column, register <-> row) are written to the
    // now idle ring as an fp32 tile [rows][BN] (alpha, bias and activation already applied), then read back ROW-wise
    // so that every thread owns 4 consecutive columns of one row: residual loads, fp32 stores and plane stores are
    // 16-/8-byte accesses on whole cache lines (a wave writes one full output row per instruction) instead of one
    // dword (or one half!) per lane.  Row-only work (row maps, ConvTranspose / residual row arithmetic with the
    // host-prepared magic division) happens once per 4 outputs.
    constexpr int LDS_FLOATS = NBUF * BUF_BYTES / 4;
    constexpr int IP_MAX = LDS_FLOATS / (WGM * 32 * BN);
    static_assert(IP_MAX >= 1, "epilogue tile does not fit the LDS ring");
    constexpr int IP = IP_MAX >= TM ? TM : (IP_MAX >= 2 && TM % 2 == 0 ? 2 : 1);   // i-tiles per pass
    constexpr int NPASS = TM / IP;
    constexpr int ROWS_P = WGM * 32 * IP;
    constexpr int C4 = BN / 4;                      // float4 units per tile row
    float* stile = reinterpret_cast<float*>(&smem[0][0]);
    const bool ct = d.ct_W > 0;
    const int act_w = d.ln_gamma ? RSP_ACT_NONE : d.act;   // LayerNorm mode: activation after the normalisation
    float bvj[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * WTN + j * 32 + l31;
      bvj[j] = (d.bias && col < N) ? d.bias[col] : 0.f;
    }
    // vector path needs 16-byte aligned rows everywhere it touches memory
    const bool vec = ((N & 3) == 0) && (!d.C || (((d.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(d.C) & 15) == 0))) &&
                     (!d.res || (((d.ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(d.res) & 15) == 0))) &&
                     (!ct || ((N >> (d.ct_dy < 0 ? 2 : 1)) & 3) == 0);
    __syncthreads();                                // every wave is done reading the last K tile
    static_for<0, NPASS>([&](auto pc) {
      constexpr int pass = decltype(pc)::value;
      static_for<0, IP>([&](auto iic) {
        constexpr int ii = decltype(iic)::value, i = pass * IP + ii;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rowp = wm * (32 * IP) + ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          const int sw = BN >= 64 ? ((rowp >> 2) & 1) << 5 : 0;   // XOR swizzle: the half waves hit different banks
          static_for<0, TN>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const int colp = wn * WTN + j * 32 + l31;
            stile[rowp * BN + (colp ^ sw)] = rsp_act(acc[i][j][r] * alpha + bvj[j], act_w);
          });
        }
      });
      __syncthreads();
      for (int u = tid; u < ROWS_P * C4; u += NT) {
        const int rowp = u / C4, c4 = u - rowp * C4;
        const int c4p = BN >= 64 ? c4 ^ (((rowp >> 2) & 1) << 3) : c4;
        const f32x4 t = *reinterpret_cast<const f32x4*>(&stile[rowp * BN + (c4p << 2)]);
        const int wmr = rowp / (32 * IP), rin = rowp - wmr * (32 * IP);     // rin = ii*32 + row in 32
        const int row = m0 + wmr * WTM + pass * IP * 32 + rin;
        const int col = n0 + c4 * 4;
        if (row >= M || col >= N) continue;
        int cr = d.c_rowmap ? d.c_rowmap[row] : row;
        if (cr < 0) continue;
        // ConvTranspose(k2,s2) column decode: ct_dy >= 0 -> columns are (dx, co) of one output-row parity;
        // ct_dy < 0 -> columns are (dy, dx, co), all four sub-pixels in one GEMM (A is read once)
        int dy = d.ct_dy, ccol = col, ct_c = N >> 1, crow = cr;
        if (ct) {
          if (d.ct_dy < 0) { dy = col >= (N >> 1); ccol = col - dy * (N >> 1); ct_c = N >> 2; }
          const int yy = p.fd_ctw.div(cr);
          crow = (yy * 2 + dy) * d.ct_W + (cr - yy * d.ct_W);
        }
        f32x4 v = t;
        if (d.hd_out) {
          // last ConvTranspose of the SAM upscaler + GELU + <., hyper_in> (HF:519-531): the 32 channels of one output
          // sub-pixel are the 8 consecutive lanes of this row; nothing of the [R, 4h, 4w, 32] tensor is stored.
          // (row validity is uniform over a row's lanes, so whole shuffle groups are active or idle together)
          const int roi = p.fd_hd.div(row);
          const f32x4 hy = *reinterpret_cast<const f32x4*>(d.hd_hyper + (int64_t)roi * 32 + (col & 31));
          float sdot = v[0] * hy[0] + v[1] * hy[1] + v[2] * hy[2] + v[3] * hy[3];
          sdot += __shfl_xor(sdot, 1, 64);
          sdot += __shfl_xor(sdot, 2, 64);
          sdot += __shfl_xor(sdot, 4, 64);
          if ((c4 & 7) == 0) {
            const int pix = row - roi * d.hd_rows;            // input pixel (y, x) of the ConvTranspose
            const int y = p.fd_ctw.div(pix), x = pix - y * d.ct_W;
            const int sp = col >> 5;                          // sub-pixel: (dy, dx) when ct_dy < 0, else dx
            const int sdy = d.ct_dy < 0 ? (sp >> 1) : d.ct_dy, sdx = sp & 1;
            d.hd_out[(int64_t)roi * (4 * d.hd_rows) + (int64_t)(2 * y + sdy) * (2 * d.ct_W) + 2 * x + sdx] = sdot;
          }
          continue;
        }
        if (d.ln_gamma) {
          // first ConvTranspose of the SAM upscaler + LayerNorm2d over the 64 channels of each output sub-pixel
          // (HF:517-520) = the 16 consecutive lanes of this row, two-pass statistics like the LayerNorm kernel
          float sm = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) sm += __shfl_xor(sm, o, 64);
          const float mean = sm * (1.0f / 64.0f);
          float sq = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float dl = v[e] - mean; sq += dl * dl; }
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) sq += __shfl_xor(sq, o, 64);
          const float rstd = 1.0f / sqrtf(sq * (1.0f / 64.0f) + d.ln_eps);
          const int ch = col & 63;
          const f32x4 g4 = *reinterpret_cast<const f32x4*>(d.ln_gamma + ch);
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(d.ln_beta + ch);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = rsp_act((v[e] - mean) * rstd * g4[e] + b4[e], d.act);
        }
        const int nv = vec ? 4 : min(4, N - col);
        if (d.res_hi) {
          // residual from fp16 planes (hi + lo carries ~22 significant bits of the fp32 value)
          int64_t rrow = crow;
          if (d.res_mod > 0) rrow = crow - p.fd_resmod.div(crow) * d.res_mod;
          if (d.res_bmap) {
            const int rb = p.fd_resb.div(crow);
            rrow = (int64_t)d.res_bmap[rb] * d.res_brows + (crow - rb * d.res_brows);
          }
          const int64_t ro = ((int64_t)(col >> 5) * d.res_rows + rrow) * 32 + (col & 31);
          const half4_t rh = *reinterpret_cast<const half4_t*>(reinterpret_cast<const half_t*>(d.res_hi) + ro);
          const half4_t rl = *reinterpret_cast<const half4_t*>(reinterpret_cast<const half_t*>(d.res_lo) + ro);
          const float rsc = ldexpf(1.0f, -RSP_PLANE_EXP(d.res_scale_log2));
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += ((float)rh[e] + (float)rl[e]) * rsc;
        }
        if (d.res) {
          int64_t rrow = crow;
          if (d.res_mod > 0) rrow = crow - p.fd_resmod.div(crow) * d.res_mod;
          if (d.res_bmap) {
            const int rb = p.fd_resb.div(crow);
            rrow = (int64_t)d.res_bmap[rb] * d.res_brows + (crow - rb * d.res_brows);
          }
          const float* rp = d.res + rrow * d.ldr + col;
          if (vec) {
            const f32x4 rv = *reinterpret_cast<const f32x4*>(rp);
            v[0] += rv[0]; v[1] += rv[1]; v[2] += rv[2]; v[3] += rv[3];
          } else {
            for (int e = 0; e < nv; ++e) v[e] += rp[e];
          }
        }
        if (d.act == RSP_ACT_RELU_POST) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = rsp_act_post(v[e], d.act);
        }
        if (d.C && (d.c_ncols <= 0 || col < d.c_ncols)) {
          float* cp = d.C + (int64_t)crow * d.ldc + ccol;
          if (vec) *reinterpret_cast<f32x4*>(cp) = v;
          else for (int e = 0; e < nv; ++e) cp[e] = v[e];
        }
        if (d.Chi && col >= d.pl_col0) {
          // KB32 planes of the [c_rows, N - pl_col0] result (columns from pl_col0 on); for a ConvTranspose the planes
          // are those of the NHWC output [.., ct_c]: the sub-pixel dx folds into the row index
          int64_t prow = crow;
          int pch = col - d.pl_col0;
          if (ct) { const int dx = ccol >= ct_c; pch = ccol - dx * ct_c; prow = (int64_t)crow * 2 + dx; }
          const int64_t po = ((int64_t)(pch >> 5) * d.c_rows + prow) * 32 + (pch & 31);
          if (vec || c_f8) {
            rsp_store_planes4(chi, clo, po, f32x4{v[0] * cs, v[1] * cs, v[2] * cs, v[3] * cs}, c_f8);
          } else {
            half4_t h4, l4;
#pragma unroll
            for (int e = 0; e < 4; ++e) { half_t a, b; rsp_split1(v[e] * cs, a, b); h4[e] = a; l4[e] = b; }
            for (int e = 0; e < nv; ++e) {      // ragged N: element-wise (a 4-group may straddle a 32-column block)
              const int pc = pch + e;
              const int64_t pe = ((int64_t)(pc >> 5) * d.c_rows + prow) * 32 + (pc & 31);
              chi[pe] = h4[e];
              clo[pe] = l4[e];
            }
          }
        }
      }
      if constexpr (pass + 1 < NPASS) __syncthreads();
    });
  }
}

template <int BM, int BN, int WGM, int WGN, int NBUF, int ABL = 0, int EPI = 0, int PIPE = 0, bool CONV = false, int ORD = 0,
          bool F8 = false>
int launch_dma(const RspGemmDesc& d, hipStream_t s) {
  GemmP p; p.d = d;
  p.group_m = (d.tile_hint >> 8) & 0xff;      // tuning field: tile_hint = tile | group_m << 8
  if (p.group_m == 0 && BM == 256 && BN == 256 && !CONV) p.group_m = 8;   // default for the big tile (1-4 % on the ViT-H shapes)
  p.fd_ctw = make_fastdiv(d.ct_W); p.fd_resmod = make_fastdiv(d.res_mod);
  p.fd_resb = make_fastdiv(d.res_brows); p.fd_hd = make_fastdiv(d.hd_rows);
  const long long nblk = (long long)((d.N + BN - 1) / BN) * ((d.M + BM - 1) / BM);
  if (nblk > 0x7fffffffLL) return RSP_EINVAL;
  hipLaunchKernelGGL((gemm_f16x3_dma_kernel<BM, BN, WGM, WGN, NBUF, ABL, EPI, PIPE, CONV, ORD, F8>), dim3((unsigned)nblk), dim3(WGM * WGN * 64), 0, s, p);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

}  // namespace

bool rsp_gemm_s2_eligible(const RspGemmDesc& d);                       // gemm_s2.hip
int rsp_gemm_s2_auto(const RspGemmDesc& d);
int rsp_gemm_s2_dispatch(const RspGemmDesc& d, int var, hipStream_t s);
bool rsp_gemm_pp_eligible(const RspGemmDesc& d);                       // gemm_pp.hip
int rsp_gemm_pp_auto(const RspGemmDesc& d);
int rsp_gemm_pp_dispatch(const RspGemmDesc& d, int var, hipStream_t s);

// called from rsp_gemm (gemm.hip) when the descriptor carries A planes
int rsp_gemm_dma_dispatch(const RspGemmDesc& d, hipStream_t s) {
  if (d.a_rows <= 0) return RSP_EINVAL;
  // tile hints 40 + v: the two-blocks-per-CU persistent kernel (gemm_s2.hip), experiment variant v
  if ((d.tile_hint & 0xff) >= 40 && (d.tile_hint & 0xff) < 168) {
    if (!rsp_gemm_s2_eligible(d)) return RSP_EINVAL;
    return rsp_gemm_s2_dispatch(d, (d.tile_hint & 0xff) - 40, s);
  }
  // tile hint 200: the ping-pong kernel (gemm_pp.hip), forced; product rule (round 5): wherever its 256 x 256 tiles fill
  // the CUs for several rounds
  // 200 + v: gemm_pp.hip, variant v: bit 0 = the 128 x 256 tile; development builds: 4 / 8 / 12 ablations, 16 = time stamps
  if ((d.tile_hint & 0xff) >= 200) {
    const int v = (d.tile_hint & 0xff) - 200;
    return rsp_gemm_pp_dispatch(d, (v & 16) ? 32 | (v & 1) : v, s);
  }
  if ((d.tile_hint & 0xff) == 0) {
    const int bm = rsp_gemm_pp_auto(d);
    if (bm) return rsp_gemm_pp_dispatch(d, bm == 128 ? 1 : 0, s);
  }
  // product rule (round 3): the two-blocks-per-CU persistent kernel wherever it has a specialised epilogue -- 7-20 %
  // faster than the kernels below on the ViT-H encoder shapes (tools/gemm_s2_exp.py).  Tile hint 1 = "the round-2 rule".
  if ((d.tile_hint & 0xff) == 0 && rsp_gemm_s2_auto(d)) return rsp_gemm_s2_dispatch(d, 0, s);
  if (d.ct_W > 0 && (d.res || d.res_hi)) return RSP_EINVAL;   // no caller needs a residual on a ConvTranspose
  if (d.res_hi && (!d.res_lo || d.res || d.res_rows <= 0 || (d.N & 3))) return RSP_EINVAL;
  if (d.hd_out && (!d.hd_hyper || d.ct_W <= 0 || d.N != (d.ct_dy < 0 ? 128 : 64) || d.hd_rows <= 0)) return RSP_EINVAL;
  if (d.ct_W > 0 && d.ct_dy < 0 && (d.N & 127)) return RSP_EINVAL;
  if (d.ct_W > 0 && d.Chi && ((d.N >> (d.ct_dy < 0 ? 2 : 1)) & 31)) return RSP_EINVAL;
  if (d.conv_k != 0 && (d.conv_C % BK) != 0) return RSP_EINVAL;
  if (d.conv_k != 0 && (long long)d.conv_H * d.conv_W * ((d.M + (long long)d.conv_Ho * d.conv_Wo - 1) / ((long long)d.conv_Ho * d.conv_Wo)) > 0x7fffffffLL)
    return RSP_EINVAL;   // the implicit-GEMM loader indexes input pixels with 32 bits
  if (d.ln_gamma && (!d.ln_beta || !(d.Chi && d.Clo) || d.C || d.ct_W <= 0 || d.ct_dy >= 0 || d.N != 256 || d.res ||
                     d.c_rowmap || d.hd_out))
    return RSP_EINVAL;
  if (d.hd_out && (d.C || d.Chi || d.res || d.c_rowmap)) return RSP_EINVAL;
  if ((d.pl_col0 || d.c_ncols) && (d.ct_W > 0 || (d.pl_col0 & 31) || (d.c_ncols & 3) || d.pl_col0 < 0 || (d.N & 3)))
    return RSP_EINVAL;   // column-range outputs: plain GEMMs only, plane range on a 32-column boundary
  const bool f8 = RSP_PLANE_IS_F8(d.a_scale_log2);   // A and W "lo" planes are cat8 planes (the caller pairs them)
  if (f8 && (d.conv_k != 0 || d.N <= 64)) return RSP_EINVAL;
  if (RSP_PLANE_IS_F8(d.c_scale_log2) && d.Chi && ((d.N & 3) || d.ct_W > 0)) return RSP_EINVAL;
  auto nblk = [&](int bm, int bn) { return (long long)((d.N + bn - 1) / bn) * ((d.M + bm - 1) / bm); };
  // Tile rule (tools/gemm_sweep.py on MI355X; run-to-run spread is a few %): the register-pipelined loops win
  // everywhere; 256x256 needs >= 4 rounds of blocks over the 256 CUs, 256x128 >= 2, else 128x128 (2 blocks/CU).
  if (d.conv_k != 0) {   // implicit-GEMM convolutions: CONV instantiations of the same kernels
    if (d.N > 128 && ((d.tile_hint & 0xff) == 17 || ((d.tile_hint & 0xff) <= 1 && nblk(256, 256) >= 1024)))
      return launch_dma<256, 256, 2, 4, 2, 0, 0, 2, true>(d, s);
    if (d.N > 64) return launch_dma<128, 128, 2, 2, 2, 0, 0, 1, true>(d, s);
    if (d.N > 32) return launch_dma<128, 64, 2, 2, 3, 0, 0, 0, true>(d, s);
    return launch_dma<128, 32, 4, 1, 3, 0, 0, 0, true>(d, s);
  }
  int tile = d.tile_hint & 0xff;
  if (tile <= 1) {
    // short K (<= 8 K tiles): the block is mostly prologue + epilogue, two 128x128 blocks per CU overlap them
    // (round 2, tools/gemm_sweep.py on the SAM-decoder shapes M = 3.3 M rows: with N = 256 the 256x256 tile reads every A
    // row once instead of twice and is 7-14 % faster even at 4-8 K tiles; N = 128 stays with 128x128)
    if (d.K <= 256) tile = (d.N > 128 && nblk(256, 256) >= 1024) ? 17 : 14;
    else {
      // cost = rounds over the 256 CUs x tile area / relative efficiency (tools/gemm_sweep.py: 256x256 1.0,
      // 256x128 0.96, 128x128 0.88 with its two blocks per CU sharing the matrix pipe): the partially filled last
      // round is what separates the candidates (e.g. M = 39200, N = 1280: 7 rounds of 256x128 vs 6 of 128x128)
      auto cost = [&](int bm, int bn, int bpc, double eff) {
        const long long nb = nblk(bm, bn), slots = 256LL * bpc;
        return (double)((nb + slots - 1) / slots) * bm * bn * bpc / eff;
      };
      double best = cost(128, 128, 2, 0.88);
      tile = 14;
      if (d.N > 64) { const double c = cost(256, 128, 1, 0.96); if (c < best) { best = c; tile = 18; } }
      if (d.N > 128) { const double c = cost(256, 256, 1, 1.0); if (c <= best) { best = c; tile = 17; } }
      // round 2 (tools/gemm_exp.py, ViT-H shapes): from two full rounds of 256x256 tiles on, the big tile wins even
      // with a ragged last round (blocks do not run in lockstep): qkv 350 vs 322 TFLOP/s, lin2 368 vs 352
      // (K = 1280 with only 3 rounds -- the proj shape -- stays with the small tile: 279 vs 252)
      if (d.N > 128 && (nblk(256, 256) >= 1024 || (nblk(256, 256) >= 512 && d.K >= 2048))) tile = 17;
    }
  }
  if (f8) {   // fp8-corrected product: the three tiles the rule above picks
    // with a third less matrix time per tile the big tile already pays from two rounds of blocks on, also at K = 1280
    // (proj shape, tools/gemm_f8_exp.py: 337 vs 301 TFLOP/s)
    if ((d.tile_hint & 0xff) <= 1 && d.N > 128 && nblk(256, 256) >= 512) tile = 17;
    if (tile == 17 && d.N > 128) return launch_dma<256, 256, 2, 4, 2, 0, 0, 2, false, 0, true>(d, s);
    if ((tile == 18 || tile == 17) && d.N > 64) return launch_dma<256, 128, 4, 2, 2, 0, 0, 2, false, 0, true>(d, s);
    return launch_dma<128, 128, 2, 2, 2, 0, 0, 1, false, 0, true>(d, s);
  }
  switch (tile) {   // hints >= 4 are benchmarking variants of the same arithmetic
    case 3: if (d.N > 128) return launch_dma<256, 256, 2, 4, 2>(d, s); break;
    case 2: if (d.N > 64) return launch_dma<256, 128, 4, 2, 3>(d, s); break;
    case 4: if (d.N > 64) return launch_dma<256, 128, 4, 2, 2>(d, s); break;
    case 5: if (d.N > 64) return launch_dma<128, 128, 2, 2, 4>(d, s); break;
    case 6: if (d.N > 64) return launch_dma<128, 128, 2, 2, 3>(d, s); break;
#ifdef RSP_S2_ABLATIONS   /* development builds only (RSP_DEV_BUILD=1): these compute WRONG results on purpose */
    case 7: if (d.N > 64) return launch_dma<128, 128, 2, 2, 2, 1>(d, s); break;   // ablation: no DMA
    case 8: if (d.N > 64) return launch_dma<128, 128, 2, 2, 2, 2>(d, s); break;   // ablation: no MFMA
    case 9: if (d.N > 128) return launch_dma<256, 256, 2, 4, 2, 1>(d, s); break;
    case 10: if (d.N > 128) return launch_dma<256, 256, 2, 4, 2, 2>(d, s); break;
    case 15: if (d.N > 128) return launch_dma<256, 256, 2, 4, 2, 1, 0, 1>(d, s); break;   // ... without the DMA
    case 16: if (d.N > 64) return launch_dma<256, 128, 4, 2, 2, 1, 0, 1>(d, s); break;
#else
    case 7: case 8: case 9: case 10: case 15: case 16: return RSP_EINVAL;
#endif
    case 11: if (d.N > 128) return launch_dma<256, 256, 2, 4, 2, 0, 0, 1>(d, s); break;   // register-pipelined loops
    case 12: if (d.N > 64) return launch_dma<256, 128, 4, 2, 2, 0, 0, 1>(d, s); break;
    case 13: if (d.N > 64) return launch_dma<256, 128, 4, 2, 3, 0, 0, 1>(d, s); break;
    case 14: if (d.N > 64) return launch_dma<128, 128, 2, 2, 2, 0, 0, 1>(d, s); break;
    case 17: if (d.N > 128) return launch_dma<256, 256, 2, 4, 2, 0, 0, 2>(d, s); break;   // ... with the DMA spread out
    case 18: if (d.N > 64) return launch_dma<256, 128, 4, 2, 2, 0, 0, 2>(d, s); break;
    case 19: if (d.N > 64) return launch_dma<256, 128, 4, 2, 3, 0, 0, 2>(d, s); break;
    case 20: if (d.N > 64) return launch_dma<128, 128, 2, 2, 2, 0, 0, 2>(d, s); break;
    case 21: return launch_dma<128, 64, 2, 2, 3, 0, 0, 1>(d, s);                         // narrow tile, 3-deep ring
    case 22: return launch_dma<128, 64, 2, 2, 4, 0, 0, 1>(d, s);
    case 31: if (d.N > 128) return launch_dma<256, 256, 2, 4, 2, 0, 0, 2, false, 1>(d, s); break;   // pass-major MFMA order
    case 32: if (d.N > 64) return launch_dma<256, 128, 4, 2, 2, 0, 0, 2, false, 1>(d, s); break;
    default: break;
  }
  if (d.N > 64) return launch_dma<128, 128, 2, 2, 2, 0, 0, 1>(d, s);
  if (d.N > 32) return launch_dma<128, 64, 2, 2, 3>(d, s);
  return launch_dma<128, 32, 4, 1, 3>(d, s);
}

// which kernel rsp_gemm runs for a descriptor of the plane path (profiler labels, tools): 1 = gemm_s2 (two blocks per
// CU, 256 x 128), 0 = one of this file's tiles
int rsp_gemm_s2_epilogue_of(const RspGemmDesc& d);                     // gemm_s2.hip
extern "C" int rsp_gemm_s2_epilogue(const RspGemmDesc* d) {
  if (!d || !rsp_gemm_uses_s2(d)) return -1;
  const int h = d->tile_hint & 0xff;
  if (h >= 40 + 64) return 64;                                         // hint 104: the run-time epilogue, forced
  return rsp_gemm_s2_epilogue_of(*d);
}

extern "C" int rsp_gemm_uses_s2(const RspGemmDesc* d) {
  if (!d || !(d->Ahi && d->Alo)) return 0;
  const int h = d->tile_hint & 0xff;
  if (h >= 40 && h < 168) return 1;
  return h == 0 && !rsp_gemm_pp_auto(*d) && rsp_gemm_s2_auto(*d);
}

extern "C" int rsp_gemm_uses_pp(const RspGemmDesc* d) {
  if (!d || !(d->Ahi && d->Alo)) return 0;
  const int h = d->tile_hint & 0xff;
  if (h >= 200) return rsp_gemm_pp_eligible(*d) ? (((h - 200) & 1) ? 128 : 256) : 0;
  if (h != 0) return 0;
  return rsp_gemm_pp_auto(*d);
}
