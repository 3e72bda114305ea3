// SAM ViT GLOBAL-layer attention, "plane-fed" form (round 2; the 14 x 14 windows moved to attn_win.hip in round 4, which
// shares the operand layouts described here): K and V are consumed as the fp16 hi / lo PLANES the qkv GEMM's epilogue
// already wrote (KB32 layout [col/32][row][32], rsp_gemm Chi/Clo with pl_col0 = D) -- no fp32 K/V tensor, no split pass,
// no transposed copy of V.  Reference semantics: SamVisionAttention.forward HF:803-831 + get_decomposed_rel_pos
// HF:761-801 (vit_sam.py:117-157, 202-221); windows of 14x14 (HF:900-952) and the global layers (S = 64 / 32).
//
//   * a block = NW waves x 32 queries of one (image | window, head); keys stream through an NBUF-deep LDS ring in tiles
//     of KT keys, copied HBM/L2 -> LDS by the DMA engine (global_load_lds_dwordx4, 16 B = 8 d-values of one key per lane);
//     one s_barrier per tile; the DMA of tile t + NBUF - 1 is in flight while tile t is multiplied.
//   * S^T = K Q^T (fp16x3 MFMA 32x32x16): the A operand is a ds_read_b128 of a K row (row pitch 176 B for dh = 80, XOR
//     swizzle for dh = 64: conflict-free); every lane owns ONE query column -> fp32 online softmax without cross-lane
//     traffic (one shuffle across the two half waves per tile), decomposed rel-pos bias from registers.
//   * O^T += V^T P^T (fp16x3): V stays ROW-major [key][d] in LDS exactly as the DMA delivers it; the V^T fragment
//     (lane = d, 4 consecutive keys) comes from ds_read_b64_tr_b16, gfx950's 4x4-transposing LDS read: inside a 16-lane
//     group lane i supplies the address of 4 contiguous halves D_i[0..3] and lane l receives D_{4j + l/4}[l % 4], j = 0..3
//     (tools/probes/tr_probe.hip, measured).  Lane i points at key k0 + i/4, d = d0 + 4 (i % 4): lane l then holds keys
//     k0..k0+3 of column d0 + l -- the A operand layout.  V rows are pitched to 192 B (zero-page DMA chunks behind the dh
//     real ones): the four rows of a read start 48 banks apart -> conflict-free, and d >= dh reads zeros.
//   * P never leaves registers: accumulator registers [8 (s & 1), +8) of score block s >> 1 are the B operand of k-step s
//     (keys 16 s + 4 hh + {0..3} and 16 s + 8 + 4 hh + {0..3}), matched by two transposing reads per plane.
//   * the 14 x 14 windows: attn_win.hip (round 4).
//   * global layers (KT = 64, 8 waves): same structure as attn_global.hip minus the split pass and its HBM round trip.
//   * What bounds it (round 2, measured): VALU ISSUE.  The loop carries 6.3 (global) / 10.9 (windows) VALU instructions
//     per MFMA -- bias fma, max, exponent shift, exp2, row sum, the 3-instruction P split, accumulator rescale -- and a
//     SIMD fits about 5 beside a 32-cycle MFMA.  Three restructurings that do NOT touch that count changed nothing:
//     register-pipelined LDS fragment reads (kept), a software-pipelined tile loop with QK^T of tile t + 1 interleaved
//     into the softmax of tile t by sched_group_barrier (ISA showed the interleave; 3.26 vs 3.13 ms, removed), and
//     de-phasing the two waves of a SIMD (no change).  The remaining lever is fewer VALU instructions per score.
//     Round 5 (after the compute / load alternation took the GEMM's K loop to the matrix pipe's own pace, gemm_pp.hip): the
//     same alternation here -- waves 0-3 / 4-7 of a block pairwise on the SIMDs, one in a pure matrix phase (PV of tile
//     t - 1, then QK^T of tile t) while its partner runs softmax, P split and DMA issue, s_barrier between phases, separate
//     two-deep K and V rings -- was built, verified on the emulator and measured on the MI355X: bit-equal, 2.04 vs 2.09 ms
//     (ViT-H, 4096 tokens), and with the step's MFMAs issued pass-major over independent accumulators 2.22 vs 2.19 ms.
//     At 313-337 TFLOP/s the kernel keeps the matrix pipe ~60 % busy -- the class of the GEMM (73 %) -- so neither wave
//     ordering nor dependent-MFMA spacing is what is left here; removed (gpurun_out/r5).
#include <stdlib.h>
#include <type_traits>
#include "rsp_common.h"

namespace {

constexpr int EQ = 6;                         // q * scale * 2^EQ before its fp16 split (k / v planes carry their own)
constexpr float P_SCALE_LOG2 = 14.0f;         // probabilities are scaled by 2^14 before the fp16 split
constexpr float LOG2E_C = 1.4426950408889634f;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef short v4s __attribute__((ext_vector_type(4)));

__device__ uint4 g_zero16s[4];                // zero page: padded V chunks, keys beyond the window
// dh = 80 leaves V^T rows 80..95 of the third 32-row block unused: the chunk behind the real ones carries 1.0 at d = 80
// and d = 84 (hi plane), so accumulator register 8 of that block is sum_k P[k] for the lane's query in BOTH half waves --
// the softmax denominator comes out of the PV MFMAs (rescaled with the rest) instead of 16-32 VALU adds per tile.
__device__ const _Float16 g_ones16s[8] = {(_Float16)1.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f,
                                          (_Float16)1.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};

__device__ __forceinline__ void split8s(const float* x, half8_t& hi, half8_t& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    half_t h, l;
    rsp_split1(x[i], h, l);
    hi[i] = h; lo[i] = l;
  }
}

// Split of values known to be inside the fp16 range (probabilities * 2^14, scaled q): no saturation needed, and the hi
// part may be TRUNCATED -- the remainder is then non-negative and lo = rtz(x - hi) still carries the next 11 bits, so
// hi + lo holds ~21 bits, the same class as the round-to-nearest pair.  v_cvt_pkrtz_f16_f32 converts two values per
// instruction: 2 VALU ops per element (round 5; 3 with the back conversion of rounds 2-4) instead of 7 (the attention
// kernels are VALU-issue bound, PMC r2).
__device__ __forceinline__ void split8_fast(const float* x, half8_t& hi, half8_t& lo) {
  float m1 = -1.0f;
  asm volatile("" : "+v"(m1));
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    const half2_t h2 = __builtin_bit_cast(half2_t, __builtin_amdgcn_cvt_pkrtz(x[i], x[i + 1]));
    // x - hi, exact, as ONE v_fma_mix_f32 each (the fp16 half is a source of the fp32 fma: no v_cvt_f32_f16 in front)
    // (the multiplier sits in a register the optimiser cannot see through: written as x - (float)h the compiler converts first)
    const float r0 = __builtin_fmaf((float)h2[0], m1, x[i]), r1 = __builtin_fmaf((float)h2[1], m1, x[i + 1]);
    const half2_t l2 = __builtin_bit_cast(half2_t, __builtin_amdgcn_cvt_pkrtz(r0, r1));
    hi[i] = h2[0]; hi[i + 1] = h2[1]; lo[i] = l2[0]; lo[i + 1] = l2[1];
  }
}

__device__ __forceinline__ unsigned xcd_contig_s(unsigned bid, unsigned nblk) {
  const unsigned q = nblk >> 3, r = nblk & 7u;
  const unsigned xcd = bid & 7u, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for_s(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for_s<I + 1, N>(f);
  }
}

struct AttnSP {
  const float* q; int64_t q_ld;                 // fp32 [Bp*T, q_ld]: head h at columns [h*DH, (h+1)*DH)
  const half_t* kv_hi; const half_t* kv_lo;     // KB32 planes of the [Bp*T, 2 D] matrix (K | V), value * 2^kv_e
  int64_t kv_rows;                              // rows of the plane tensors (>= Bp*T)
  int kv_e;
  const float* rel;                             // [Bp*nh, T, 2 S]
  float* out; half_t* out_hi; half_t* out_lo; float out_pscale; int64_t out_rows;
  bool out_f8;                                  // output planes in the cat8 format (plane format word)
  int T, S, nh, D;
  float scale;
};

template <int DH, int NW, int KT, int NBUF>
__global__ __launch_bounds__(NW * 64) void attn_stream_kernel(const AttnSP p) {
  constexpr int NT = NW * 64;
  constexpr int DSTEPS = DH / 16;
  constexpr int DBLK = (DH + 31) / 32;
  constexpr int NBLK = KT / 32;                       // 32-key score blocks per tile
  constexpr int KCH = DH / 8;                         // real 16-byte chunks per K / V row
  constexpr bool KSWZ = (KCH == 8);                   // dh = 64: 128-byte K rows with an XOR swizzle
  constexpr int KCPR = KSWZ ? 8 : KCH + 1;            // dh = 80: 11 chunks = 176-byte pitch
  constexpr int VCPR = 12;                            // V rows: 192-byte pitch
  constexpr int K_UNITS = KT * KCPR, V_UNITS = KT * VCPR;
  constexpr int TILE_UNITS = 2 * K_UNITS + 2 * V_UNITS;
  constexpr int NDMA = (TILE_UNITS + NT - 1) / NT;
  constexpr int BUF_BYTES = (TILE_UNITS * 16 + 1023) / 1024 * 1024;   // lanes past the image are masked off the DMA
  constexpr bool LSUM_MFMA = (DH % 32) != 0;          // spare V^T rows exist: row sums through the MFMA (see g_ones16s)
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NBUF][BUF_BYTES];
  // rel_h of every key row for this wave's 32 queries, log2 domain: [wave][key row][query] (round 6).  Rounds 2-5 fetched the
  // next tile's value from global memory inside the tile loop, "a whole tile ahead of its use" -- but hipcc waits for a load
  // that crosses the loop's back edge at once, with s_waitcnt vmcnt(0): behind the QK^T products of every tile the wave drained
  // the DMA burst of the next tile it had issued at the top of the tile.
  __shared__ float sRelH[NW][64][32];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, l31 = lane & 31;
  const int T = p.T, S = p.S, nh = p.nh;
  const int QB = NW * 32;
  const int nqb = T / QB;
  const unsigned lb = xcd_contig_s(blockIdx.x, gridDim.x);
  const int bp = (int)(lb / (unsigned)(nqb * nh));
  const int h = (int)(lb / (unsigned)nqb) - bp * nh;
  const int q0 = (int)(lb % (unsigned)nqb) * QB;
  const int q = q0 + wave * 32 + l31;
  const bool qv = q < T;
  const float* rel_b = p.rel + ((int64_t)bp * nh + h) * T * (2 * S);
  const int64_t row0 = (int64_t)bp * T;               // first row of this image / window in q, planes, out

  // ---- per-thread DMA slots: unit u = i*NT + tid of the tile image [K_hi | K_lo | V_hi | V_lo] ----
  const unsigned char* dsrc[NDMA];
  int drow[NDMA];                                     // key row inside the tile (validity of the window's last tile)
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(g_zero16s);
#pragma unroll
  for (int i = 0; i < NDMA; ++i) {
    const int u = i * NT + tid;
    dsrc[i] = zero;
    drow[i] = -1;                                     // -1: always the zero page (padding chunk / beyond the image)
    int pl, row, c, colbase;
    bool real = false;
    if (u < 2 * K_UNITS) {
      pl = u / K_UNITS;
      const int v = u - pl * K_UNITS;
      row = v / KCPR;
      const int pc = v - row * KCPR;
      c = KSWZ ? (pc ^ ((row >> 1) & 7)) : pc;
      colbase = 0;
      real = c < KCH;
    } else if (u < TILE_UNITS) {
      const int w = u - 2 * K_UNITS;
      pl = w / V_UNITS;
      const int v = w - pl * V_UNITS;
      row = v / VCPR;
      c = v - row * VCPR;
      colbase = p.D;
      real = c < KCH;
    }
    if (LSUM_MFMA && u >= 2 * K_UNITS && u < 2 * K_UNITS + V_UNITS && !real && c == KCH) {
      dsrc[i] = reinterpret_cast<const unsigned char*>(g_ones16s);      // hi plane, first padding chunk: the ones
      drow[i] = -2;                                                     // constant source, valid for every tile
    }
    if (real) {
      const int col = colbase + h * DH + c * 8;
      const half_t* base = pl == 0 ? p.kv_hi : p.kv_lo;
      dsrc[i] = reinterpret_cast<const unsigned char*>(base + ((int64_t)(col >> 5) * p.kv_rows + row0 + row) * 32 + (col & 31));
      drow[i] = row;
    }
  }
  const rsp_lds_addr_t smem_a = rsp_lds_addr((lptr_t)&smem[0][0]);
  auto issue_tile = [&](int kt, int buf) {
    const rsp_lds_addr_t lbase = smem_a + buf * BUF_BYTES;
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      const bool ok = drow[i] >= 0;
      const unsigned char* src = ok ? dsrc[i] + (int64_t)kt * (KT * 64) : (drow[i] == -2 ? dsrc[i] : zero);
      if ((i + 1) * NT <= TILE_UNITS || i * NT + tid < TILE_UNITS)     // the last instruction may be partly masked
        RSP_GLOBAL_LOAD_LDS_B128(src, lbase + (i * NT + wave * 64) * 16);      // (inline assembly: csrc/rsp_common.h)
    }
  };

  const int nt = T / KT;
#pragma unroll
  for (int t = 0; t < NBUF - 1; ++t)
    if (t < nt) issue_tile(t, t);

  // ---- Q fragments (B operand of S^T = K Q^T), scaled and split once ----
  half8_t qh[DSTEPS], qlo[DSTEPS];
  {
    const float qs = p.scale * ldexpf(1.0f, EQ);
    const float* q_b = p.q + (row0 + (qv ? q : 0)) * p.q_ld + (int64_t)h * DH;
#pragma unroll
    for (int st = 0; st < DSTEPS; ++st) {
      float x[8];
      const f32x4 a = *reinterpret_cast<const f32x4*>(q_b + st * 16 + hh * 8);
      const f32x4 b = *reinterpret_cast<const f32x4*>(q_b + st * 16 + hh * 8 + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) { x[i] = qv ? a[i] * qs : 0.f; x[4 + i] = qv ? b[i] * qs : 0.f; }
      split8s(x, qh[st], qlo[st]);
    }
  }
  // ---- rel-pos bias in the log2 domain ----
  // rel_w of this lane's key columns (tile invariant) + one rel_h per key row
  f32x16 bw[2];                                       // rel_w of key columns 0..31 and 32..63 (mod S) for this lane
  const float* rq = rel_b + (int64_t)(qv ? q : 0) * (2 * S);
#pragma unroll
  for (int blk = 0; blk < 2; ++blk)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kl = 32 * blk + (r & 3) + 8 * (r >> 2) + 4 * hh;
      bw[blk][r] = qv ? rq[S + (kl % S)] * LOG2E_C : 0.f;
    }
  // rel_h of all S key rows -> LDS: the half waves split the rows (hh = 0: rows 0 .. S/2 - 1); read back per tile by both
  {
    const int k0 = hh * (S >> 1);
    for (int k = 0; k < (S >> 1); k += 4) {
      const f32x4 v = qv ? *reinterpret_cast<const f32x4*>(rq + k0 + k) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) sRelH[wave][k0 + k + e][l31] = v[e] * LOG2E_C;
    }
  }
  auto load_bh = [&](int kt, float& b0, float& b1) {    // key row(s) of the tile's two 32-key blocks (behind the tile's barrier)
    const int kh0 = (kt * KT) / S, kh1 = (kt * KT + 32) / S;
    b0 = sRelH[wave][kh0][l31];
    b1 = KT / S > 1 ? sRelH[wave][kh1][l31] : b0;       // S = 32: two key rows per tile
  };

  f32x16 acc_o[DBLK];
#pragma unroll
  for (int db = 0; db < DBLK; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[db][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float s_unscale2 = ldexpf(1.0f, -(EQ + p.kv_e)) * LOG2E_C;
  f32x16 s_unscale16;
#pragma unroll
  for (int r = 0; r < 16; ++r) s_unscale16[r] = s_unscale2;

  // per-lane byte offsets of the transposing V reads (see header): row = 4 hh + (i >> 2), d = 16 g16 + 4 (i & 3)
  const int li = lane & 15, g16 = (lane >> 4) & 1;
  const int v_lane_off = (4 * hh + (li >> 2)) * (VCPR * 16) + (16 * g16 + 4 * (li & 3)) * 2;

  auto tile_body = [&](int kt, int buf) {
    // this wave's part of tile kt must have landed; the (up to NBUF - 2) younger tiles may stay in flight.  vmcnt retires
    // in order, so the count also covers the few register loads (rel_h) issued since: conservative, never too weak.
    {
      const int ahead = min(NBUF - 2, nt - 1 - kt);
      if (NBUF >= 4 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");
      else if (NBUF >= 3 && ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                        // ... everybody's part; the buffer of tile kt - 1 is drained
    if (kt + NBUF - 1 < nt) {
      int nb = buf + NBUF - 1;
      if (nb >= NBUF) nb -= NBUF;
      issue_tile(kt + NBUF - 1, nb);
    }
    const unsigned char* sb = &smem[buf][0];
    const half_t* sK0 = reinterpret_cast<const half_t*>(sb);
    const half_t* sK1 = reinterpret_cast<const half_t*>(sb + K_UNITS * 16);
    const unsigned char* sV0 = sb + 2 * K_UNITS * 16;
    const unsigned char* sV1 = sb + (2 * K_UNITS + V_UNITS) * 16;

    // ---- S^T = K Q^T, register-pipelined: the K fragments of step i + 1 are read from LDS before the three MFMAs of
    // step i are issued (with 1-2 waves per SIMD nothing else hides the ~130-cycle LDS latency; PMC r2: 39 % issue stalls)
    f32x16 sc[NBLK];
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk)
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[blk][r] = 0.f;
    {
      half8_t kfh[2], kfl[2];
      auto kread = [&](int blk, int st, half8_t& h8, half8_t& l8) {
        const int row = blk * 32 + l31;
        const int c = st * 2 + hh;
        const int off = row * (KCPR * 8) + ((KSWZ ? (c ^ ((row >> 1) & 7)) : c) << 3);     // halves
        h8 = *reinterpret_cast<const half8_t*>(sK0 + off);
        l8 = *reinterpret_cast<const half8_t*>(sK1 + off);
      };
      kread(0, 0, kfh[0], kfl[0]);
      static_for_s<0, NBLK * DSTEPS>([&](auto ic) {
        constexpr int i = decltype(ic)::value, blk = i / DSTEPS, st = i % DSTEPS, cur = i & 1;
        if constexpr (i + 1 < NBLK * DSTEPS) kread((i + 1) / DSTEPS, (i + 1) % DSTEPS, kfh[cur ^ 1], kfl[cur ^ 1]);
        __builtin_amdgcn_sched_barrier(0);        // keep the next step's reads AHEAD of this step's matrix work
        sc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfl[cur], qh[st], sc[blk], 0, 0, 0);
        sc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[cur], qlo[st], sc[blk], 0, 0, 0);
        sc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[cur], qh[st], sc[blk], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      });
    }

    // ---- bias + online softmax in the log2 domain (per-lane query column) ----
    float tmax = -INFINITY;
    float k0s[NBLK];
    {
      float bh0, bh1;
      load_bh(kt, bh0, bh1);
#pragma unroll
      for (int blk = 0; blk < NBLK; ++blk) {
        float tm = -INFINITY;
        sc[blk] = __builtin_elementwise_fma(sc[blk], s_unscale16, bw[blk]);      // 8 v_pk_fma_f32
#pragma unroll
        for (int r = 0; r < 16; ++r) tm = fmaxf(tm, sc[blk][r]);
        k0s[blk] = blk == 0 ? bh0 : bh1;                 // the key row's rel_h enters as one scalar per block
        tmax = fmaxf(tmax, tm + k0s[blk]);
      }
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);              // finite: every tile has a valid key for hh == 0
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
      const float kk = k0s[blk] - m_new + P_SCALE_LOG2;
      sc[blk] = sc[blk] + kk;                                                    // 8 v_pk_add_f32
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(sc[blk][r]);
        sc[blk][r] = pv;
        if constexpr (!LSUM_MFMA) psum += pv;
      }
    }
    if constexpr (!LSUM_MFMA) l_run = l_run * alpha + psum;
    m_run = m_new;
    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {   // no lane saw a new maximum: skip the 16 DBLK multiplies
#pragma unroll
      for (int db = 0; db < DBLK; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[db][r] *= alpha;
    }

    // ---- O^T += V^T P^T : V^T fragments through the transposing LDS read, register-pipelined like the K reads:
    // the four reads of step (s, db) + 1 are issued before the MFMAs of step (s, db); the P split of 16-key step s is
    // VALU work that runs while the first reads of that step are in flight
    {
      typedef __attribute__((address_space(3))) v4s* lv4;
      v4s va[2][4];                                        // [buffer][hi k0..3, hi k8..11, lo k0..3, lo k8..11]
      auto vread = [&](int s_, int db, v4s* f) {
        const int off = v_lane_off + (16 * s_) * (VCPR * 16) + db * 64;      // bytes: key 16 s (+ 8), d block db
        f[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lv4)(sV0 + off));
        f[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lv4)(sV0 + off + 8 * (VCPR * 16)));
        f[2] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lv4)(sV1 + off));
        f[3] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lv4)(sV1 + off + 8 * (VCPR * 16)));
      };
      constexpr int NS = KT / 16, NSTEP = NS * DBLK;
      vread(0, 0, va[0]);
      half8_t ph, pl;
      static_for_s<0, NSTEP>([&](auto ic) {
        constexpr int i = decltype(ic)::value, s_ = i / DBLK, db = i % DBLK, cur = i & 1;
        if constexpr (db == 0) {
          float pf[8];
#pragma unroll
          for (int t = 0; t < 8; ++t) pf[t] = sc[s_ >> 1][8 * (s_ & 1) + t];
          split8_fast(pf, ph, pl);
        }
        if constexpr (i + 1 < NSTEP) vread((i + 1) / DBLK, (i + 1) % DBLK, va[cur ^ 1]);
        __builtin_amdgcn_sched_barrier(0);
        union { v4s s4[2]; half8_t h8; } uh, ul;
        uh.s4[0] = va[cur][0]; uh.s4[1] = va[cur][1]; ul.s4[0] = va[cur][2]; ul.s4[1] = va[cur][3];
        acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ul.h8, ph, acc_o[db], 0, 0, 0);
        acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(uh.h8, pl, acc_o[db], 0, 0, 0);
        acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(uh.h8, ph, acc_o[db], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      });
    }
  };

  {
    int buf = 0;
    for (int kt = 0; kt < nt; ++kt) {
      tile_body(kt, buf);
      if (++buf == NBUF) buf = 0;
    }
  }

  // ---- normalise and store: lane holds O[q][d], d = db*32 + (r&3) + 8*(r>>2) + 4*hh ----
  // denominator: per-lane partial sums (both half waves hold half of the keys) or the MFMA row sum (all keys already)
  const float l_tot = LSUM_MFMA ? acc_o[DBLK - 1][8] : l_run + __shfl_xor(l_run, 32, 64);
  if (qv) {
    const float inv = ldexpf(1.0f, -p.kv_e) / l_tot;
    float* dst = p.out ? p.out + (row0 + q) * p.D + (int64_t)h * DH : nullptr;
#pragma unroll
    for (int db = 0; db < DBLK; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = db * 32 + 8 * g + 4 * hh;
        if (d0 < DH) {
          f32x4 o;
#pragma unroll
          for (int c = 0; c < 4; ++c) o[c] = acc_o[db][4 * g + c] * inv;
          if (dst) *reinterpret_cast<f32x4*>(dst + d0) = o;
          if (p.out_hi) {
            const int col = h * DH + d0;
            const int64_t eo = ((int64_t)(col >> 5) * p.out_rows + (row0 + q)) * 32 + (col & 31);
            rsp_store_planes4(p.out_hi, p.out_lo, eo, o * p.out_pscale, p.out_f8);
          }
        }
      }
  }
}

// (round 5: a three-deep ring, <DH, NW, 64, 3> = 141 KB of LDS at dh = 80, measured 8.89-8.93 against 8.97 ms for the four
//  ViT-H layers on one box: DMA distance is not what the waves wait for; the two-deep ring stays)
template <int DH>
int launch_stream(const AttnSP& p, int Bp, hipStream_t s) {
  constexpr int NW = 8;
  if (p.T % (NW * 32) || p.T % 64) return RSP_EINVAL;
  hipLaunchKernelGGL((attn_stream_kernel<DH, NW, 64, 2>), dim3((unsigned)(p.T / (NW * 32)) * p.nh * Bp),
                     dim3(NW * 64), 0, s, p);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

}  // namespace

// attn_win.hip: the 14 x 14 windows (rel-pos terms given as a tensor, or computed in the kernel from packed tables)
int rsp_attn_win_dispatch(const float* q, int64_t q_ld, const uint16_t* kv_hi, const uint16_t* kv_lo, int64_t kv_rows,
                          int32_t kv_scale_log2, const float* rel, const uint16_t* rel_tab, float* out, uint16_t* out_hi,
                          uint16_t* out_lo, int32_t out_scale_log2, int32_t Bp, int32_t nh, int32_t dh, float scale,
                          int32_t win_per_side, int32_t win_real_last, int32_t variant, hipStream_t s);

extern "C" int rsp_vit_attention_planes_ex(const float* q, int64_t q_ld, const uint16_t* kv_hi, const uint16_t* kv_lo,
                                           int64_t kv_rows, int32_t kv_scale_log2, const float* rel, float* out,
                                           uint16_t* out_hi, uint16_t* out_lo, int32_t out_scale_log2, int32_t Bp,
                                           int32_t S, int32_t nh, int32_t dh, float scale, int32_t win_per_side,
                                           int32_t win_real_last, rsp_stream_t stream) {
  if (!q || !kv_hi || !kv_lo || !rel || Bp <= 0 || nh <= 0) return RSP_EINVAL;
  if (!out && !(out_hi && out_lo)) return RSP_EINVAL;
  if ((out_hi == nullptr) != (out_lo == nullptr)) return RSP_EINVAL;
  if (!(S == 14 || S == 32 || S == 64)) return RSP_EINVAL;
  if (S == 14)
    return rsp_attn_win_dispatch(q, q_ld, kv_hi, kv_lo, kv_rows, kv_scale_log2, rel, nullptr, out, out_hi, out_lo,
                                 out_scale_log2, Bp, nh, dh, scale, win_per_side, win_real_last, 0, (hipStream_t)stream);
  const int D = nh * dh;
  if ((D & 31) || (q_ld & 3) || kv_rows < (int64_t)Bp * S * S) return RSP_EINVAL;
  if (RSP_PLANE_IS_F8(kv_scale_log2)) return RSP_EINVAL;   // K | V are consumed as fp16 hi / lo planes
  AttnSP p;
  p.q = q; p.q_ld = q_ld; p.kv_hi = reinterpret_cast<const half_t*>(kv_hi); p.kv_lo = reinterpret_cast<const half_t*>(kv_lo);
  p.kv_rows = kv_rows; p.kv_e = RSP_PLANE_EXP(kv_scale_log2); p.rel = rel; p.out = out;
  p.out_hi = reinterpret_cast<half_t*>(out_hi); p.out_lo = reinterpret_cast<half_t*>(out_lo);
  p.out_pscale = ldexpf(1.0f, RSP_PLANE_EXP(out_scale_log2)); p.out_f8 = out_hi && RSP_PLANE_IS_F8(out_scale_log2);
  p.out_rows = (int64_t)Bp * S * S;
  p.T = S * S; p.S = S; p.nh = nh; p.D = D; p.scale = scale;
  hipStream_t s = (hipStream_t)stream;
  if (dh == 64) return launch_stream<64>(p, Bp, s);
  if (dh == 80) return launch_stream<80>(p, Bp, s);
  return RSP_EINVAL;
}

extern "C" int rsp_vit_attention_planes(const float* q, int64_t q_ld, const uint16_t* kv_hi, const uint16_t* kv_lo,
                                        int64_t kv_rows, int32_t kv_scale_log2, const float* rel, float* out,
                                        uint16_t* out_hi, uint16_t* out_lo, int32_t out_scale_log2, int32_t Bp,
                                        int32_t S, int32_t nh, int32_t dh, float scale, rsp_stream_t stream) {
  return rsp_vit_attention_planes_ex(q, q_ld, kv_hi, kv_lo, kv_rows, kv_scale_log2, rel, out, out_hi, out_lo, out_scale_log2,
                                     Bp, S, nh, dh, scale, 0, 0, stream);
}
