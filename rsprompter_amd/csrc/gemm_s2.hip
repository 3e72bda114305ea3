// hipcc-flags: -mllvm -amdgpu-atomic-optimizer-strategy=None
// fp16x3 GEMM, "two blocks per CU" persistent form (round 3).  Same arithmetic and the same plane operands as
// gemm_dma.hip (a_lo*b_hi + a_hi*b_lo + a_hi*b_hi per 16 k, fp32 accumulate in v_mfma_f32_32x32x16_f16; results are
// bit-identical to that kernel), other structure:
//
//   * 256-thread blocks (4 waves, ONE per SIMD), block tile 256 x 128, wave tile 128 x 64 (8 accumulators = 128 VGPRs),
//     72 KB of LDS -> TWO independent blocks per CU.  The 8-wave 256 x 256 kernel runs its two waves per SIMD in
//     lockstep through one barrier per K tile, the fragment-read burst behind it, its prologue and its epilogue: whenever
//     one wave of a SIMD stalls its partner stalls too (PMC r2: matrix pipe busy 58 %, and a K = 1280 tile spends ~18 % of
//     its time outside the main loop).  Two independent blocks share nothing but the CU: one block's barrier waits, DMA
//     issue, prologue latency and (VALU / store bound) epilogue run under the other block's matrix work.
//   * persistent: min(#tiles, 512) blocks walk the tile list; the first three ring stages of the NEXT tile are in flight
//     while the current tile's epilogue runs.
//   * ring of 3 stages x 16 k (24 KB each: A_hi | A_lo | B_hi | B_lo, 32-byte rows), fed by the DMA engine through
//     BUFFER loads (`buffer_load_dwordx4 ... offen lds`): the per-lane part of a source address is a 32-bit offset that is
//     constant for the whole tile (row * 64 + chunk * 16), the K position is a scalar offset, rows that do not exist
//     (tile overhang, padded window rows a_rowmap < 0) carry an offset beyond the descriptor's range and read as zeros.
//     No per-DMA vector ALU work at all (gemm_dma.hip: 5 VALU + 3 SALU per DMA instruction for the 64-bit address
//     select against a zero page).
//   * accumulators are kept TRANSPOSED (mfma(W fragment, A fragment): lane = output row, 4 consecutive registers =
//     4 consecutive output columns), so the epilogue needs no LDS staging and no block barrier: every lane owns rows,
//     loads / stores 16 bytes (fp32) or 8 bytes per plane at a time, and each wave finishes on its own.
//   * tile order: every XCD owns a contiguous range of the (grouped, M-fastest) tile list and its 64 resident blocks
//     work on 64 consecutive entries = an 8 x 8 patch of tiles sharing 8 A panels and 8 W panels in that XCD's L2.
//
// Scope: plain GEMMs of the plane path (row gather / scatter, bias, activation, fp32 or plane residual, fp32 and / or
// plane outputs incl. column ranges).  Convolutions, ConvTranspose / fused-LayerNorm / hyper-network epilogues and the
// fp8-corrected product stay with gemm_dma.hip.
#include <atomic>
#include <type_traits>
#include "rsp_common.h"

namespace {

constexpr int KS = 16;                      // k per ring stage
constexpr int BM = 256, BN = 128;           // block tile
constexpr int NTHR = 256;
constexpr int NS = 3;                       // ring depth
constexpr int A_PL = BM * 32, B_PL = BN * 32;            // bytes of one plane of one stage
constexpr int OFF_ALO = A_PL, OFF_BHI = 2 * A_PL, OFF_BLO = 2 * A_PL + B_PL;
constexpr int STAGE = 2 * A_PL + 2 * B_PL;               // 24576
constexpr int NDMA = STAGE / (NTHR * 16);                // 6 DMA instructions per wave and stage
constexpr int TM = 4, TN = 2;
constexpr unsigned OOB = 0x80000000u;                    // voffset of a row that does not exist: reads zeros

typedef __attribute__((address_space(3))) void* lptr_t;

struct S2P {
  RspGemmDesc d;
  FastDiv fd_resmod, fd_resb;
  int nbm, nbn, ntiles, per_xcd, group_m;
  unsigned* ticket;              // 9 words of this launch: next tile of XCD 0..7, blocks done
  unsigned long long* trace;     // tools only (VAR bit 5): per block and tile {start, loop end, epilogue end, hw id}
};
unsigned long long* g_s2_trace = nullptr;

// Tile tickets: the blocks of an XCD draw their tiles from that XCD's counter (one returning device-scope atomic per
// tile, ~1 us against a ~100 us tile) instead of a static stride -- co-resident blocks do not run at the same speed (the
// older wave of a SIMD wins the matrix pipe, the younger fills its gaps), a static split leaves the fast half idle at
// the end.  Every launch takes the next of TICKET_SLOTS slots (one 64-byte line each) of the CURRENT device's copy of
// g_s2_tickets; the last block to finish zeroes the slot again.  Host side (launch_s2): ONE process-wide atomic slot
// counter shared by all instantiations and host threads, the array's address looked up per device -- concurrent launches
// (streams, host threads, several devices in one process) never share ticket words unless more than TICKET_SLOTS GEMMs
// are in flight on one device at once.
constexpr int TICKET_SLOTS = 1024, TICKET_WORDS = 16;
__device__ unsigned g_s2_tickets[TICKET_SLOTS * TICKET_WORDS];
std::atomic<unsigned> g_s2_slot{0};
constexpr int S2_MAX_DEVICES = 64;
std::atomic<unsigned*> g_s2_ticket_base[S2_MAX_DEVICES];

unsigned* s2_ticket_base() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= S2_MAX_DEVICES) return nullptr;
  unsigned* b = g_s2_ticket_base[dev].load(std::memory_order_acquire);
  if (!b) {                       // (a race here looks the same address up twice)
    if (hipGetSymbolAddress(reinterpret_cast<void**>(&b), HIP_SYMBOL(g_s2_tickets)) != hipSuccess) return nullptr;
    g_s2_ticket_base[dev].store(b, std::memory_order_release);
  }
  return b;
}

// epilogue specialisations (EPI template argument): bit flags of what the tile's outputs need; E_GENERIC = everything at
// run time (all modes of the descriptor, slow: branches per 4 outputs)
constexpr int E_RES = 1, E_GELU = 2, E_C = 4, E_PL = 8, E_RMAP = 16, E_GENERIC = 64;

template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, N>(f);
  }
}

// VAR (experiment switches; 0 = product, the only value instantiated unless the library is built with
// -DRSP_S2_ABLATIONS -- the ablations compute WRONG results on purpose and do not belong into a shipped library): VAR == 1
// and VAR == 33 are the two scheduling experiments of round 4 (see `step`; profiles/r4_gemm_s2_priority_and_burst_experiments.txt:
// priority +5 % on proj, -2 % on lin2, 0 elsewhere; burst -3 ... -6 %: neither adopted); otherwise bit 0 = result stores with the non-temporal hint (was: raised wave priority in the epilogue -- no effect), bit 1 = epilogue without its stores (ablation), bit 2 = no DMA
// inside the K loop (ablation, garbage results), bit 3 = no epilogue (ablation), bit 4 = every DMA reads the first K block
// (cache-hot sources: separates memory latency from issue / LDS-write cost; garbage results), bit 5 = time stamps
template <int VAR, int EPI>
__global__ __launch_bounds__(NTHR, 2) void gemm_f16x3_s2_kernel(const S2P p) {
  // ONE LDS object (a second __shared__ variable makes hipcc drain the DMA queue before every fragment read): the ring
  // + one word for the block-wide broadcast of the next tile ticket
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * STAGE + 16];
  const RspGemmDesc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int hh = lane >> 5, l31 = lane & 31;
  const int M = d.M, N = d.N;
  const int nk = d.K / KS;

  const unsigned hw_id = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));     // HW_REG_HW_ID
  int trace_n = 0;

  // ---- buffer descriptors of the four operand planes (host checked: every plane < 2^31 bytes) ----
  const int a_kstr = d.a_rows * 64, b_kstr = (d.b_rows > 0 ? d.b_rows : N) * 64;   // bytes between K blocks of 32
  const int a_bytes = (d.K / 32) * a_kstr, b_bytes = (d.K / 32) * b_kstr;
  const __amdgpu_buffer_rsrc_t rAh = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(d.Ahi), 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rAl = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(d.Alo), 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rBh = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(d.Bhi), 0, b_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rBl = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(d.Blo), 0, b_bytes, 0x00020000);

  // ---- DMA lane constants: unit u = slot * 256 + tid of a stage image; inside a plane v = row * 2 + physical chunk.
  // A 32-byte-row image read with ds_read_b128 (16-lane groups {0-3, 12-15, 20-27}, ...) is conflict free when the
  // 16-byte chunk is XOR-swizzled with bit 3 of the row; the swizzle goes on the SOURCE offset (DMA writes lane-linear).
  const int d_row = tid >> 1;                                        // 0..127 (+128 for the second A slot)
  const int d_chunk = ((tid & 1) ^ ((d_row >> 3) & 1)) << 4;         // logical chunk * 16 held at this physical place

  // ---- fragment read offsets ----
  const int f_chunk = (hh ^ ((l31 >> 3) & 1)) << 4;
  const int a_lane = (wm * 128 + l31) * 32 + f_chunk;
  const int b_lane = OFF_BHI + (wn * 64 + l31) * 32 + f_chunk;

  struct Tile { int m0, n0; unsigned vA0, vA1, vB; };
  auto tile_setup = [&](int id) {
    Tile t;
    int mb = id / p.nbn, nb = id - mb * p.nbn;
    if (p.group_m > 1) {
      const int per = p.group_m * p.nbn;
      const int grp = id / per, rem = id - grp * per;
      const int first = grp * p.group_m;
      const int gsz = min(p.nbm - first, p.group_m);
      nb = rem / gsz;
      mb = first + (rem - nb * gsz);
    }
    t.m0 = mb * BM; t.n0 = nb * BN;
    auto arow = [&](int r) -> unsigned {
      const int gm = t.m0 + r;
      if (gm >= M) return OOB;
      const int srow = d.a_rowmap ? d.a_rowmap[gm] : gm;
      return srow < 0 ? OOB : (unsigned)srow * 64u + (unsigned)d_chunk;
    };
    t.vA0 = arow(d_row);
    t.vA1 = arow(d_row + 128);
    const int gn = t.n0 + d_row;
    t.vB = gn < N ? (unsigned)gn * 64u + (unsigned)d_chunk : OOB;
    return t;
  };
  // ONE DMA instruction (1 KiB) of stage s into the ring buffer at byte offset sb
  auto issue_slot = [&](auto ic, const Tile& t, int s, int sb) {
    constexpr int I = decltype(ic)::value;
    const int so = ((VAR & 16) ? 0 : (s >> 1) * (I < 4 ? a_kstr : b_kstr)) + (s & 1) * 32;
    unsigned char* l = smem + sb + I * 4096 + wave * 1024;
    if constexpr (I == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rAh, (lptr_t)l, 16, (int)t.vA0, so, 0, 0);
    if constexpr (I == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rAh, (lptr_t)l, 16, (int)t.vA1, so, 0, 0);
    if constexpr (I == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rAl, (lptr_t)l, 16, (int)t.vA0, so, 0, 0);
    if constexpr (I == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rAl, (lptr_t)l, 16, (int)t.vA1, so, 0, 0);
    if constexpr (I == 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rBh, (lptr_t)l, 16, (int)t.vB, so, 0, 0);
    if constexpr (I == 5) __builtin_amdgcn_raw_ptr_buffer_load_lds(rBl, (lptr_t)l, 16, (int)t.vB, so, 0, 0);
  };
  auto issue_stage = [&](const Tile& t, int s, int sb) { sfor<0, NDMA>([&](auto ic) { issue_slot(ic, t, s, sb); }); };

  struct Frags { half8_t ah[TM], al[TM], bh[TN], bl[TN]; };
  // fragment read number q (0..11) of the stage at byte offset sb
  auto read_frag = [&](auto qc, Frags& f, int sb) {
    constexpr int q = decltype(qc)::value;
    const unsigned char* a = smem + sb + a_lane;
    const unsigned char* b = smem + sb + b_lane;
    if constexpr (q < 4) f.ah[q] = *reinterpret_cast<const half8_t*>(a + q * 1024);
    else if constexpr (q < 8) f.al[q - 4] = *reinterpret_cast<const half8_t*>(a + OFF_ALO + (q - 4) * 1024);
    else if constexpr (q < 10) f.bh[q - 8] = *reinterpret_cast<const half8_t*>(b + (q - 8) * 1024);
    else f.bl[q - 10] = *reinterpret_cast<const half8_t*>(b + B_PL + (q - 10) * 1024);
  };

  f32x16 acc[TM][TN];
  // MFMA number q (0..23) of a stage, pass-major: dependent MFMAs are 8 issue slots apart
  auto mfma_q = [&](auto qc, const Frags& f) {
    constexpr int q = decltype(qc)::value, ps = q / 8, g = q % 8, i = g / TN, j = g % TN;
    if constexpr (ps == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[j], f.al[i], acc[i][j], 0, 0, 0);
    else if constexpr (ps == 1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bl[j], f.ah[i], acc[i][j], 0, 0, 0);
    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[j], f.ah[i], acc[i][j], 0, 0, 0);
  };

  // s_waitcnt through the builtin (the compiler's own waitcnt insertion understands it):
  // gfx9 encoding vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14]
  constexpr auto wc_vm_lgkm0 = [](int n) { return (n & 15) | ((n >> 4) << 14) | 0x70; };
  constexpr int WC_LGKM0 = 0xC07F;

  // one ring step: [my DMA of stage t+1 has landed, fc has landed | barrier] then the 24 MFMAs on fc with the 12
  // fragment reads of stage t+1 (ring offset on) and the 6 DMA instructions of stage t+3 (-> ring offset oc, whose last
  // reads every wave completed before the barrier) spread between them
  // (a software L2 prefetch -- one dword per operand row touched 4 K blocks ahead of the ring, by LDS-DMA into a
  // landing zone -- was built and measured in round 3: 25-40 % SLOWER (64 separate lines per wave instruction cost the
  // texture path more than the hidden latency is worth); removed)
  auto step = [&](const Frags& fc, Frags& fn, const Tile& tl, int t, int oc, int on, auto rdc, auto dmac, auto vmc, auto evc) {
    constexpr bool RD = decltype(rdc)::value, DMA = decltype(dmac)::value && !(VAR & 4);
    constexpr int VM = decltype(vmc)::value;
    __builtin_amdgcn_s_waitcnt(wc_vm_lgkm0(VM));
    __builtin_amdgcn_s_barrier();
    // development-build experiments (round 4): VAR == 1 raises the wave's priority while it issues its DMA slots (the
    // partner wave of the other block then yields issue slots to the 60-185-cycle buffer_load ... lds instructions);
    // VAR == 33 issues the six DMA instructions in one burst behind the barrier instead of between the last MFMAs
    constexpr bool X_PRIO = (VAR == 1), X_BURST = (VAR == 33);
    if constexpr (DMA && X_BURST) {
      sfor<0, NDMA>([&](auto ic) { issue_slot(ic, tl, t + 3, oc); });
      __builtin_amdgcn_sched_barrier(0);
    }
    sfor<0, 24>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      mfma_q(qc, fc);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (RD && q < 12) {
        read_frag(qc, fn, on);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (DMA && !X_BURST && q >= 12 && (q & 1) == 0) {
        if constexpr (X_PRIO && q == 12) __builtin_amdgcn_s_setprio(1);
        issue_slot(std::integral_constant<int, (q - 12) / 2>{}, tl, t + 3, oc);
        if constexpr (X_PRIO && q == 22) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
      }
    });
  };
  using T_ = std::true_type; using F_ = std::false_type;

  // ---- generic epilogue: every mode of the descriptor at run time (plane residual, cat8 planes, relu / sigmoid,
  // ragged N ...); ~5x the instructions of the specialised forms, kept for the rare shapes ----
  auto epilogue_generic = [&](const Tile& done) {
    const float alpha = d.alpha;
    const float cs = d.Chi ? ldexpf(1.0f, RSP_PLANE_EXP(d.c_scale_log2)) : 1.0f;
    const bool c_f8 = RSP_PLANE_IS_F8(d.c_scale_log2);
    half_t* const chi = reinterpret_cast<half_t*>(d.Chi);
    half_t* const clo = reinterpret_cast<half_t*>(d.Clo);
    const float rsc = d.res_hi ? ldexpf(1.0f, -RSP_PLANE_EXP(d.res_scale_log2)) : 1.0f;
    const int colw = done.n0 + wn * 64 + 4 * hh;           // + j * 32 + 8 * q
    f32x4 bias4[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = colw + j * 32 + 8 * q;
        bias4[j][q] = (d.bias && col < N) ? *reinterpret_cast<const f32x4*>(d.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    sfor<0, TM>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int row = done.m0 + wm * 128 + i * 32 + l31;
      int cr = -1;
      if (row < M) cr = d.c_rowmap ? d.c_rowmap[row] : row;
      if (cr >= 0) {
        int64_t rrow = cr;
        if (d.res || d.res_hi) {
          if (d.res_mod > 0) rrow = cr - p.fd_resmod.div(cr) * d.res_mod;
          if (d.res_bmap) {
            const int rb = p.fd_resb.div(cr);
            rrow = (int64_t)d.res_bmap[rb] * d.res_brows + (cr - rb * d.res_brows);
          }
        }
        sfor<0, TN * 4>([&](auto jqc) {
          constexpr int jq = decltype(jqc)::value, j = jq / 4, q = jq % 4;
          const int col = colw + j * 32 + 8 * q;
          if (col < N) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = rsp_act(acc[i][j][4 * q + e] * alpha + bias4[j][q][e], d.act);
            if (d.res_hi) {
              const int64_t ro = ((int64_t)(col >> 5) * d.res_rows + rrow) * 32 + (col & 31);
              const half4_t rh = *reinterpret_cast<const half4_t*>(reinterpret_cast<const half_t*>(d.res_hi) + ro);
              const half4_t rl = *reinterpret_cast<const half4_t*>(reinterpret_cast<const half_t*>(d.res_lo) + ro);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += ((float)rh[e] + (float)rl[e]) * rsc;
            }
            if (d.res) {
              const f32x4 rv = *reinterpret_cast<const f32x4*>(d.res + rrow * d.ldr + col);
              v[0] += rv[0]; v[1] += rv[1]; v[2] += rv[2]; v[3] += rv[3];
            }
            if (d.act == RSP_ACT_RELU_POST) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = rsp_act_post(v[e], d.act);
            }
            if (d.C && (d.c_ncols <= 0 || col < d.c_ncols)) *reinterpret_cast<f32x4*>(d.C + (int64_t)cr * d.ldc + col) = v;
            if (d.Chi && col >= d.pl_col0) {
              const int pch = col - d.pl_col0;
              const int64_t po = ((int64_t)(pch >> 5) * d.c_rows + cr) * 32 + (pch & 31);
              rsp_store_planes4(chi, clo, po, f32x4{v[0] * cs, v[1] * cs, v[2] * cs, v[3] * cs}, c_f8);
            }
          }
        });
      }
    });
  };

  // ---- persistent tile walk ----
  const int xcd = blockIdx.x & 7;
  const int id_base = xcd * p.per_xcd, id_end = min((xcd + 1) * p.per_xcd, p.ntiles);
  auto next_id = [&]() -> int {                       // block-uniform: lane 0 of wave 0 draws, LDS-free broadcast
    int v = 0;
    if (tid == 0) v = (int)__hip_atomic_fetch_add(p.ticket + xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
  };
  volatile int* s_next = reinterpret_cast<volatile int*>(smem + NS * STAGE);
  auto draw = [&]() -> int {                          // called with no DMA in flight (kernel entry, end of a K loop)
    if (tid == 0) *s_next = id_base + next_id();
    __syncthreads();
    const int v = *s_next;
    __syncthreads();
    return __builtin_amdgcn_readfirstlane(v);
  };
  auto finish = [&]() {                               // the last block of the launch re-arms the ticket slot
    if (tid == 0) {
      const unsigned dn = __hip_atomic_fetch_add(p.ticket + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (dn == gridDim.x - 1) {
        for (int i = 0; i < 9; ++i) __hip_atomic_store(p.ticket + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  };
  int id = draw();
  if (id >= id_end) { finish(); return; }
  Tile cur = tile_setup(id);
  sfor<0, NS>([&](auto sc) { issue_stage(cur, decltype(sc)::value, decltype(sc)::value * STAGE); });

  while (true) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    unsigned long long ts0 = 0, ts1 = 0;
    if constexpr (VAR & 32) ts0 = __builtin_amdgcn_s_memtime();
    Frags f0, f1;
    // stage 0 of this tile has landed (mine: <= 12 younger DMA instructions outstanding; everybody's: barrier)
    __builtin_amdgcn_s_waitcnt(wc_vm_lgkm0(2 * NDMA));
    __builtin_amdgcn_s_barrier();
    sfor<0, 12>([&](auto qc) { read_frag(qc, f0, 0); });

    int oc = 0, on = STAGE;
    auto adv = [&]() { oc = on; on += STAGE; if (on == NS * STAGE) on = 0; };
    int t = 0;
    for (; t + 4 < nk; t += 2) {
      step(f0, f1, cur, t, oc, on, T_{}, T_{}, std::integral_constant<int, NDMA>{}, T_{}); adv();
      step(f1, f0, cur, t + 1, oc, on, T_{}, T_{}, std::integral_constant<int, NDMA>{}, F_{}); adv();
    }
    // t = nk - 4: the last stage that still has a DMA to issue (nk - 1)
    step(f0, f1, cur, t, oc, on, T_{}, T_{}, std::integral_constant<int, NDMA>{}, T_{}); adv();
    // The NEXT tile's ticket is drawn here, behind the tile's last DMA: the atomic's round trip to the L2 (about a
    // microsecond, formerly exposed between the K loop and the epilogue) passes under the last three steps.  vmcnt retires
    // in order, so the youngest operation may stay outstanding if the two remaining waits allow one more; every wave
    // issues one (waves 1-3 on a scratch word) so that the immediates are the same for all of them.
    unsigned tk = 0;
    if (lane == 0)
      tk = __hip_atomic_fetch_add(wave == 0 ? p.ticket + xcd : p.ticket + 12, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_sched_barrier(0);
    step(f1, f0, cur, t + 1, oc, on, T_{}, F_{}, std::integral_constant<int, NDMA + 1>{}, F_{}); adv();
    step(f0, f1, cur, t + 2, oc, on, T_{}, F_{}, std::integral_constant<int, 1>{}, T_{}); adv();
    // last stage: its fragments are in f1; once every wave holds its own the ring is free for the next tile
    __builtin_amdgcn_s_waitcnt(WC_LGKM0);
    __builtin_amdgcn_s_barrier();
    sfor<0, 24>([&](auto qc) { mfma_q(qc, f1); });
    __builtin_amdgcn_sched_barrier(0);

    if constexpr (VAR & 32) ts1 = __builtin_amdgcn_s_memtime();
    // The epilogue is VALU / VMEM work issued into the gaps of the co-resident block's MFMA stream; at equal priority
    // the SIMD's arbiter gives the older wave's stream nearly every slot (measured, time stamps: the younger block's
    // epilogue took 145k cycles against 19k alone).  An MFMA stream needs one issue slot in 32 cycles: it loses little
    // when the epilogue wave goes first.
    const Tile done = cur;
    if (tid == 0) *s_next = id_base + (int)tk;        // (the next write of this word is a whole tile of barriers away)
    __syncthreads();
    id = __builtin_amdgcn_readfirstlane(*s_next);
    const bool more = id < id_end;
    auto queue_next = [&]() {                         // the next tile's first three stages fly during the epilogue
      if (more) sfor<0, NS>([&](auto sc) { issue_stage(cur, decltype(sc)::value, decltype(sc)::value * STAGE); });
    };

    // ---- epilogue of tile `done`, straight from the (transposed) accumulators: lane = row, register quad = 4 columns ----
    if constexpr (VAR & 8) {
      if (more) cur = tile_setup(id);
      queue_next();
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(acc[i][j]));
    } else if constexpr (EPI & E_GENERIC) {
      if (more) cur = tile_setup(id);
      queue_next();
      epilogue_generic(done);
    } else {
      // Specialised form.  Host guarantees: N, c_ncols, pl_col0 multiples of 64 (a wave's 64 columns are all fp32
      // output, all plane output, or both), act in {none, GELU per E_GELU}, no plane residual, no cat8 output, every
      // tensor the epilogue touches < 2^31 bytes.
      //   * The accumulators hold one output ROW per lane: stored as they are, every store / residual load instruction
      //     touches 32 rows with 16-32 bytes each (measured: the stores of a GELU + plane epilogue cost 10 % of the
      //     GEMM, the residual loads of the proj epilogue 16 %).  Each wave therefore turns its tile by 90 degrees
      //     through a PRIVATE 8 KB piece of the (now idle) ring, 32 rows at a time: ds_write_b128 of the quads, XOR
      //     swizzle of the 16-byte unit with row & 7 (conflict free both ways), read back with 16 lanes per row -- a wave
      //     instruction then moves 4 rows x 256 contiguous bytes (fp32) or 2 x 256 contiguous bytes (a plane).  No
      //     block barrier is involved: one wave's LDS operations execute in order.
      //   * BRANCH-FREE: all memory operations are buffer operations -- a row that is not stored (beyond M,
      //     c_rowmap < 0) carries an offset beyond the descriptor's range (stores are dropped, loads return 0), an absent
      //     bias / row map is a descriptor of 0 bytes.  Straight-line code lets hipcc count its waits, and vmcnt retires
      //     in order and counts stores, so ORDER matters: the destination rows and the residual of row group i + 1 are
      //     requested BEFORE row group i is stored (waiting for them never waits for stores).
      //   * The next tile's DMA is queued once every wave has read its last piece back (one barrier) and no loaded
      //     value is pending (hipcc answers any use of a loaded value with vmcnt(0) while an LDS-DMA is in flight).
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      constexpr int NG = 8;                                   // read-back groups per 32-row pass: 4 rows x 64 columns
      // VAR & 1 (experiment): results stored with the non-temporal hint -- they are not read again by this kernel and
      // should not push operand lines out of the L2 (the K-streams of 64 concurrent tiles per XCD live there)
      constexpr int ST_AUX = 0;      // (round 3: non-temporal result stores, VAR & 1 -- measured +-2 %, not adopted)
      const float alpha = d.alpha;
      const float cs = (EPI & E_PL) ? ldexpf(1.0f, RSP_PLANE_EXP(d.c_scale_log2)) : 1.0f;
      const int cols0 = done.n0 + wn * 64;                    // scalar: first column of this wave
      const bool active = cols0 < N;                          // N % 64 == 0: all 64 columns or none
      const bool do_c = (EPI & E_C) && active && (d.c_ncols <= 0 || cols0 < d.c_ncols);
      const bool do_p = (EPI & E_PL) && active && cols0 >= d.pl_col0;
      const int lr0 = lane >> 4, c4 = (lane & 15) * 4;        // read-back mapping: row g * 4 + lr0, columns c4 .. c4 + 3
      unsigned char* const et = smem + wave * (32 * 256);     // this wave's transposition piece
      const int wr_off = l31 * 256;                           // write side: row l31, unit (j*8 + 2q + hh) ^ (l31 & 7)
      const int rd_off = lr0 * 256;                           // read side: row g*4 + lr0, unit (c4 / 4) ^ (row & 7)
      const auto rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.bias), 0, d.bias ? N * 4 : 0, 0x00020000);
      const f32x4 bias4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rB, (cols0 + c4) * 4, 0, 0));
      const auto rM = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(d.c_rowmap), 0, d.c_rowmap ? M * 4 : 0, 0x00020000);
      const auto rR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.res), 0, (EPI & E_RES) ? 0x7fffffff : 0, 0x00020000);
      const int row_w = done.m0 + wm * 128 + lr0;             // + i * 32 + g * 4
      // destination rows: without a row map (E_RMAP clear) plain arithmetic; with one, ALL of the tile's are requested
      // up front (32 registers): a request issued between two passes would be waited for behind the previous pass's
      // stores
      int crow_m[(EPI & E_RMAP) ? TM : 1][NG];
      if constexpr (EPI & E_RMAP) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int g = 0; g < NG; ++g)
            crow_m[i][g] = (int)__builtin_amdgcn_raw_buffer_load_b32(rM, (row_w + i * 32 + g * 4) * 4, 0, 0);
      }
      auto crow_of = [&](auto ic, int g) -> int {             // -1: nothing stored for this row
        constexpr int i = decltype(ic)::value;
        const int row = row_w + i * 32 + g * 4;
        if constexpr (EPI & E_RMAP) return row < M ? crow_m[i][g] : -1;
        else return row < M ? row : -1;
      };
      f32x4 rv[2][NG];
      auto res_load = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const int crc = max(crow_of(ic, g), 0);             // rows that are not stored read row 0 (never used)
          int rrow = crc;
          if (d.res_mod > 0) rrow = crc - p.fd_resmod.div(crc) * d.res_mod;
          rv[i & 1][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rR, (rrow * d.ldr + cols0 + c4) * 4, 0, 0));
        }
      };
      if (more) cur = tile_setup(id);                         // (its row-map loads join the batch)
      if constexpr (EPI & E_RES) res_load(std::integral_constant<int, 0>{});
      // output descriptors: without a row map rows are relative to the tile (any tensor size), with one to the tensor
      const int64_t c_row0 = d.c_rowmap ? 0 : done.m0;
      const int c_rows_here = d.c_rowmap ? 0x7fffffff / max(d.ldc * 4, 1) : M - done.m0;
      const auto rC = __builtin_amdgcn_make_buffer_rsrc(d.C ? d.C + c_row0 * d.ldc : nullptr, 0,
                                                        do_c ? (int)min((int64_t)c_rows_here * d.ldc * 4, (int64_t)0x7fffffff) : 0, 0x00020000);
      const int pl_bytes = do_p ? (int)(((int64_t)((N - d.pl_col0) >> 5) * d.c_rows) << 6) : 0;
      const auto rH = __builtin_amdgcn_make_buffer_rsrc(d.Chi, 0, pl_bytes, 0x00020000);
      const auto rL = __builtin_amdgcn_make_buffer_rsrc(d.Clo, 0, pl_bytes, 0x00020000);
      // plane offset of this lane's 4 columns: K block (cols0 - pl_col0) / 32 + c4 / 32, 8 bytes at (c4 & 31) * 2
      const int pl_lane = (((cols0 - d.pl_col0) >> 5) + (c4 >> 5)) * (d.c_rows << 6) + (c4 & 31) * 2;
      sfor<0, TM>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        // (1) this pass's 32 x 64 accumulators -> LDS, row-major, swizzled
        sfor<0, TN * 4>([&](auto jqc) {
          constexpr int jq = decltype(jqc)::value, j = jq / 4, q = jq % 4;
          const f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
          *reinterpret_cast<f32x4*>(et + wr_off + (((j * 8 + 2 * q + hh) ^ (l31 & 7)) << 4)) = v;
        });
        RSP_WAVE_LOCKSTEP();
        // (3) read back: 4 rows x 64 columns per instruction
        f32x4 x[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const int lr = g * 4 + lr0;
          x[g] = *reinterpret_cast<const f32x4*>(et + rd_off + g * 1024 + ((((c4 >> 2)) ^ (lr & 7)) << 4));
        }
        RSP_WAVE_LOCKSTEP();                                  // (the next pass overwrites the piece)
        if constexpr (i + 1 == TM) {                          // every wave has its last piece: the ring is free
          asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
          __builtin_amdgcn_s_waitcnt(WC_LGKM0);
          __builtin_amdgcn_s_barrier();
        }
        // (4) value = act(acc * alpha + bias) + residual
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          x[g] = x[g] * alpha + bias4;
          if constexpr (EPI & E_GELU) x[g] = rsp_gelu4(x[g]);
          if constexpr (EPI & E_RES) x[g] += rv[i & 1][g];
        }
        // (the residual of the NEXT pass is requested before this pass stores)
        if constexpr (i + 1 < TM) {
          if constexpr (EPI & E_RES) res_load(std::integral_constant<int, i + 1>{});
        } else {
          asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
          queue_next();                                       // every loaded value has been consumed
        }
        // (5) stores
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const int cr = crow_of(ic, g);
          if constexpr ((EPI & E_C) && !(VAR & 2)) {
            const unsigned co = cr < 0 ? OOB : (unsigned)(((cr - (int)c_row0) * d.ldc + cols0 + c4) * 4);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x[g]), rC, co, 0, ST_AUX);
          }
          if constexpr (EPI & E_PL) {
            half4_t h4, l4;
            f32x4 rem;
            rsp_split4(x[g] * cs, h4, l4, rem);
            if constexpr (VAR & 2) {
              asm volatile("" ::"v"(h4), "v"(l4));
            } else {
              const unsigned po = cr < 0 ? OOB : (unsigned)(cr * 64 + pl_lane);
              __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, h4), rH, po, 0, ST_AUX);
              __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, l4), rL, po, 0, ST_AUX);
            }
          }
          if constexpr ((EPI & E_C) && (VAR & 2)) asm volatile("" ::"v"(x[g]));
        }
      });
    }
    if constexpr (VAR & 32) {
      if (p.trace && tid == 0 && trace_n < 16) {
        unsigned long long* t = p.trace + ((size_t)blockIdx.x * 16 + trace_n) * 4;
        t[0] = ts0; t[1] = ts1; t[2] = __builtin_amdgcn_s_memtime();
        t[3] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) << 32) | hw_id;
      }
      ++trace_n;
    }
    if (!more) break;
  }
  finish();
}

template <int VAR, int EPI>
int launch_s2(const RspGemmDesc& d, hipStream_t s) {
  unsigned* const tickets = s2_ticket_base();
  if (!tickets) return RSP_ELAUNCH;
  S2P p; p.d = d;
  p.fd_resmod = make_fastdiv(d.res_mod); p.fd_resb = make_fastdiv(d.res_brows);
  p.nbm = (d.M + BM - 1) / BM; p.nbn = (d.N + BN - 1) / BN;
  const long long nt = (long long)p.nbm * p.nbn;
  if (nt > 0x3fffffffLL) return RSP_EINVAL;
  p.ntiles = (int)nt;
  p.per_xcd = (p.ntiles + 7) / 8;
  p.group_m = (d.tile_hint >> 8) & 0xff;
  // tile order: groups of group_m M-tiles x all N-tiles, M fastest (the ~64 tiles an XCD runs at once then share
  // group_m A panels and 64 / group_m W panels).  tools/gemm_s2_group_sweep.py on the ViT-H shapes: 8 is best for the
  // K = 1280, N <= 3840 shapes, the wide lin1 (N = 5120) prefers 4 (+2.7 %), the long-K lin2 (K = 5120: an A panel is
  // 5 MB) prefers 2 (+2.4 %)
  if (p.group_m == 0) p.group_m = d.K >= 4096 ? 2 : (d.N >= 4096 ? 4 : 8);
  p.trace = g_s2_trace;
  p.ticket = tickets + (size_t)(g_s2_slot.fetch_add(1u, std::memory_order_relaxed) % TICKET_SLOTS) * TICKET_WORDS;
  int nblk = p.ntiles < 512 ? (p.ntiles + 7) / 8 * 8 : 512;     // a multiple of 8: every XCD gets nblk / 8 walkers
  hipLaunchKernelGGL((gemm_f16x3_s2_kernel<VAR, EPI>), dim3((unsigned)nblk), dim3(NTHR), 0, s, p);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

}  // namespace

// the shapes / modes this kernel implements (everything else stays with gemm_dma.hip)
bool rsp_gemm_s2_eligible(const RspGemmDesc& d) {
  if (!(d.Ahi && d.Alo) || d.conv_k != 0 || d.ct_W > 0 || d.ln_gamma || d.hd_out) return false;
  if (RSP_PLANE_IS_F8(d.a_scale_log2)) return false;
  if ((d.N & 3) || (d.K & 31) || d.K < 64 || d.M <= 0 || d.a_rows <= 0) return false;
  if (d.C && ((d.ldc & 3) || (reinterpret_cast<uintptr_t>(d.C) & 15))) return false;
  if (d.res && ((d.ldr & 3) || (reinterpret_cast<uintptr_t>(d.res) & 15))) return false;
  if (d.bias && (reinterpret_cast<uintptr_t>(d.bias) & 15)) return false;
  if (d.res_hi && (!d.res_lo || d.res || d.res_rows <= 0)) return false;
  if ((d.pl_col0 & 31) || (d.c_ncols & 3) || d.pl_col0 < 0) return false;
  const long long brows = d.b_rows > 0 ? d.b_rows : d.N;
  const long long kb = d.K / 32;
  if (kb * d.a_rows * 64 >= (1LL << 31) || kb * brows * 64 >= (1LL << 31)) return false;   // 32-bit buffer offsets
  return true;
}

// the epilogue specialisation that serves this descriptor (E_GENERIC: only the run-time form does)
static int s2_epilogue_of(const RspGemmDesc& d) {
  // the branch-free epilogue addresses every tensor it touches with 32-bit buffer offsets
  const long long GB2 = 1LL << 31;
  const long long c_bytes = d.C ? (d.c_rowmap ? (long long)d.c_rows * d.ldc * 4 : 0) : 0;   // no row map: tile-relative
  const long long pl_bytes = d.Chi ? (((long long)((d.N - d.pl_col0) >> 5) * d.c_rows) << 6) : 0;
  const long long res_bytes = d.res ? (long long)(d.res_mod > 0 ? d.res_mod : (d.c_rowmap ? d.c_rows : d.M)) * d.ldr * 4 : 0;
  const bool fast = !(d.N & 63) && !(d.c_ncols & 63) && !(d.pl_col0 & 63) && !d.res_hi &&
                    (d.act == RSP_ACT_NONE || d.act == RSP_ACT_GELU) && !(d.Chi && RSP_PLANE_IS_F8(d.c_scale_log2)) &&
                    !(d.c_rowmap && d.c_rows <= 0) && c_bytes < GB2 && pl_bytes < GB2 && res_bytes < GB2 &&
                    !(d.res && d.res_bmap) &&      /* gathered residual rows: size unknown here, generic path */
                    (long long)256 * d.ldc * 4 < GB2;
  if (!fast) return E_GENERIC;
  return (d.res ? E_RES : 0) | (d.act == RSP_ACT_GELU ? E_GELU : 0) | (d.C ? E_C : 0) | (d.Chi ? E_PL : 0) |
         (d.c_rowmap ? E_RMAP : 0);
}

// 1: this kernel is the product choice for the descriptor (eligible, a specialised epilogue exists, enough tiles to fill
// the 512 block slots once); 0: stay with gemm_dma.hip
int rsp_gemm_s2_epilogue_of(const RspGemmDesc& d);
int rsp_gemm_s2_auto(const RspGemmDesc& d) {
  const int e = rsp_gemm_s2_epilogue_of(d);
  const bool have = e >= 0 && e != E_GENERIC;
  const long long nt = (long long)((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN);
  return have && nt >= 256 && d.K >= 128;
}

// tests / profiler labels: the epilogue form rsp_gemm_s2_dispatch(d, 0) runs (bit flags E_RES = 1, E_GELU = 2, E_C = 4,
// E_PL = 8, E_RMAP = 16; 64 = the run-time form), -1 for a descriptor this kernel does not implement
int rsp_gemm_s2_epilogue_of(const RspGemmDesc& d) {
  if (!rsp_gemm_s2_eligible(d)) return -1;
  const int e = s2_epilogue_of(d);
  const bool have = e == E_C || e == (E_C | E_RES) || e == (E_C | E_RES | E_RMAP) || e == (E_C | E_PL) || e == (E_C | E_PL | E_RMAP) || e == E_PL ||
                    e == (E_PL | E_GELU) || e == (E_C | E_GELU);
  return have ? e : E_GENERIC;
}

// var: experiment switches of the kernel (0 = product); the epilogue specialisation follows from the descriptor
int rsp_gemm_s2_dispatch(const RspGemmDesc& d, int var, hipStream_t s) {
  const int epi = (var & 64) ? E_GENERIC : s2_epilogue_of(d);
  var &= 63;
#define S2_CASE(V, E) if (var == V && epi == (E)) return launch_s2<V, (E)>(d, s)
  S2_CASE(0, E_C);                 // plain
  S2_CASE(0, E_C | E_RES);         // proj (global layers), lin2, patch embed
  S2_CASE(0, E_C | E_RES | E_RMAP); // proj of a windowed layer (row scatter)
  S2_CASE(0, E_C | E_PL);          // qkv (column ranges), fp32 + planes
  S2_CASE(0, E_C | E_PL | E_RMAP); // qkv of a windowed layer: token rows scattered to window order
  S2_CASE(0, E_PL);                // planes only
  S2_CASE(0, E_PL | E_GELU);       // lin1
  S2_CASE(0, E_C | E_GELU);
  S2_CASE(0, E_GENERIC);
#ifdef RSP_S2_ABLATIONS          /* tools/gemm_s2_exp.py time: RSP_DEV_BUILD=1 python -m rsprompter_amd.build */
  S2_CASE(1, E_C | E_RES); S2_CASE(1, E_PL | E_GELU);          // raised priority around the DMA slots (round 4)
  S2_CASE(2, E_C | E_RES); S2_CASE(2, E_PL | E_GELU); S2_CASE(2, E_PL); S2_CASE(8, E_PL);   // no stores
  S2_CASE(4, E_C | E_RES); S2_CASE(4, E_PL | E_GELU);          // no DMA in the loop
  S2_CASE(8, E_C | E_RES); S2_CASE(8, E_PL | E_GELU);          // no epilogue
  S2_CASE(16, E_C | E_RES); S2_CASE(16, E_PL | E_GELU);        // cache-hot DMA sources
  S2_CASE(32, E_C | E_RES); S2_CASE(32, E_PL | E_GELU);        // time stamps
  S2_CASE(33, E_C | E_RES); S2_CASE(33, E_PL | E_GELU);        // the six DMA slots of a step as one burst (round 4)
  S2_CASE(1, E_C); S2_CASE(1, E_C | E_PL);
#endif
#undef S2_CASE
  if (var != 0) return RSP_EINVAL;
  return launch_s2<0, E_GENERIC>(d, s);       // any other combination of outputs
}

#ifdef RSP_S2_ABLATIONS
// tools only (not part of include/rsp_hip.h): device buffer [512][16][4] u64 for the time-stamp variant
extern "C" void rsp_debug_s2_trace(void* p) { g_s2_trace = reinterpret_cast<unsigned long long*>(p); }
#endif
