// fp16x3 GEMM, "ping-pong" form (round 5): one 512-thread block per CU, block tile 256 x 256, in which the two waves of
// every SIMD ALTERNATE between a matrix phase and a load phase.  Same arithmetic, operands and MFMA order as
// gemm_s2.hip / gemm_dma.hip (a_lo*b_hi + a_hi*b_lo + a_hi*b_hi per 16 k, fp32 accumulate): results are bit-identical.
//
// Why: in gemm_s2.hip (two independent 4-wave blocks per CU) the wave that issues MFMAs also issues its block's DMA
// (`buffer_load ... lds`, 60-185 issue cycles each), meets its block at a barrier per K step, and the SIMD's two waves
// overlap only by chance: matrix pipe busy 71 %, 427-506 TFLOP/s without the DMA instructions.  Here
//   * waves 0-3 (group 0: tile rows 0-127) and waves 4-7 (group 1: rows 128-255) sit pairwise on the four SIMDs; a wave
//     tile is 128 x 64 (8 accumulators) as before;
//   * phase 2t:     group 0 issues the 24 MFMAs of K step t from fragment REGISTERS -- nothing else;
//                   group 1 reads its 12 fragments of step t from LDS and issues its 4 DMA instructions of stage t + 3;
//     phase 2t + 1: the roles swap (group 0 loads step t + 1).  One s_barrier between phases: the matrix pipe of a SIMD
//     always has exactly one wave feeding it, DMA issue / LDS reads / waits sit in the other wave's phase by construction
//     (MI355X_MICROARCH.md "Two waves per SIMD": the 8-wave compute / load alternation);
//   * one 256 x 256 tile instead of two 256 x 128: a K step moves 32 KB instead of 48 KB through L2 -> LDS for the same
//     matrix work (4 DMA instructions per wave and step instead of 6), ring of 4 stages x 32 KB;
//   * static tile walk: the blocks of an XCD take consecutive entries of that XCD's part of the (grouped, M-fastest)
//     tile list, so the 32 tiles an XCD runs at once are an 8 x 4 patch sharing 8 A panels and 4 W panels in its L2, and
//     -- the blocks run in step -- in the same K phase.
// Cost: one block per CU, so a tile's epilogue is exposed (in gemm_s2 it runs under the other block's K loop).  Measured
// (profiles/r5_gemm_pp/): the K loop runs at 765 of the ideal 768 cycles per phase; the epilogue of a 256 x 256 tile takes
// 18-33 k cycles of a 159 k cycle tile (K = 1280) -- 256 KB through a CU's ~14 B/clk vector-store path, plus 11 k cycles of
// GELU -- whether 8 or 256 CUs run, staggered or not (block-start stagger and CU-count sweeps: no change), i.e. a per-CU cost
// that only a co-resident block could hide.  It is barrier-free here: every wave turns its accumulators through LDS bytes that only its own DMA slots overwrite, stores,
// and queues ITS part of the next tile's first four stages -- no wave waits for another until the next tile starts.
//
// Two instantiations: BM = 256 (above) and BM = 128 -- block tile 128 x 256, wave tile 64 x 64, a phase = 12 MFMAs, ring of
// 6 stages x 24 KB (the same prefetch distance in time), 3 DMA instructions per wave and stage -- for shapes whose 256 x 256
// tiles fill the last round of CUs badly: a static walk idles whole CUs there (ViT-H N = 1280: 640 tiles = 2.5 rounds, but
// 1280 tiles of 128 x 256 = 5 rounds).
// (A third form -- gemm_pp2: 128 x 256 tiles with a second accumulator set and the previous tile's epilogue inside the load
// phases -- was built and measured: bit-exact, and 5-15 % SLOWER than BM = 256 everywhere: its 3 x 48 KB ring leaves ~2
// phases of DMA slack, VALU beside the partner's MFMA stream runs at a quarter of its rate, and untransposed stores cost
// the texture path 32 lines per instruction; profiles/r5_gemm_pp2_negative/.)
// (Also measured and dropped: splitting the tiles of an XCD's ragged last round in K among its idle blocks -- partial sums
// through a workspace tile, release / acquire flags -- for the 2.5-round shapes (ViT-H lin2 / proj: 640 tiles).  Correct
// (1e-6 of the unsplit sums, emulator + GPU), but worth +3 % on lin2 and -14 % on proj: the chip runs these kernels at
// its POWER limit, and a half-empty last round simply clocks higher -- the idle CUs are not the loss they look like.
// profiles/r5_gemm_pp_split_tail_negative/.)
// Scope: what gemm_s2.hip's specialised epilogues cover (plain plane-path GEMMs); chosen by rsp_gemm for shapes with
// enough tiles to fill the 256 CUs several times (rsp_gemm_pp_auto).
#include <type_traits>
#include "rsp_common.h"

namespace {

constexpr int KS = 16;                       // k per ring stage
constexpr int BN = 256;
constexpr int NTHR = 512;
constexpr int B_PL = BN * 32;                // bytes of one W plane of one stage
constexpr int TN = 2;
constexpr unsigned OOB = 0x80000000u;

typedef __attribute__((address_space(3))) void* lptr_t;

struct PPP {
  RspGemmDesc d;
  FastDiv fd_resmod;
  int nbm, nbn, ntiles, per_xcd, group_m, nper;
  unsigned long long* trace;     // tools only (VAR bit 5): per block, tile and group {loop start, loop end, epilogue end, hw id}
};
unsigned long long* g_pp_trace = nullptr;

constexpr int E_RES = 1, E_GELU = 2, E_C = 4, E_PL = 8, E_RMAP = 16;

template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, N>(f);
  }
}

// gfx9 s_waitcnt immediate: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14]
constexpr int wc(int vm, int lgkm) { return (vm & 15) | ((vm >> 4) << 14) | 0x70 | ((lgkm & 15) << 8); }
constexpr int LGKM_ANY = 15, VM_ANY = 63;

// VAR (development builds only, -DRSP_S2_ABLATIONS; 0 = product): bit 2 = no DMA inside the K loop (garbage results), bit 3 =
// no epilogue (nothing stored), bit 5 = time stamps into g_pp_trace
template <int EPI, int VAR = 0, int BM = 256>
__global__ __launch_bounds__(NTHR, 2) void gemm_f16x3_pp_kernel(const PPP p) {
  constexpr int TM = BM / 64;                  // 32-row blocks of a wave tile (the two groups split the tile's rows)
  constexpr int NS = BM == 256 ? 4 : 6;        // ring depth
  constexpr int PD = NS - 1;                   // load(t) requests stage t + PD
  constexpr int A_PL = BM * 32;                // bytes of one A plane of one stage
  constexpr int OFF_ALO = A_PL, OFF_BHI = 2 * A_PL, OFF_BLO = 2 * A_PL + B_PL;
  constexpr int STAGE = 2 * A_PL + 2 * B_PL;   // 32768 | 24576
  constexpr int NDMA = STAGE / (NTHR * 16);    // DMA instructions per wave and stage: 4 | 3
  constexpr int NRD = 2 * TM + 2 * TN, NMF = 3 * TM * TN;
  static_assert(NS * NDMA >= 8, "the epilogue turns its tile through 8 of the wave's own DMA slots");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * STAGE];
  const RspGemmDesc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wn = wave & 3;
  const int hh = lane >> 5, l31 = lane & 31;
  const int M = d.M, N = d.N;
  const int nk = d.K / KS;

  const int a_kstr = d.a_rows * 64, b_kstr = (d.b_rows > 0 ? d.b_rows : N) * 64;
  const int a_bytes = (d.K / 32) * a_kstr, b_bytes = (d.K / 32) * b_kstr;
  const __amdgpu_buffer_rsrc_t rAh = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(d.Ahi), 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rAl = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(d.Alo), 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rBh = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(d.Bhi), 0, b_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rBl = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(d.Blo), 0, b_bytes, 0x00020000);

  // DMA lane constants (1 KiB per instruction, lane-linear in LDS; the chunk swizzle -- bit 3 of the row -- goes on the
  // SOURCE offset, as in gemm_s2.hip).  BM = 256: wave w fills rows w * 32 .. + 31 of all four planes (slots A_hi, A_lo, W_hi,
  // W_lo); BM = 128: rows (w & 3) * 32 .. of ONE A plane (waves 0-3 hi, 4-7 lo) and rows w * 32 .. of both W planes.
  const int dA_row = (BM == 256 ? wave : (wave & 3)) * 32 + (lane >> 1);
  const int dB_row = wave * 32 + (lane >> 1);
  const int d_chunk = ((lane & 1) ^ ((lane >> 4) & 1)) << 4;
  // byte offset inside a stage of this wave's DMA slot I
  auto slot_off = [&](int I) -> int {
    if constexpr (BM == 256) return (I < 2 ? I * A_PL : OFF_BHI + (I - 2) * B_PL) + wave * 1024;
    else return I == 0 ? grp * A_PL + (wave & 3) * 1024 : OFF_BHI + (I - 1) * B_PL + wave * 1024;
  };

  const int f_chunk = (hh ^ ((l31 >> 3) & 1)) << 4;
  const int a_lane = (grp * (BM / 2) + l31) * 32 + f_chunk;
  const int b_lane = OFF_BHI + (wn * 64 + l31) * 32 + f_chunk;

  struct Tile { int m0, n0; unsigned vA, vB; };
  auto tile_setup = [&](int id) {
    Tile t;
    int mb = id / p.nbn, nb = id - mb * p.nbn;
    if (p.group_m > 1) {
      const int per = p.group_m * p.nbn;
      const int g = id / per, rem = id - g * per;
      const int first = g * p.group_m;
      const int gsz = min(p.nbm - first, p.group_m);
      nb = rem / gsz;
      mb = first + (rem - nb * gsz);
    }
    t.m0 = mb * BM; t.n0 = nb * BN;
    const int gm = t.m0 + dA_row;
    int srow = -1;
    if (gm < M) srow = d.a_rowmap ? d.a_rowmap[gm] : gm;
    t.vA = srow < 0 ? OOB : (unsigned)srow * 64u + (unsigned)d_chunk;
    const int gn = t.n0 + dB_row;
    t.vB = gn < N ? (unsigned)gn * 64u + (unsigned)d_chunk : OOB;
    return t;
  };
  const __amdgpu_buffer_rsrc_t rAg = grp ? rAl : rAh;     // BM = 128: the A plane this wave requests
  auto issue_slot = [&](auto ic, const Tile& t, int s, int sb) {
    constexpr int I = decltype(ic)::value;
    constexpr bool isA = BM == 256 ? I < 2 : I == 0;
    const int so = (s >> 1) * (isA ? a_kstr : b_kstr) + (s & 1) * 32;
    unsigned char* l = smem + sb + slot_off(I);
    if constexpr (BM == 256) {
      if constexpr (I == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rAh, (lptr_t)l, 16, (int)t.vA, so, 0, 0);
      if constexpr (I == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rAl, (lptr_t)l, 16, (int)t.vA, so, 0, 0);
      if constexpr (I == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rBh, (lptr_t)l, 16, (int)t.vB, so, 0, 0);
      if constexpr (I == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rBl, (lptr_t)l, 16, (int)t.vB, so, 0, 0);
    } else {
      if constexpr (I == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rAg, (lptr_t)l, 16, (int)t.vA, so, 0, 0);
      if constexpr (I == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rBh, (lptr_t)l, 16, (int)t.vB, so, 0, 0);
      if constexpr (I == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rBl, (lptr_t)l, 16, (int)t.vB, so, 0, 0);
    }
  };
  auto issue_stage = [&](const Tile& t, int s, int sb) { sfor<0, NDMA>([&](auto ic) { issue_slot(ic, t, s, sb); }); };

  struct Frags { half8_t ah[TM], al[TM], bh[TN], bl[TN]; };
  auto read_frag = [&](auto qc, Frags& f, int sb) {
    constexpr int q = decltype(qc)::value;
    const unsigned char* a = smem + sb + a_lane;
    const unsigned char* b = smem + sb + b_lane;
    if constexpr (q < TM) f.ah[q] = *reinterpret_cast<const half8_t*>(a + q * 1024);
    else if constexpr (q < 2 * TM) f.al[q - TM] = *reinterpret_cast<const half8_t*>(a + OFF_ALO + (q - TM) * 1024);
    else if constexpr (q < 2 * TM + TN) f.bh[q - 2 * TM] = *reinterpret_cast<const half8_t*>(b + (q - 2 * TM) * 1024);
    else f.bl[q - 2 * TM - TN] = *reinterpret_cast<const half8_t*>(b + B_PL + (q - 2 * TM - TN) * 1024);
  };

  f32x16 acc[TM][TN];
  auto mfma_q = [&](auto qc, const Frags& f) {      // pass-major, the order of gemm_s2.hip (bit-identical sums)
    constexpr int q = decltype(qc)::value, ps = q / (TM * TN), g = q % (TM * TN), i = g / TN, j = g % TN;
    if constexpr (ps == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[j], f.al[i], acc[i][j], 0, 0, 0);
    else if constexpr (ps == 1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bl[j], f.ah[i], acc[i][j], 0, 0, 0);
    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[j], f.ah[i], acc[i][j], 0, 0, 0);
  };


  Frags fr;
  // load phase of K step t: fragments of stage t (ring offset sb), then this wave's DMA of stage t + PD into the buffer
  // stage t - 1 lived in (offset db; every wave finished reading it before the barrier in front of this phase)
  auto load_phase = [&](const Tile& tl, int t, int sb, int db, auto dmac) {
    sfor<0, NRD>([&](auto qc) { read_frag(qc, fr, sb); });
    if constexpr (decltype(dmac)::value && !(VAR & 4)) {
      __builtin_amdgcn_sched_barrier(0);
      issue_stage(tl, t + PD, db);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto compute_phase = [&]() {
    sfor<0, NMF>([&](auto qc) {
      mfma_q(qc, fr);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  auto phase_end = [&](auto vmc, auto lgc) {       // [my DMA of the stage read next has landed | my fragments are here] barrier
    __builtin_amdgcn_s_waitcnt(wc(decltype(vmc)::value, decltype(lgc)::value));
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  using T_ = std::true_type; using F_ = std::false_type;
#define VMC(n) std::integral_constant<int, (n)>{}
#define LG0 std::integral_constant<int, 0>{}
#define LGX std::integral_constant<int, LGKM_ANY>{}

  // ---- static tile walk ----
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int id_end = min((xcd + 1) * p.per_xcd, p.ntiles);
  int id = xcd * p.per_xcd + jb;
  if (id >= id_end) return;
  Tile cur = tile_setup(id);
  sfor<0, NS>([&](auto sc) { issue_stage(cur, decltype(sc)::value, decltype(sc)::value * STAGE); });
  // stage 0 has landed (mine: the instructions of stages 1 .. NS - 1 may stay outstanding; everybody's: barrier)
  phase_end(VMC((NS - 1) * NDMA), LGX);

  int trace_n = 0;
  while (true) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // K loop.  "Phase t" of a wave = compute(t) for group 0, load(t) for group 1; at its end the wave's own DMA of stage
    // t + 1 must have landed (group 0 reads it right behind that barrier).  Issued by then: stages 0 .. t + PD, so PD - 1
    // stages may stay outstanding; in the last PD steps (nothing left to request) one stage less each.
    unsigned long long ts0 = 0, ts1 = 0;
    if constexpr (VAR & 32) ts0 = __builtin_amdgcn_s_memtime();
    int sb = 0, db = PD * STAGE;                    // ring offsets of stage t and of stage t + PD (= t - 1)
    auto adv = [&]() { db = sb; sb += STAGE; if (sb == NS * STAGE) sb = 0; };
    constexpr int VM_LOOP = (PD - 1) * NDMA;
    if (grp == 0) {
      // group 0: load(t) | compute(t) | ...
      load_phase(cur, 0, sb, db, F_{}); phase_end(VMC(VM_ANY), LG0); adv();
      compute_phase(); phase_end(VMC(VM_LOOP), LGX);
      int t = 1;
      for (; t + PD < nk; ++t) {
        load_phase(cur, t, sb, db, T_{}); phase_end(VMC(VM_ANY), LG0); adv();
        compute_phase(); phase_end(VMC(VM_LOOP), LGX);
      }
      sfor<0, PD>([&](auto rc) {                    // t = nk - PD + r
        constexpr int r = decltype(rc)::value;
        load_phase(cur, t + r, sb, db, F_{}); phase_end(VMC(VM_ANY), LG0); adv();
        compute_phase(); phase_end(VMC(r <= PD - 2 ? (PD - 2 - r) * NDMA : VM_ANY), LGX);
      });
    } else {
      // group 1: one phase behind: | load(t) | compute(t) ...
      phase_end(VMC(VM_ANY), LGX);
      load_phase(cur, 0, sb, db, F_{}); phase_end(VMC(VM_LOOP), LG0); adv();
      compute_phase(); phase_end(VMC(VM_ANY), LGX);
      int t = 1;
      for (; t + PD < nk; ++t) {
        load_phase(cur, t, sb, db, T_{}); phase_end(VMC(VM_LOOP), LG0); adv();
        compute_phase(); phase_end(VMC(VM_ANY), LGX);
      }
      sfor<0, PD>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        load_phase(cur, t + r, sb, db, F_{}); phase_end(VMC(r <= PD - 2 ? (PD - 2 - r) * NDMA : VM_ANY), LG0); adv();
        compute_phase();
        if constexpr (r + 1 < PD) phase_end(VMC(VM_ANY), LGX);     // (no barrier behind the last: the epilogue needs none)
      });
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue (the specialised form of gemm_s2.hip, per wave: 128 rows x 64 columns) ----
    // Host guarantees as there: N, c_ncols, pl_col0 multiples of 64, act none / GELU per E_GELU, no plane residual, no
    // cat8 output, every tensor the epilogue touches < 2^31 bytes.  Every group-1 fragment read of the K loop finished
    // before the last barrier both groups passed, so the ring is dead; the transposition piece of wave w is made of the
    // eight 1 KiB slots its OWN DMA instructions of stages 0 and 1 write -- no other wave ever touches them, and the
    // wave queues its DMA only behind its last read-back.
    if constexpr (VAR & 32) ts1 = __builtin_amdgcn_s_memtime();
    const Tile done = cur;
    id += p.nper;
    const bool more = id < id_end;
    if constexpr (VAR & 8) {
      if (more) {
        cur = tile_setup(id);
        sfor<0, NS>([&](auto sc) { issue_stage(cur, decltype(sc)::value, decltype(sc)::value * STAGE); });
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(acc[i][j]));
    } else {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    constexpr int NG = 8;
    const float alpha = d.alpha;
    const float cs = (EPI & E_PL) ? ldexpf(1.0f, RSP_PLANE_EXP(d.c_scale_log2)) : 1.0f;
    const int cols0 = done.n0 + wn * 64;
    const bool active = cols0 < N;
    const bool do_c = (EPI & E_C) && active && (d.c_ncols <= 0 || cols0 < d.c_ncols);
    const bool do_p = (EPI & E_PL) && active && cols0 >= d.pl_col0;
    const int lr0 = lane >> 4, c4 = (lane & 15) * 4;
    // piece k (rows 4k .. 4k + 3 of a 32-row pass, 256 B each) = this wave's DMA slot k % NDMA of stage k / NDMA
    const int wr_piece = l31 >> 2;
    const int wr_off = (wr_piece / NDMA) * STAGE + slot_off(wr_piece % NDMA) + (l31 & 3) * 256;
    const int rd_off = lr0 * 256;
    const auto rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.bias), 0, d.bias ? N * 4 : 0, 0x00020000);
    const f32x4 bias4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rB, (cols0 + c4) * 4, 0, 0));
    const auto rM = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(d.c_rowmap), 0, d.c_rowmap ? M * 4 : 0, 0x00020000);
    const auto rR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.res), 0, (EPI & E_RES) ? 0x7fffffff : 0, 0x00020000);
    const int row_w = done.m0 + grp * (BM / 2) + lr0;
    int crow_m[(EPI & E_RMAP) ? TM : 1][NG];
    if constexpr (EPI & E_RMAP) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int g = 0; g < NG; ++g)
          crow_m[i][g] = (int)__builtin_amdgcn_raw_buffer_load_b32(rM, (row_w + i * 32 + g * 4) * 4, 0, 0);
    }
    auto crow_of = [&](auto ic, int g) -> int {
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
        const int crc = max(crow_of(ic, g), 0);
        int rrow = crc;
        if (d.res_mod > 0) rrow = crc - p.fd_resmod.div(crc) * d.res_mod;
        rv[i & 1][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rR, (rrow * d.ldr + cols0 + c4) * 4, 0, 0));
      }
    };
    if (more) cur = tile_setup(id);
    if constexpr (EPI & E_RES) res_load(std::integral_constant<int, 0>{});
    const int64_t c_row0 = d.c_rowmap ? 0 : done.m0;
    const int c_rows_here = d.c_rowmap ? 0x7fffffff / max(d.ldc * 4, 1) : M - done.m0;
    const auto rC = __builtin_amdgcn_make_buffer_rsrc(d.C ? d.C + c_row0 * d.ldc : nullptr, 0,
                                                      do_c ? (int)min((int64_t)c_rows_here * d.ldc * 4, (int64_t)0x7fffffff) : 0, 0x00020000);
    const int pl_bytes = do_p ? (int)(((int64_t)((N - d.pl_col0) >> 5) * d.c_rows) << 6) : 0;
    const auto rH = __builtin_amdgcn_make_buffer_rsrc(d.Chi, 0, pl_bytes, 0x00020000);
    const auto rL = __builtin_amdgcn_make_buffer_rsrc(d.Clo, 0, pl_bytes, 0x00020000);
    const int pl_lane = (((cols0 - d.pl_col0) >> 5) + (c4 >> 5)) * (d.c_rows << 6) + (c4 & 31) * 2;
    sfor<0, TM>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      // (1) 32 x 64 accumulators -> LDS, row-major, 16-byte units XOR-swizzled with row & 7
      sfor<0, TN * 4>([&](auto jqc) {
        constexpr int jq = decltype(jqc)::value, j = jq / 4, q = jq % 4;
        const f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        *reinterpret_cast<f32x4*>(smem + wr_off + (((j * 8 + 2 * q + hh) ^ (l31 & 7)) << 4)) = v;
      });
      RSP_WAVE_LOCKSTEP();
      // (2) read back: 4 rows x 64 columns per instruction (group g = rows 4g .. 4g + 3 = piece g)
      f32x4 x[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int lr = g * 4 + lr0;
        x[g] = *reinterpret_cast<const f32x4*>(smem + (g / NDMA) * STAGE + slot_off(g % NDMA) + rd_off + (((c4 >> 2) ^ (lr & 7)) << 4));
      }
      RSP_WAVE_LOCKSTEP();
      if constexpr (i + 1 == TM) {
        asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
      }
      // (3) value = act(acc * alpha + bias) + residual
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        x[g] = x[g] * alpha + bias4;
        if constexpr (EPI & E_GELU) x[g] = rsp_gelu4(x[g]);
        if constexpr (EPI & E_RES) x[g] += rv[i & 1][g];
      }
      if constexpr (i + 1 < TM) {
        if constexpr (EPI & E_RES) res_load(std::integral_constant<int, i + 1>{});
      }
      // (4) stores
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int cr = crow_of(ic, g);
        if constexpr (EPI & E_C) {
          const unsigned co = cr < 0 ? OOB : (unsigned)(((cr - (int)c_row0) * d.ldc + cols0 + c4) * 4);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x[g]), rC, co, 0, 0);
        }
        if constexpr (EPI & E_PL) {
          half4_t h4, l4;
          f32x4 rem;
          rsp_split4(x[g] * cs, h4, l4, rem);
          const unsigned po = cr < 0 ? OOB : (unsigned)(cr * 64 + pl_lane);
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, h4), rH, po, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, l4), rL, po, 0, 0);
        }
      }
      if constexpr (i + 1 == TM) {
        // every loaded value has been consumed, the piece has been read back, the tile's last stores are queued (OLDER
        // than the DMA below: vmcnt retires in order, the K loop's counts are DMA instructions only): this wave's part
        // of the next tile's first four stages
        __builtin_amdgcn_sched_barrier(0);
        if (more) sfor<0, NS>([&](auto sc) { issue_stage(cur, decltype(sc)::value, decltype(sc)::value * STAGE); });
      }
    });
    }
    if constexpr (VAR & 32) {
      if (p.trace && (tid & 255) == 0 && trace_n < 16) {
        unsigned long long* t = p.trace + (((size_t)blockIdx.x * 16 + trace_n) * 2 + grp) * 4;
        t[0] = ts0; t[1] = ts1; t[2] = __builtin_amdgcn_s_memtime();
        t[3] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
      }
      ++trace_n;
    }
    if (!more) break;
    // stage 0 of the next tile: mine has landed (the younger DMA instructions may stay outstanding), everybody's: barrier.  The barrier also re-aligns the groups (group 0 came here one phase early).
    __builtin_amdgcn_sched_barrier(0);
    phase_end(VMC((NS - 1) * NDMA), LG0);
  }
#undef VMC
#undef LG0
#undef LGX
}

template <int EPI, int VAR = 0, int BM = 256>
int launch_pp(const RspGemmDesc& d, hipStream_t s) {
  PPP p; p.d = d;
  p.fd_resmod = make_fastdiv(d.res_mod);
  p.nbm = (d.M + BM - 1) / BM; p.nbn = (d.N + BN - 1) / BN;
  const long long nt = (long long)p.nbm * p.nbn;
  if (nt > 0x3fffffffLL) return RSP_EINVAL;
  p.ntiles = (int)nt;
  p.per_xcd = (p.ntiles + 7) / 8;
  p.group_m = (d.tile_hint >> 8) & 0xff;
  if (p.group_m == 0) p.group_m = p.nbn <= 5 ? 4 : 8;   // 32 concurrent tiles per XCD = 8 M-tiles x 4 N-tiles (narrow N: 4 x all)
  int cap = (d.tile_hint >> 16) & 0xff;          // tests: fewer blocks per XCD (several tiles per block on small shapes)
  if (cap == 0 || cap > 32) cap = 32;            // blocks per XCD: one per CU
  p.nper = p.per_xcd < cap ? p.per_xcd : cap;
  p.trace = g_pp_trace;
  hipLaunchKernelGGL((gemm_f16x3_pp_kernel<EPI, VAR, BM>), dim3((unsigned)(8 * p.nper)), dim3(NTHR), 0, s, p);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

}  // namespace

bool rsp_gemm_s2_eligible(const RspGemmDesc& d);                       // gemm_s2.hip
int rsp_gemm_s2_epilogue_of(const RspGemmDesc& d);

// the descriptors this kernel implements: gemm_s2.hip's, with one of its compile-time epilogues, K >= 128 (the K loop is
// peeled by up to five steps at its end)
bool rsp_gemm_pp_eligible(const RspGemmDesc& d) {
  if (!rsp_gemm_s2_eligible(d) || d.K < 128) return false;
  const int e = rsp_gemm_s2_epilogue_of(d);
  return e >= 0 && e != 64;
}

// product rule: 0 = not this kernel; 256 / 128 = the block tile's rows.  One block per CU and a static walk: a ragged last
// round idles whole CUs, and the exposed epilogue wants a long K loop.  Measured on the ViT-B / L / H encoder shapes
// (profiles/r5_gemm_pp/pp_{base,large,huge}.txt, TFLOP/s against gemm_s2): 256 x 256 wins wherever its tiles fill >= 2
// rounds of the 256 CUs to >= 80 % (+6 ... +16 %; +1 ... 3 % at 2.5 rounds); 128 x 256 (phases of 12 MFMAs: 440 instead of 384
// cycles each) only pays for residual GEMMs whose 256-row tiles do not fill the chip (ViT-B proj / lin2: +3 ... 9 %).
int rsp_gemm_pp_auto(const RspGemmDesc& d) {
  if (!rsp_gemm_pp_eligible(d) || d.K < 512) return 0;
  const long long nbn = (d.N + BN - 1) / BN;
  const long long nt256 = (long long)((d.M + 255) / 256) * nbn, r256 = (nt256 + 255) / 256;
  if (nt256 >= 512 && nt256 * 100 >= r256 * 256 * 80) return 256;
  const long long nt128 = (long long)((d.M + 127) / 128) * nbn, r128 = (nt128 + 255) / 256;
  if (d.res && nt128 >= 512 && nt128 * 100 >= r128 * 256 * 90) return 128;
  return 0;
}

// var: bit 0 = the 128-row tile; development builds: bit 2 = no DMA in the K loop, bit 3 = no epilogue, bit 5 = time stamps
int rsp_gemm_pp_dispatch(const RspGemmDesc& d, int var, hipStream_t s) {
  if (!rsp_gemm_pp_eligible(d)) return RSP_EINVAL;
  const int epi = rsp_gemm_s2_epilogue_of(d);
#ifdef RSP_S2_ABLATIONS          /* RSP_DEV_BUILD=1 python -m rsprompter_amd.build; tools/gemm_pp_exp.py */
#define PP_VCASE(V, E) if (var == V && epi == (E)) return launch_pp<(E), V, 256>(d, s); \
                       if (var == V + 1 && epi == (E)) return launch_pp<(E), V, 128>(d, s)
  PP_VCASE(4, E_PL | E_GELU); PP_VCASE(32, E_PL | E_GELU); PP_VCASE(4, E_C | E_RES); PP_VCASE(32, E_C | E_RES);
  PP_VCASE(32, E_C | E_PL);
#undef PP_VCASE
#endif
  if (var & ~1) return RSP_EINVAL;
#define PP_CASE(E) if (epi == (E)) return (var & 1) ? launch_pp<(E), 0, 128>(d, s) : launch_pp<(E), 0, 256>(d, s)
  PP_CASE(E_C);
  PP_CASE(E_C | E_RES);
  PP_CASE(E_C | E_RES | E_RMAP);
  PP_CASE(E_C | E_PL);
  PP_CASE(E_C | E_PL | E_RMAP);
  PP_CASE(E_PL);
  PP_CASE(E_PL | E_GELU);
  PP_CASE(E_C | E_GELU);
#undef PP_CASE
  return RSP_EINVAL;
}

#ifdef RSP_S2_ABLATIONS
// tools only (not part of include/rsp_hip.h): device buffer [256 blocks][16 tiles][2 groups][4] u64 for the time-stamp variant
extern "C" void rsp_debug_pp_trace(void* p) { g_pp_trace = reinterpret_cast<unsigned long long*>(p); }
#endif
