// SAM ViT WINDOW attention (14 x 14 windows, HF:803-831 + get_decomposed_rel_pos HF:761-801, windows HF:900-952), round 4.
// Successor of the WINDOW instantiation of attn_stream.hip (same plane-fed operands, same DMA ring, same transposing V
// reads -- see that file's header); what changed is everything that kernel spent its VALU issue slots on (PMC r3: 11.0 VALU
// instructions per MFMA, matrix pipe busy 21 %):
//
//   * the decomposed rel-pos bias enters through the MATRIX CORES.  bias[k, q] = rel_h[q, kh(k)] + rel_w[q, kw(k)] is the
//     product of a one-hot matrix E[k, k'] (k' = kh for k' < 14, 14 + kw for 14 <= k' < 28, k' = 28 marks the padded keys
//     196..223 of the last tile) with the query's 28 rel-pos terms B[k', q] (B[28, q] = a large negative number): two
//     16-k steps x (B_hi, B_lo) = 4 MFMAs per 32-key tile on the score accumulator instead of, per score, two adds, a
//     half-wave select and the validity select.  E lives in LDS as ready A fragments (a constant of the kernel, 17.5 KB).
//   * the rel-pos terms themselves are computed IN the kernel (rsp_vit_window_attention): G[idx, q] = R_all[idx, :] . q
//     for both tables on the matrix cores with the attention's own Q fragments (tables pre-split into fp16 hi / lo planes
//     once per layer, rsp_pack_relpos_tables, DMA'd into LDS), then the Toeplitz band rel_h[q, kh] = G_h[qy - kh + 13, q]
//     through a per-wave LDS piece.  The stand-alone vit_relpos_win_kernel, its [Bp*nh, T, 28] HBM round trip and 2.8 ms
//     per ViT-H step are gone from the product path (the rel tensor form stays behind rsp_vit_attention_planes_ex).
//   * LAZY online softmax: the running maximum moves only when a tile's maximum exceeds it by more than 2^8, so
//     probabilities stay below 2^8 (they travel as fp16 hi / lo scaled by 2^7) and the 48-register accumulator rescale
//     happens about once per window instead of in nearly every tile; tile maxima through v_max3.
//   Per score that leaves: fma + exp2 + the 2-instruction truncating P split (+ 0.5 for the maximum) -- about 2.5 VALU
//   instructions per MFMA.
#include <type_traits>
#include "rsp_common.h"

namespace {

constexpr int EQ = 6;                         // q * scale * 2^EQ before its fp16 split (k / v planes carry their own)
constexpr int RT = 6;                         // rel-pos tables * 2^RT before their fp16 split (rsp_pack_relpos_tables)
constexpr float P_SCALE_LOG2 = 7.0f;          // probabilities (< 2^8, see LAZY_LOG2) are scaled by 2^7 before the fp16 split
constexpr float LAZY_LOG2 = 8.0f;             // the running maximum follows a tile maximum only beyond this distance
constexpr float LOG2E_C = 1.4426950408889634f;
constexpr float NEG_RAW = -60000.0f;          // bias of a padded key in the raw score domain (exp2 underflows to 0)
constexpr int WT = 196, WS = 14, KT = 32, NTILE = 7;
// ring of 4 buffers, tile kt in buffer kt % 4, worked on in pairs (one barrier per two tiles): while tiles 2p, 2p + 1 are
// multiplied, tiles 2p + 2, 2p + 3 land in the other two buffers; the next window's tiles 0, 1 are queued behind tile 6
// into buffers 0, 1 -- every window starts at buffer 0 (compile-time addresses)
constexpr int NBUF = 4;
constexpr int OH_PITCH = 80;                  // bytes per key row of the one-hot table (32 halves + pad: conflict-free b128)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef short v4s __attribute__((ext_vector_type(4)));

__device__ uint4 g_zero16w[4];                // zero page: padded V chunks, keys beyond the window

// the one-hot matrix E of the header as ready A fragments: row = key 0..223 (pitch OH_PITCH bytes = 40 halves), halves
// k' = 0..31; a compile-time constant of the library, DMA'd into LDS by every block
struct OneHotTab {
  uint16_t v[NTILE * KT * (OH_PITCH / 2)];
  constexpr OneHotTab() : v{} {
    for (int key = 0; key < NTILE * KT; ++key)
      for (int kp = 0; kp < 32; ++kp) {
        const int kh = key / WS, kw = key % WS;
        const bool one = key < WT ? (kp == kh || kp == WS + kw) : kp == 2 * WS;
        v[key * (OH_PITCH / 2) + kp] = one ? 0x3C00 : 0;          // fp16 1.0
      }
  }
};
__device__ const OneHotTab g_onehot{};
// dh = 80: the chunk behind the real ones of a V row carries 1.0 at d = 80 and d = 84 (hi plane), so accumulator
// register 8 of the third V^T block is sum_k P[k] in both half waves (attn_stream.hip)
__device__ const _Float16 g_ones16w[8] = {(_Float16)1.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f,
                                          (_Float16)1.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};

__device__ __forceinline__ void split8w(const float* x, half8_t& hi, half8_t& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    half_t h, l;
    rsp_split1(x[i], h, l);
    hi[i] = h; lo[i] = l;
  }
}

// values inside the fp16 range: truncating split, 2 VALU per element (attn_stream.hip split8_fast)
__device__ __forceinline__ void split8w_fast(const float* x, half8_t& hi, half8_t& lo) {
  float m1 = -1.0f;                               // opaque multiplier: x - hi as ONE v_fma_mix_f32 (attn_stream.hip)
  asm volatile("" : "+v"(m1));
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    const half2_t h2 = __builtin_bit_cast(half2_t, __builtin_amdgcn_cvt_pkrtz(x[i], x[i + 1]));
    const float r0 = __builtin_fmaf((float)h2[0], m1, x[i]), r1 = __builtin_fmaf((float)h2[1], m1, x[i + 1]);
    const half2_t l2 = __builtin_bit_cast(half2_t, __builtin_amdgcn_cvt_pkrtz(r0, r1));
    hi[i] = h2[0]; hi[i + 1] = h2[1]; lo[i] = l2[0]; lo[i + 1] = l2[1];
  }
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for_w(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for_w<I + 1, N>(f);
  }
}

struct AttnWP {
  const float* q; int64_t q_ld;                 // fp32 [Bp*196, q_ld]: head h at columns [h*DH, (h+1)*DH)
  const half_t* kv_hi; const half_t* kv_lo;     // KB32 planes of the [Bp*196, 2 D] matrix (K | V), value * 2^kv_e
  int64_t kv_rows;
  int kv_e;
  const float* rel;                             // REL_IN: [Bp*nh, 196, 28] rel-pos terms (rsp_vit_relpos_*)
  const half_t* rel_tab;                        // else: [2 tables][hi, lo][32][DH + 8] fp16 (rsp_pack_relpos_tables)
  float* out; half_t* out_hi; half_t* out_lo; float out_pscale; int64_t out_rows;
  bool out_f8;
  int nh, D;
  float scale;
  int win_n, win_real;                          // windows per image side / real rows (columns) of the last one; 0: unknown
};

template <int DH, bool REL_IN>
__global__ __launch_bounds__(448) void attn_win_kernel(const AttnWP p, const int n_items) {
  constexpr int NW = 7, NT = NW * 64;
  constexpr int DSTEPS = DH / 16;
  constexpr int DBLK = (DH + 31) / 32;
  constexpr int KCH = DH / 8;                         // real 16-byte chunks per K / V row
  constexpr bool KSWZ = (KCH == 8);                   // dh = 64: 128-byte K rows with an XOR swizzle
  constexpr int KCPR = KSWZ ? 8 : KCH + 1;            // dh = 80: 11 chunks = 176-byte pitch
  constexpr int VCPR = 12;                            // V rows: 192-byte pitch
  constexpr int K_UNITS = KT * KCPR, V_UNITS = KT * VCPR;
  constexpr int TILE_UNITS = 2 * K_UNITS + 2 * V_UNITS;
  constexpr int NDMA = (TILE_UNITS + NT - 1) / NT;
  constexpr int BUF_BYTES = (TILE_UNITS * 16 + 1023) / 1024 * 1024;
  constexpr bool LSUM_MFMA = (DH % 32) != 0;          // spare V^T rows exist: row sums through the MFMA
  constexpr int LDT = DH + 8;                         // halves per rel-pos table row in LDS (conflict-free b128 reads)
  constexpr int OH_OFF = NBUF * BUF_BYTES, OH_BYTES = NTILE * KT * OH_PITCH;
  constexpr int TAB_OFF = OH_OFF + OH_BYTES, TAB_BYTES = REL_IN ? 0 : 2 * 2 * 32 * LDT * 2;
  constexpr int TAB_UNITS = TAB_BYTES / 16, NTAB = (TAB_UNITS + NT - 1) / NT;
  constexpr int OH_UNITS = OH_BYTES / 16, NOH = (OH_UNITS + NT - 1) / NT;
  // per-wave piece [query][table row 0..26], odd pitch (conflict-free); 4 ring buffers + tables + pieces = 157 KB at dh = 80
  constexpr int PIECE_OFF = TAB_OFF + TAB_BYTES, PIECE_LD = 29, PIECE_BYTES = REL_IN ? 0 : 32 * PIECE_LD * 4;
  // ONE LDS object (a second __shared__ variable makes hipcc drain the DMA queue before every fragment read)
  __shared__ __attribute__((aligned(1024))) unsigned char smem[PIECE_OFF + NW * PIECE_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, l31 = lane & 31;
  const int nh = p.nh;
  // PERSISTENT: gridDim.x is a multiple of nh, so a block keeps its head h = blockIdx.x % nh and walks the windows
  // bp0, bp0 + gridDim.x / nh, ...  The one-hot table and the rel-pos tables are loaded once; the K | V tile stream never
  // stops at a window boundary (tiles 0, 1 of the next window are queued behind tiles 5, 6 of the current one; every real
  // DMA source simply advances by a constant) and the next window's q rows are requested before the current one's results
  // are stored.  A block per (window, head) paid ~8k cycles of cold DMA (87 KB at the ~11 B/clk a CU's prologue burst
  // gets) before its first MFMA, and with one block per CU nothing hid it.
  const int h = (int)(blockIdx.x % (unsigned)nh);
  const int bp_step = (int)(gridDim.x / (unsigned)nh);
  const int n_win = n_items / nh;
  int bp = (int)(blockIdx.x / (unsigned)nh);
  if (bp >= n_win) return;

  const rsp_lds_addr_t smem_a = rsp_lds_addr((lptr_t)smem);   // wave-uniform LDS address of the block's image (DMA destinations)
  // ---- the one-hot table and the rel-pos tables of this layer: DMA first (oldest in the queue) ----
  {
    const unsigned char* osrc = reinterpret_cast<const unsigned char*>(g_onehot.v);
#pragma unroll
    for (int i = 0; i < NOH; ++i)
      if ((i + 1) * NT <= OH_UNITS || i * NT + tid < OH_UNITS)
        RSP_GLOBAL_LOAD_LDS_B128(osrc + (size_t)(i * NT + tid) * 16, smem_a + OH_OFF + (i * NT + wave * 64) * 16);
  }
  if constexpr (!REL_IN) {
    const unsigned char* tsrc = reinterpret_cast<const unsigned char*>(p.rel_tab);
#pragma unroll
    for (int i = 0; i < NTAB; ++i)
      if ((i + 1) * NT <= TAB_UNITS || i * NT + tid < TAB_UNITS)
        RSP_GLOBAL_LOAD_LDS_B128(tsrc + (size_t)(i * NT + tid) * 16, smem_a + TAB_OFF + (i * NT + wave * 64) * 16);
  }

  // ---- per-thread DMA slots: unit u = i*NT + tid of the tile image [K_hi | K_lo | V_hi | V_lo]; item-independent part ----
  const unsigned char* dsrc[NDMA];                    // source of the slot in tile 0 of the CURRENT window
  int drow[NDMA];                                     // key row inside the tile; -1: zero page, -2: the ones chunk
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(g_zero16w);
#pragma unroll
  for (int i = 0; i < NDMA; ++i) {
    const int u = i * NT + tid;
    dsrc[i] = zero; drow[i] = -1;
    int pl = 0, row = 0, c = 0, colbase = 0;
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
      dsrc[i] = reinterpret_cast<const unsigned char*>(g_ones16w);      // hi plane, first padding chunk: the ones
      drow[i] = -2;                                                     // constant source, valid for every tile
    }
    if (real) {
      const int col = colbase + h * DH + c * 8;
      const half_t* base = pl == 0 ? p.kv_hi : p.kv_lo;
      dsrc[i] = reinterpret_cast<const unsigned char*>(base + ((int64_t)(col >> 5) * p.kv_rows + (int64_t)bp * WT + row) * 32 + (col & 31));
      drow[i] = row;
    }
  }
  const int64_t win_delta = (int64_t)bp_step * WT * 64;     // bytes between the K | V rows of consecutive windows of a block
  // tile kt of the current window (ahead = 0) or of the block's next one (ahead = 1) into ring buffer buf
  auto issue_tile = [&](int kt, int buf, int ahead) {
    const rsp_lds_addr_t lbase = smem_a + buf * BUF_BYTES;
    const int64_t adv = (int64_t)kt * (KT * 64) + (ahead ? win_delta : 0);
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      const bool ok = drow[i] >= 0 && kt * KT + drow[i] < WT;
      const unsigned char* src = ok ? dsrc[i] + adv : (drow[i] == -2 ? dsrc[i] : zero);
      if ((i + 1) * NT <= TILE_UNITS || i * NT + tid < TILE_UNITS)     // the last instruction may be partly masked
        // (as inline assembly, round 6: hipcc models the builtin as a write to LDS and drained the burst it had just let this
        // wave issue -- s_waitcnt vmcnt(0) in front of the second tile's fragment reads, four times per window: csrc/rsp_common.h)
        RSP_GLOBAL_LOAD_LDS_B128(src, lbase + (i * NT + wave * 64) * 16);
    }
  };
  issue_tile(0, 0, 0);
  issue_tile(1, 1, 0);

  // ---- per item: which query this lane owns.  Windows cut from a padded grid (HF:900-922): the windows of the last
  // row / column hold only win_real real rows / columns; their padded tokens are keys like any other (k = v = bias) but
  // nobody reads their outputs, so the real queries are packed into the first waves and the remaining waves only keep the
  // block's barriers and DMA slots going.
  struct Item { int q; bool qv, dead; };
  auto item_setup = [&](int bp_) {
    Item c;
    int q = wave * 32 + l31;
    bool qv = q < WT, dead = wave * 32 >= WT;
    if (p.win_n > 0) {
      const int wi = bp_ % (p.win_n * p.win_n);
      const int wy = wi / p.win_n, wx = wi - wy * p.win_n;
      const int rh = wy == p.win_n - 1 ? p.win_real : WS, cw = wx == p.win_n - 1 ? p.win_real : WS;
      const int cc = wave * 32 + l31, cy = cc / cw;
      qv = cc < rh * cw;
      q = cy * WS + (cc - cy * cw);
      dead = wave * 32 >= rh * cw;                   // wave-uniform
    }
    c.q = qv ? q : 0; c.qv = qv; c.dead = dead;
    return c;
  };
  // the lane's q slice as it lies in memory: 2 x 16 bytes per 16-d step
  auto load_q = [&](int bp_, const Item& c, f32x4* qa, f32x4* qb) {
    const float* q_b = p.q + ((int64_t)bp_ * WT + c.q) * p.q_ld + (int64_t)h * DH;
#pragma unroll
    for (int st = 0; st < DSTEPS; ++st) {
      qa[st] = *reinterpret_cast<const f32x4*>(q_b + st * 16 + hh * 8);
      qb[st] = *reinterpret_cast<const f32x4*>(q_b + st * 16 + hh * 8 + 4);
    }
  };
  Item cur = item_setup(bp);
  f32x4 qa[DSTEPS], qb[DSTEPS];                      // raw q of the current window; the next one's is requested behind tile 6
  load_q(bp, cur, qa, qb);

  const float c2 = ldexpf(1.0f, -(EQ + p.kv_e)) * LOG2E_C;   // raw score -> log2 domain
  const float lazy_raw = LAZY_LOG2 / c2;
  // per-lane byte offsets of the transposing V reads: row = 4 hh + (i >> 2), d = 16 g16 + 4 (i & 3)
  const int li = lane & 15, g16 = (lane >> 4) & 1;
  const int v_lane_off = (4 * hh + (li >> 2)) * (VCPR * 16) + (16 * g16 + 4 * (li & 3)) * 2;
  const unsigned char* oh_lane = smem + OH_OFF + l31 * OH_PITCH + hh * 16;

  // every wave's DMA pieces of the tables have landed (a workgroup-scope barrier alone does not drain vmcnt)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  while (true) {
    const int bp_next = bp + bp_step;
    const bool has_next = bp_next < n_win;            // block-uniform
    const int q = cur.q;
    const bool qv = cur.qv, dead = cur.dead;
    const int64_t row0 = (int64_t)bp * WT;            // first row of this window in q, planes, out

    // ---- Q fragments (B operand of S^T = K Q^T and of the rel-pos products), scaled and split once ----
    half8_t qh[DSTEPS], qlo[DSTEPS];
    half8_t bbh[2], bbl[2];
    if (!dead) {
      const float qs = p.scale * ldexpf(1.0f, EQ);
#pragma unroll
      for (int st = 0; st < DSTEPS; ++st) {
        float x[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) { x[i] = qv ? qa[st][i] * qs : 0.f; x[4 + i] = qv ? qb[st][i] * qs : 0.f; }
        split8w_fast(x, qh[st], qlo[st]);       // truncating conversions saturate at the fp16 maximum (never inf)
      }
      // ---- the query's 28 rel-pos terms as B fragments of the bias product: lane (q, hh), step s holds
      // k' = 16 s + 8 hh + e.  Raw score domain: K planes carry 2^kv_e, q carries scale * 2^EQ, so a logit t appears
      // as t * 2^(EQ + kv_e).
      float vals[16];
#pragma unroll
      for (int n = 0; n < 16; ++n) vals[n] = 0.f;
      if constexpr (REL_IN) {
        const float* rq = p.rel + (((int64_t)bp * nh + h) * WT + q) * (2 * WS);   // 112-byte rows: 16-byte aligned
        const float bs = ldexpf(1.0f, EQ + p.kv_e);
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(rq + 8 * hh), a1 = *reinterpret_cast<const f32x4*>(rq + 8 * hh + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(rq + 16 + 8 * hh);       // hh = 1: k' = 24..27
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(rq + 16 + 4);            // hh = 0 only: k' = 20..23
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          vals[e] = qv ? a0[e] * bs : 0.f;
          vals[4 + e] = qv ? a1[e] * bs : 0.f;
          vals[8 + e] = qv ? b0[e] * bs : 0.f;
          vals[12 + e] = (qv && hh == 0) ? b1[e] * bs : 0.f;
        }
      } else {
        const int qy = q / WS, qx = q - qy * WS;
        // G_raw = sum_d (R 2^RT)(q scale 2^EQ); bias_raw = rel 2^(EQ + kv_e) = G_raw 2^(kv_e - RT) / scale
        const float gs = ldexpf(1.0f, p.kv_e - RT) / p.scale;
        float* piece = reinterpret_cast<float*>(smem + PIECE_OFF + wave * PIECE_BYTES);
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
          const half_t* th_ = reinterpret_cast<const half_t*>(smem + TAB_OFF) + (tb * 2 + 0) * 32 * LDT;
          const half_t* tl_ = reinterpret_cast<const half_t*>(smem + TAB_OFF) + (tb * 2 + 1) * 32 * LDT;
          f32x16 g;
#pragma unroll
          for (int r = 0; r < 16; ++r) g[r] = 0.f;
#pragma unroll
          for (int st = 0; st < DSTEPS; ++st) {
            const int off = l31 * LDT + st * 16 + hh * 8;
            const half8_t th8 = *reinterpret_cast<const half8_t*>(th_ + off);
            const half8_t tl8 = *reinterpret_cast<const half8_t*>(tl_ + off);
            g = __builtin_amdgcn_mfma_f32_32x32x16_f16(tl8, qh[st], g, 0, 0, 0);
            g = __builtin_amdgcn_mfma_f32_32x32x16_f16(th8, qlo[st], g, 0, 0, 0);
            g = __builtin_amdgcn_mfma_f32_32x32x16_f16(th8, qh[st], g, 0, 0, 0);
          }
          // lane (q, hh) holds G[idx = (r & 3) + 8 (r >> 2) + 4 hh][q]: through the wave's piece [q][idx], then the band
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int idx = (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (idx < 2 * WS - 1) piece[l31 * PIECE_LD + idx] = g[r] * gs;     // (rows 27..31 of the tables are padding)
          }
          RSP_WAVE_LOCKSTEP();                          // a query's row comes from both half waves (lanes q and q + 32)
          const int pos = (tb ? qx : qy) + WS - 1;
#pragma unroll
          for (int n = 0; n < 16; ++n) {
            const int kp = 16 * (n >> 3) + 8 * hh + (n & 7);
            const int j = kp - tb * WS;                 // kh (table 0) / kw (table 1)
            if (j >= 0 && j < WS) vals[n] = piece[l31 * PIECE_LD + pos - j];
          }
          RSP_WAVE_LOCKSTEP();
        }
#pragma unroll
        for (int n = 0; n < 16; ++n) vals[n] = qv ? vals[n] : 0.f;
      }
      if (hh == 1) vals[12] = NEG_RAW;                  // k' = 28: the padded keys of the last tile
      split8w_fast(vals, bbh[0], bbl[0]);
      split8w_fast(vals + 8, bbh[1], bbl[1]);
    }

    f32x16 acc_o[DBLK];
#pragma unroll
    for (int db = 0; db < DBLK; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc_o[db][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;               // m_run in the RAW score domain

    // A tile = phase Q (raw scores: 19 MFMAs) + phase SP (softmax on the VALU, then P V: 18 MFMAs).
    // (Measured and dropped, round 4: running the waves 4..6 -- the SIMD partners of waves 0..2 -- one phase late, so that
    // one wave's softmax sits beside its partner's matrix phase: no gain at dh = 64, and at dh = 80 the scores held across
    // the barrier pushed the kernel into scratch.)
    f32x16 sc;                                             // scores -> probabilities of the tile in flight

    typedef __attribute__((address_space(3))) v4s* lv4;
    // V^T fragments of 16-key step s_, d block db of the tile in ring buffer `buf` (transposing LDS reads)
    auto vread = [&](int buf, int s_, int db, v4s* f) {
      const unsigned char* sV0 = smem + buf * BUF_BYTES + 2 * K_UNITS * 16;
      const unsigned char* sV1 = sV0 + V_UNITS * 16;
      const int off = v_lane_off + (16 * s_) * (VCPR * 16) + db * 64;      // bytes: key 16 s (+ 8), d block db
      f[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lv4)(sV0 + off));
      f[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lv4)(sV0 + off + 8 * (VCPR * 16)));
      f[2] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lv4)(sV1 + off));
      f[3] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lv4)(sV1 + off + 8 * (VCPR * 16)));
    };
    v4s va[2][4];                                          // [buffer][hi k0..3, hi k8..11, lo k0..3, lo k8..11]

    // raw scores S^T = E B + K Q^T: bias product first (constant operands), K fragments register-pipelined one step ahead.
    // (Measured and dropped, round 4: all ten K fragment reads issued before the DMA issue, first V reads before the
    // softmax -- 0.349 vs 0.324 ms per ViT-H layer.)
    auto phase_q = [&](auto tc) {
      constexpr int kt = decltype(tc)::value, buf = kt % NBUF;
      const half8_t oh0 = *reinterpret_cast<const half8_t*>(oh_lane + kt * KT * OH_PITCH);
      const half8_t oh1 = *reinterpret_cast<const half8_t*>(oh_lane + kt * KT * OH_PITCH + 32);
      const half_t* sK0 = reinterpret_cast<const half_t*>(smem + buf * BUF_BYTES);
      const half_t* sK1 = reinterpret_cast<const half_t*>(smem + buf * BUF_BYTES + K_UNITS * 16);
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[r] = 0.f;
      half8_t kfh[2], kfl[2];
      auto kread = [&](int st, half8_t& h8, half8_t& l8) {
        const int c = st * 2 + hh;
        const int off = l31 * (KCPR * 8) + ((KSWZ ? (c ^ ((l31 >> 1) & 7)) : c) << 3);     // halves
        h8 = *reinterpret_cast<const half8_t*>(sK0 + off);
        l8 = *reinterpret_cast<const half8_t*>(sK1 + off);
      };
      kread(0, kfh[0], kfl[0]);
      __builtin_amdgcn_sched_barrier(0);
      sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(oh0, bbl[0], sc, 0, 0, 0);
      sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(oh0, bbh[0], sc, 0, 0, 0);
      sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(oh1, bbl[1], sc, 0, 0, 0);
      sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(oh1, bbh[1], sc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      static_for_w<0, DSTEPS>([&](auto ic) {
        constexpr int st = decltype(ic)::value, cur_ = st & 1;
        if constexpr (st + 1 < DSTEPS) kread(st + 1, kfh[cur_ ^ 1], kfl[cur_ ^ 1]);
        __builtin_amdgcn_sched_barrier(0);        // keep the next step's reads AHEAD of this step's matrix work
        sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfl[cur_], qh[st], sc, 0, 0, 0);
        sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[cur_], qlo[st], sc, 0, 0, 0);
        sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[cur_], qh[st], sc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      });
    };

    auto phase_sp = [&](auto tc) {
      constexpr int kt = decltype(tc)::value, buf = kt % NBUF;
      // ---- lazy online softmax (per-lane query column; the two half waves hold the two halves of the tile's keys) ----
      float tmax = fmaxf(fmaxf(sc[0], sc[1]), sc[2]);
#pragma unroll
      for (int r = 3; r < 15; r += 2) tmax = fmaxf(fmaxf(tmax, sc[r]), sc[r + 1]);
      tmax = fmaxf(tmax, sc[15]);
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      const bool upd = tmax > m_run + lazy_raw;            // first tile: m_run = -inf, always
      const float m_new = upd ? tmax : m_run;
      const float kk = fmaf(-m_new, c2, P_SCALE_LOG2);
      float psum = 0.f;
      {
        f32x16 c2v, kkv;
#pragma unroll
        for (int r = 0; r < 16; ++r) { c2v[r] = c2; kkv[r] = kk; }
        sc = __builtin_elementwise_fma(sc, c2v, kkv);       // 8 v_pk_fma_f32
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        sc[r] = __builtin_amdgcn_exp2f(sc[r]);
        if constexpr (!LSUM_MFMA) psum += sc[r];
      }
      if constexpr (kt > 0) {
        if (__builtin_amdgcn_ballot_w64(upd) != 0) {       // about once per window after the first tile
          const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);   // 1 for the lanes that keep their maximum
#pragma unroll
          for (int db = 0; db < DBLK; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_o[db][r] *= alpha;
          if constexpr (!LSUM_MFMA) l_run *= alpha;
        }
      }
      if constexpr (!LSUM_MFMA) l_run += psum;
      m_run = m_new;

      // ---- O^T += V^T P^T : V^T fragments through the transposing LDS read, register-pipelined ----
      constexpr int NS = KT / 16, NSTEP = NS * DBLK;
      vread(buf, 0, 0, va[0]);
      half8_t ph, pl;
      static_for_w<0, NSTEP>([&](auto ic) {
        constexpr int i = decltype(ic)::value, s_ = i / DBLK, db = i % DBLK, cur_ = i & 1;
        if constexpr (db == 0) {
          float pf[8];
#pragma unroll
          for (int t = 0; t < 8; ++t) pf[t] = sc[8 * s_ + t];
          split8w_fast(pf, ph, pl);
        }
        if constexpr (i + 1 < NSTEP) vread(buf, (i + 1) / DBLK, (i + 1) % DBLK, va[cur_ ^ 1]);
        __builtin_amdgcn_sched_barrier(0);
        union { v4s s4[2]; half8_t h8; } uh, ul;
        uh.s4[0] = va[cur_][0]; uh.s4[1] = va[cur_][1]; ul.s4[0] = va[cur_][2]; ul.s4[1] = va[cur_][3];
        acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ul.h8, ph, acc_o[db], 0, 0, 0);
        acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(uh.h8, pl, acc_o[db], 0, 0, 0);
        acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(uh.h8, ph, acc_o[db], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      });
    };

    // Two tiles per barrier: behind barrier #p (p = 0..3) the waves run tiles 2p and 2p + 1 (the 7th tile stands alone)
    // and queue the two tiles after those -- the next window's first two behind tile 6 -- into the buffers that tiles
    // 2p - 2 and 2p - 1 left before the barrier.  Everything a wave waits for at a barrier was queued a whole pair earlier
    // (vmcnt(0): nothing younger is in flight), and the waves meet 4 times per window instead of 7 (PMC, one tile per
    // barrier: the waves stood at barriers / waits for half of their cycles).
    auto pair_body = [&](auto pc) {
      constexpr int pr = decltype(pc)::value, t0 = 2 * pr, t1 = 2 * pr + 1;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's parts of tiles t0 and t1 have landed
      __builtin_amdgcn_s_barrier();                        // ... everybody's; tiles t0 - 2 and t0 - 1 are drained
      constexpr bool last = t0 + 1 >= NTILE;               // the pair of the window's last tile: the next window's turn
      if constexpr (t0 + 2 < NTILE) issue_tile(t0 + 2, (t0 + 2) % NBUF, 0);
      else if constexpr (last) { if (has_next) issue_tile(0, 0, 1); }
      if (!dead) phase_q(std::integral_constant<int, t0>{});
      if constexpr (t0 + 3 < NTILE) issue_tile(t0 + 3, (t0 + 3) % NBUF, 0);
      else if constexpr (last) { if (has_next) issue_tile(1, 1, 1); }
      if (dead) return;                                    // no real query in this wave
      phase_sp(std::integral_constant<int, t0>{});
      if constexpr (t1 < NTILE) {
        phase_q(std::integral_constant<int, t1>{});
        phase_sp(std::integral_constant<int, t1>{});
      }
    };

    static_for_w<0, (NTILE + 1) / 2>([&](auto pc) { pair_body(pc); });

    // the next window's q rows travel while this one's results are normalised and stored (qh / qlo are dead by now)
    Item nxt = cur;
    if (has_next) {
      nxt = item_setup(bp_next);
      load_q(bp_next, nxt, qa, qb);
    }

    // ---- normalise and store: lane holds O[q][d], d = db*32 + (r&3) + 8*(r>>2) + 4*hh ----
    const float l_tot = LSUM_MFMA ? acc_o[DBLK - 1][8] : l_run + __shfl_xor(l_run, 32, 64);
    if (qv && !dead) {
      // (opaque copy of the block's head: keeps the dozen 64-bit store addresses below from being hoisted out of the window
      // loop, where they lived in scratch)
      int h_ep = h;
      asm volatile("" : "+s"(h_ep));
      const float inv = ldexpf(1.0f, -p.kv_e) / l_tot;
      float* dst = p.out ? p.out + (row0 + q) * p.D + (int64_t)h_ep * DH : nullptr;
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
              const int col = h_ep * DH + d0;
              const int64_t eo = ((int64_t)(col >> 5) * p.out_rows + (row0 + q)) * 32 + (col & 31);
              if (p.out_f8) {
                rsp_store_planes4(p.out_hi, p.out_lo, eo, o * p.out_pscale, true);
              } else {          // |o| <= max |v|: inside the fp16 range of the planes; truncating pair (hi + lo ~ 21 bits)
                const f32x4 y = o * p.out_pscale;
                const half2_t h01 = __builtin_bit_cast(half2_t, __builtin_amdgcn_cvt_pkrtz(y[0], y[1]));
                const half2_t h23 = __builtin_bit_cast(half2_t, __builtin_amdgcn_cvt_pkrtz(y[2], y[3]));
                const half2_t l01 = __builtin_bit_cast(half2_t, __builtin_amdgcn_cvt_pkrtz(y[0] - (float)h01[0], y[1] - (float)h01[1]));
                const half2_t l23 = __builtin_bit_cast(half2_t, __builtin_amdgcn_cvt_pkrtz(y[2] - (float)h23[0], y[3] - (float)h23[1]));
                *reinterpret_cast<half4_t*>(p.out_hi + eo) = half4_t{h01[0], h01[1], h23[0], h23[1]};
                *reinterpret_cast<half4_t*>(p.out_lo + eo) = half4_t{l01[0], l01[1], l23[0], l23[1]};
              }
            }
          }
        }
    }
    if (!has_next) break;
    bp = bp_next;
    cur = nxt;
#pragma unroll
    for (int i = 0; i < NDMA; ++i)
      if (drow[i] >= 0) dsrc[i] += win_delta;
  }
}

// fp32 rel-pos tables [2S-1, dh] -> [2 tables][hi, lo][32][dh + 8] fp16 planes scaled by 2^RT (rows >= 2S-1 and the 8 pad
// columns are zero): the A operands of the in-kernel rel-pos products, split once per layer
__global__ void pack_relpos_tables_kernel(const float* __restrict__ rph, const float* __restrict__ rpw,
                                          half_t* __restrict__ out, int nrow, int dh) {
  const int ldt = dh + 8;
  const int n = 2 * 32 * ldt;
  const float ts = ldexpf(1.0f, RT);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int tb = i / (32 * ldt), rem = i - tb * 32 * ldt;
    const int r = rem / ldt, c = rem - r * ldt;
    float v = 0.f;
    if (r < nrow && c < dh) v = (tb ? rpw : rph)[(int64_t)r * dh + c] * ts;
    half_t hi, lo;
    rsp_split1(v, hi, lo);
    out[(tb * 2 + 0) * 32 * ldt + rem] = hi;
    out[(tb * 2 + 1) * 32 * ldt + rem] = lo;
  }
}

template <int DH, bool REL_IN>
int launch_win(const AttnWP& p, int Bp, int grid16, hipStream_t s) {
  // persistent blocks, one per CU (the kernel's LDS and registers allow no second one), a multiple of nh of them so that
  // a block keeps its head; `variant` of the C entry sets the grid in units of 16 blocks for tests (1: every block walks
  // several windows even on a small input) and measurements; 0 = 256 blocks
  const int n_items = Bp * p.nh;
  // 768 blocks = three rounds per CU: the hardware's block dispatch evens out what windows with fewer real queries finish
  // early (per ViT-H layer, batch 8: 0.325 ms with exactly one block per CU, 0.296 with 512 blocks, 0.274 with 768,
  // 0.279 with 1024: profiles/r4_attn_win_micro_pairs.txt)
  int grid = 16 * (grid16 > 0 ? grid16 : 48);
  if (grid > n_items) grid = n_items;
  grid = grid / p.nh * p.nh;
  if (grid < p.nh) grid = p.nh;
  hipLaunchKernelGGL((attn_win_kernel<DH, REL_IN>), dim3((unsigned)grid), dim3(448), 0, s, p, n_items);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

}  // namespace

// shared argument checks + launch of both entry forms (attn_stream.hip forwards its S = 14 calls here)
int rsp_attn_win_dispatch(const float* q, int64_t q_ld, const uint16_t* kv_hi, const uint16_t* kv_lo, int64_t kv_rows,
                          int32_t kv_scale_log2, const float* rel, const uint16_t* rel_tab, float* out, uint16_t* out_hi,
                          uint16_t* out_lo, int32_t out_scale_log2, int32_t Bp, int32_t nh, int32_t dh, float scale,
                          int32_t win_per_side, int32_t win_real_last, int32_t variant, hipStream_t s) {
  if (!q || !kv_hi || !kv_lo || (!rel && !rel_tab) || Bp <= 0 || nh <= 0 || !(scale > 0.f)) return RSP_EINVAL;
  if (!out && !(out_hi && out_lo)) return RSP_EINVAL;
  if ((out_hi == nullptr) != (out_lo == nullptr)) return RSP_EINVAL;
  const int D = nh * dh;
  if ((D & 31) || (q_ld & 3) || kv_rows < (int64_t)Bp * WT) return RSP_EINVAL;
  if (RSP_PLANE_IS_F8(kv_scale_log2)) return RSP_EINVAL;   // K | V are consumed as fp16 hi / lo planes
  // the padded-key mask (NEG_RAW) and the rel-pos bias enter the scores as fp16 values scaled by 2^(EQ + kv_e): with a key
  // plane exponent above 4 the mask would no longer underflow the softmax (-60000 * 2^-(EQ + kv_e) * log2 e > -85) and a
  // bias beyond |t| ~ 64 would saturate fp16; below -8 the biases lose their low bits.  The encoder uses 2.
  if (RSP_PLANE_EXP(kv_scale_log2) > 4 || RSP_PLANE_EXP(kv_scale_log2) < -8) return RSP_EINVAL;
  if (rel_tab && (reinterpret_cast<uintptr_t>(rel_tab) & 15)) return RSP_EINVAL;
  if (variant < 0 || variant > 64) return RSP_EINVAL;
  AttnWP p;
  p.q = q; p.q_ld = q_ld; p.kv_hi = reinterpret_cast<const half_t*>(kv_hi); p.kv_lo = reinterpret_cast<const half_t*>(kv_lo);
  p.kv_rows = kv_rows; p.kv_e = RSP_PLANE_EXP(kv_scale_log2); p.rel = rel; p.rel_tab = reinterpret_cast<const half_t*>(rel_tab);
  p.out = out; p.out_hi = reinterpret_cast<half_t*>(out_hi); p.out_lo = reinterpret_cast<half_t*>(out_lo);
  p.out_pscale = ldexpf(1.0f, RSP_PLANE_EXP(out_scale_log2)); p.out_f8 = out_hi && RSP_PLANE_IS_F8(out_scale_log2);
  p.out_rows = (int64_t)Bp * WT;
  p.nh = nh; p.D = D; p.scale = scale;
  p.win_n = 0; p.win_real = 0;
  if (win_per_side > 0) {
    if (win_real_last < 1 || win_real_last > WS || Bp % (win_per_side * win_per_side)) return RSP_EINVAL;
    p.win_n = win_per_side; p.win_real = win_real_last;
  }
  if (rel) {
    if (dh == 64) return launch_win<64, true>(p, Bp, variant, s);
    if (dh == 80) return launch_win<80, true>(p, Bp, variant, s);
  } else {
    if (dh == 64) return launch_win<64, false>(p, Bp, variant, s);
    if (dh == 80) return launch_win<80, false>(p, Bp, variant, s);
  }
  return RSP_EINVAL;
}

extern "C" int rsp_pack_relpos_tables(const float* rel_pos_h, const float* rel_pos_w, uint16_t* out, int32_t S, int32_t dh,
                                      rsp_stream_t stream) {
  if (!rel_pos_h || !rel_pos_w || !out || S < 1 || 2 * S - 1 > 32 || dh < 8 || (dh & 7)) return RSP_EINVAL;
  hipLaunchKernelGGL(pack_relpos_tables_kernel, dim3(8), dim3(256), 0, (hipStream_t)stream, rel_pos_h, rel_pos_w,
                     reinterpret_cast<half_t*>(out), 2 * S - 1, dh);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

extern "C" int rsp_vit_window_attention(const float* q, int64_t q_ld, const uint16_t* kv_hi, const uint16_t* kv_lo,
                                        int64_t kv_rows, int32_t kv_scale_log2, const uint16_t* rel_tab, float* out,
                                        uint16_t* out_hi, uint16_t* out_lo, int32_t out_scale_log2, int32_t Bp, int32_t nh,
                                        int32_t dh, float scale, int32_t win_per_side, int32_t win_real_last,
                                        int32_t variant, rsp_stream_t stream) {
  if (!rel_tab) return RSP_EINVAL;
  return rsp_attn_win_dispatch(q, q_ld, kv_hi, kv_lo, kv_rows, kv_scale_log2, nullptr, rel_tab, out, out_hi, out_lo,
                               out_scale_log2, Bp, nh, dh, scale, win_per_side, win_real_last, variant, (hipStream_t)stream);
}
