// hipcc-flags: -fslp-vectorize
// SAM two-way transformer, token -> image attention with the K | V projections of the per-RoI keys FOLDED into the kernel
// (HF:326-331 / 397-400: q = tokens + pe_q through q_proj, k = k_proj(keys + pe), v = v_proj(keys), 8 heads x 16).
//
// Layers 1 and "final" attend from the T <= 12 prompt tokens of every RoI to that RoI's OWN 4096 keys ([R * N, 256] fp16
// planes written by the image -> token block).  The unfolded path projects all R * N keys to K | V first: a 429 GFLOP GEMM
// that writes 3.4 GB of fp32 which the attention kernel reads back (2.9 ms per call at R = 800).  The algebra allows the
// opposite order -- with tq[t, h, :] the projected query of token t and head h (16 values),
//   score[(h, t), n] = tq[t, h] . (Wk_h (keys[n] + pe[n]) + bk_h)
//                    = keys[n] . (Wk_h^T tq[t, h])  +  PEK[n, h] . tq[t, h],       PEK = k_proj(pe) + bk  ([N, 128], per model)
//   out[(h, t), :]   = Wv_h (sum_n p[(h, t), n] keys[n]) + bv_h                    (the softmax weights sum to 1)
// -- so the kernel is an attention with 8 T <= 96 query columns q' = Wk_h^T tq[t, h] of width 256 over keys that are BOTH
// its K (plus the small PEK term) and its V: one pass over the key planes, no [R * N, 256] K | V tensor.  The two small
// GEMMs around it (q' from the block-diagonal tq, Wv on the 256-wide result) are host-side calls of rsp_gemm.
//
// Structure = attn_stream.hip (same fp16x3 products, online softmax with the query column per lane, transposing LDS reads
// for the V^T fragments), with what dh = 256 changes:
//   * one RoI per block, 3 waves x 32 query columns.  The B operands live for the whole RoI: the hi halves of q' (16
//     k-steps) and the tq fragments of the <= NPE heads a wave's 32 columns touch in registers, the lo halves of q' in a
//     per-lane LDS slot (48 KB next to the two tile buffers) -- with everything in registers the kernel spilled (the block
//     runs one wave per SIMD: 256 VGPRs + 256 accumulation registers);
//   * ONE key image per tile serves both access patterns: rows of 576 bytes (512 + 64: the four rows of a transposing read
//     start 16 banks apart), 16-byte chunks XOR-swizzled with (row >> 2) & 3 inside their 64-byte group (the ds_read_b128
//     of 16 consecutive rows then hits 16 different 16-byte slots) -- the swizzle stays inside the 64 bytes a transposing
//     read covers, so both are conflict free;
//   * tiles of 32 keys, 54 KB each (keys hi / lo 2 x 18 KB, PEK hi / lo 2 x 9 KB), 2 buffers, filled by `buffer_load ... lds`
//     (18 instructions per lane and tile, all full: the PEK image is padded to an instruction boundary); padding chunks are
//     out-of-range offsets (the buffer returns zeros), the tile advance is the scalar offset.
#include <type_traits>
#include "rsp_common.h"

namespace {

typedef __attribute__((address_space(3))) void* lptr_f;
typedef short v4s_f __attribute__((ext_vector_type(4)));

constexpr int FKT = 32, FNW = 3, FNT = FNW * 64, FNBUF = 2;
constexpr int FD = 256;                                   // key width
constexpr int FKCPR = 36, FPCPR = 17;                     // 16-byte units per image row: keys (32 + 4 pad), PEK (16 + 1 pad)
constexpr int FK_UNITS = FKT * FKCPR;                     // 1152 = 6 x 192
constexpr int FP_UNITS = 576;                             // 32 x 17 = 544, padded to 3 x 192
constexpr int FTILE_UNITS = 2 * FK_UNITS + 2 * FP_UNITS;  // 3456 = 18 x 192
constexpr int FNDMA = FTILE_UNITS / FNT;                  // 18
constexpr int FBUF_BYTES = FTILE_UNITS * 16;              // 55296
constexpr float F_PSCALE_LOG2 = 14.0f;
constexpr unsigned F_OOB = 0x80000000u;                 // a padding chunk: beyond every descriptor, the buffer returns zeros

static_assert(FK_UNITS % FNT == 0 && FP_UNITS % FNT == 0, "every DMA instruction reads one plane");

template <int I, int N, class F>
__device__ __forceinline__ void static_for_f(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for_f<I + 1, N>(f);
  }
}

__device__ __forceinline__ void split8_fast_f(const float* x, half8_t& hi, half8_t& lo) {
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

struct T2iFoldP {
  const half_t* khi; const half_t* klo; int64_t k_rows;   // key planes KB32 [8][k_rows][32]
  const half_t* phi; const half_t* plo;                   // PEK planes KB32 [4][N][32]
  const half_t* qhi; const half_t* qlo; int64_t q_rows;   // q' planes KB32 [8][q_rows][32], rows r * 96 + column
  const half_t* thi; const half_t* tlo;                   // block-diagonal tq planes KB32 [4][q_rows][32]
  float* u;                                               // [R * 96, 256]: sum_n p keys[n] per column
  int N, ncols;                                           // keys per RoI, real columns (8 T)
  float c_main, c_pe;                                     // raw products -> log2-domain scores
  float u_unscale;                                        // 2^-(key plane exponent)
};

// NPE: heads whose PEK term a wave evaluates (its 32 columns h * T + t touch at most 5 heads when T >= 7, else all 8)
// (Round 5, first GPU run of the three other issue schedules written on the emulator -- the next tile's DMA spread between
// the score MFMAs: 11.3 instead of 4.5 ms per step; one score accumulator with deeper LDS lookahead: 4.48 vs 4.51 --
// gpurun_out/r5/job1, profiles/r5_decoder_paths_ab.txt: removed.)
template <int NPE>
__global__ __launch_bounds__(FNT) void sam_t2i_fold_kernel(const T2iFoldP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[FNBUF][FBUF_BYTES];
  __shared__ __attribute__((aligned(1024))) unsigned char sQl[FNW * 16 * 1024];      // q' lo fragments: [wave][k-step][lane] x 16 B
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, l31 = lane & 31;
  const int r = blockIdx.x;
  const int N = p.N;
  const int nt = N / FKT;

  // ---- DMA slots: unit u = i * 192 + tid of the tile image [keys hi | keys lo | PEK hi | PEK lo] ----
  const __amdgpu_buffer_rsrc_t rKh = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.khi), 0, (int)(p.k_rows * 512), 0x00020000);
  const __amdgpu_buffer_rsrc_t rKl = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.klo), 0, (int)(p.k_rows * 512), 0x00020000);
  const __amdgpu_buffer_rsrc_t rPh = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.phi), 0, N * 256, 0x00020000);
  const __amdgpu_buffer_rsrc_t rPl = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.plo), 0, N * 256, 0x00020000);
  // (the hi and lo plane of an image have the same offsets: 6 slots for the keys, 3 for PEK)
  unsigned voff_k[6], voff_p[3];
#pragma unroll
  for (int i = 0; i < 6; ++i) {                           // keys: chunk c of the row = columns 8 c .. 8 c + 7
    const int v = i * FNT + tid;
    const int row = v / FKCPR, pc = v - row * FKCPR;
    const int c = pc ^ ((row >> 2) & 3);
    voff_k[i] = pc < 32 ? (unsigned)(((int64_t)(c >> 2) * p.k_rows + row) * 64 + (c & 3) * 16) : F_OOB;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {                           // PEK: 16 chunks per row
    const int v = i * FNT + tid;
    const int row = v / FPCPR, pc = v - row * FPCPR;
    voff_p[i] = (row < FKT && pc < 16) ? (unsigned)(((int64_t)(pc >> 2) * N + row) * 64 + (pc & 3) * 16) : F_OOB;
  }
  const unsigned so_k0 = (unsigned)((int64_t)r * N * 64);  // first key row of the RoI (bytes inside a 32-column block)
  const rsp_lds_addr_t smem_a = rsp_lds_addr((lptr_f)&smem[0][0]);
  auto issue_slot = [&](auto ic, int kt, int buf) {        // DMA instruction i of tile kt
    constexpr int i = decltype(ic)::value;
    const unsigned so_k = so_k0 + (unsigned)kt * (FKT * 64), so_p = (unsigned)kt * (FKT * 64);
    const rsp_lds_addr_t l = smem_a + buf * FBUF_BYTES + (i * FNT + wave * 64) * 16;
    // (RSP_BUFFER_LOAD_LDS_B128 = the DMA as inline assembly, round 6: through the builtin hipcc knows that the instruction
    // writes LDS and put its own s_waitcnt vmcnt(0) in front of the fragment reads of tile kt -- BEHIND this burst for tile
    // kt + 1: rounds 4-5 ran this kernel without any overlap of the key stream and the matrix work.  The waits that order
    // the ring are the loop's own: s_waitcnt vmcnt(0) + s_barrier at the top of a tile.)
    // (operands through locals: an asm operand inside a generic lambda does not capture by itself)
    const __amdgpu_buffer_rsrc_t rs = i < 6 ? rKh : (i < 12 ? rKl : (i < 15 ? rPh : rPl));
    const unsigned vo = i < 12 ? voff_k[i < 12 ? i % 6 : 0] : voff_p[i >= 12 ? (i - 12) % 3 : 0];
    const unsigned so = i < 12 ? so_k : so_p;
    RSP_BUFFER_LOAD_LDS_B128(rs, l, vo, so);
  };
  static_for_f<0, FNDMA>([&](auto ic) { issue_slot(ic, 0, 0); });

  // ---- B operands of this lane's column for the whole RoI: q' (16 k-steps; lo halves parked in the lane's own LDS
  // slots: written and read by the same lane) and the block-diagonal tq of heads hb .. hb + NPE - 1 ----
  const int64_t qrow = (int64_t)r * 96 + wave * 32 + l31;
  const int hb = NPE == 8 ? 0 : min((wave * 32) / (p.ncols >> 3), 8 - NPE);   // first head of the wave's columns
  half8_t qh[16], th[NPE], tl[NPE];
  unsigned char* const my_ql = sQl + wave * (16 * 1024) + lane * 16;
#pragma unroll
  for (int st = 0; st < 16; ++st) {
    const int64_t o = ((int64_t)(st >> 1) * p.q_rows + qrow) * 32 + 16 * (st & 1) + 8 * hh;
    qh[st] = *reinterpret_cast<const half8_t*>(p.qhi + o);
    *reinterpret_cast<half8_t*>(my_ql + st * 1024) = *reinterpret_cast<const half8_t*>(p.qlo + o);
  }
#pragma unroll
  for (int s_ = 0; s_ < NPE; ++s_) {
    const int st = hb + s_;
    const int64_t o = ((int64_t)(st >> 1) * p.q_rows + qrow) * 32 + 16 * (st & 1) + 8 * hh;
    th[s_] = *reinterpret_cast<const half8_t*>(p.thi + o);
    tl[s_] = *reinterpret_cast<const half8_t*>(p.tlo + o);
  }

  f32x16 acc_o[8];
#pragma unroll
  for (int db = 0; db < 8; ++db)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc_o[db][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // per-lane parts of the image addresses
  const int k_row_off = l31 * (FKCPR * 16);               // A-operand reads: row l31
  const int k_swz = (l31 >> 2) & 3;
  const int p_row_off = l31 * (FPCPR * 16);
  const int li = lane & 15, g16 = (lane >> 4) & 1;
  // transposing reads: row = 16 s + 4 hh + (li >> 2) (+ 8), 16-byte chunk 4 db + 2 g16 + ((li & 3) >> 1), 8 (li & 1) inside
  const int v_row = 4 * hh + (li >> 2);
  const int v_c = 2 * g16 + ((li & 3) >> 1);
  const int v_b = 8 * (li & 1);

  for (int kt = 0; kt < nt; ++kt) {
    const int buf = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's part of tile kt has landed
    __builtin_amdgcn_s_barrier();                         // ... everybody's; the other buffer has been read
    // the 18 DMA instructions of tile kt + 1 (into the buffer everybody has just left), as a burst
    const bool has_next = kt + 1 < nt;
    if (has_next) static_for_f<0, FNDMA>([&](auto ic) { issue_slot(ic, kt + 1, buf ^ 1); });
    const unsigned char* sb = &smem[buf][0];
    const unsigned char* sK0 = sb;
    const unsigned char* sK1 = sb + FK_UNITS * 16;
    const unsigned char* sP0 = sb + 2 * FK_UNITS * 16;
    const unsigned char* sP1 = sP0 + FP_UNITS * 16;

    // ---- S^T = keys q'^T (two accumulators: even / odd k-steps) and the PEK term, fragment reads one step ahead ----
    f32x16 s0, s1, sp;
#pragma unroll
    for (int e = 0; e < 16; ++e) { s0[e] = 0.f; s1[e] = 0.f; sp[e] = 0.f; }
    {
      // fragment reads run LA k-steps ahead of their MFMAs through a ring of LA + 1 register sets (one step = 3 MFMAs)
      constexpr int LA = 1, NB = LA + 1;
      half8_t kfh[NB], kfl[NB];
      auto kread = [&](int st, half8_t& h8, half8_t& l8) {
        const int off = k_row_off + (((2 * st + hh) ^ k_swz) << 4);
        h8 = *reinterpret_cast<const half8_t*>(sK0 + off);
        l8 = *reinterpret_cast<const half8_t*>(sK1 + off);
      };
      auto pread = [&](int s_, half8_t& h8, half8_t& l8) {
        const int off = p_row_off + ((2 * (hb + s_) + hh) << 4);
        h8 = *reinterpret_cast<const half8_t*>(sP0 + off);
        l8 = *reinterpret_cast<const half8_t*>(sP1 + off);
      };
      half8_t qlf[NB];
      auto fetch = [&](auto jc) {                       // the operands of k-step j into ring slot j % NB
        constexpr int j = decltype(jc)::value, sl = j % NB;
        if constexpr (j < 16) {
          kread(j, kfh[sl], kfl[sl]);
          qlf[sl] = *reinterpret_cast<const half8_t*>(my_ql + j * 1024);
        } else if constexpr (j < 16 + NPE) pread(j - 16, kfh[sl], kfl[sl]);
      };
      static_for_f<0, LA>([&](auto jc) { fetch(jc); });
      static_for_f<0, 16 + NPE>([&](auto ic) {
        constexpr int i = decltype(ic)::value, cur = i % NB;
        fetch(std::integral_constant<int, i + LA>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (i < 16) {
          if constexpr ((i & 1) == 0) {
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfl[cur], qh[i], s0, 0, 0, 0);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[cur], qlf[cur], s0, 0, 0, 0);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[cur], qh[i], s0, 0, 0, 0);
          } else {
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfl[cur], qh[i], s1, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[cur], qlf[cur], s1, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[cur], qh[i], s1, 0, 0, 0);
          }
        } else {
          sp = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfl[cur], th[i - 16], sp, 0, 0, 0);
          sp = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[cur], tl[i - 16], sp, 0, 0, 0);
          sp = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[cur], th[i - 16], sp, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    }

    // keys^T fragments for U^T += keys^T P^T come through the transposing read of the SAME image, VLA steps ahead of their
    // MFMAs
    typedef __attribute__((address_space(3))) v4s_f* lv4;
    constexpr int VLA = 1, VNB = VLA + 1;
    v4s_f va[VNB][4];
    auto vread = [&](int s_, int db, v4s_f* f) {
      const int row0 = 16 * s_ + v_row;
      const int c = 4 * db + v_c;
      const int a0 = row0 * (FKCPR * 16) + ((c ^ hh) << 4) + v_b;                     // (row >> 2) & 3 == hh
      const int a1 = (row0 + 8) * (FKCPR * 16) + ((c ^ ((hh + 2) & 3)) << 4) + v_b;   // ... == (hh + 2) & 3
      f[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lv4)(sK0 + a0));
      f[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lv4)(sK0 + a1));
      f[2] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lv4)(sK1 + a0));
      f[3] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lv4)(sK1 + a1));
    };

    // ---- online softmax in the log2 domain: this lane's query column, 16 of the tile's 32 keys per half wave ----
    float sc[16];
    float tmax = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      sc[e] = fmaf(s0[e] + s1[e], p.c_main, sp[e] * p.c_pe);
      tmax = fmaxf(tmax, sc[e]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    const float kk = F_PSCALE_LOG2 - m_new;
    float psum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      sc[e] = __builtin_amdgcn_exp2f(sc[e] + kk);
      psum += sc[e];
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
      for (int db = 0; db < 8; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc_o[db][e] *= alpha;
    }

    // ---- U^T += keys^T P^T ----
    {
      vread(0, 0, va[0]);
      half8_t ph, pl;
      static_for_f<0, 16>([&](auto ic) {
        constexpr int i = decltype(ic)::value, s_ = i / 8, db = i % 8, cur = i % VNB;
        if constexpr (db == 0) split8_fast_f(sc + 8 * s_, ph, pl);
        if constexpr (i + VLA < 16) vread((i + VLA) / 8, (i + VLA) % 8, va[(i + VLA) % VNB]);
        __builtin_amdgcn_sched_barrier(0);
        union { v4s_f s4[2]; half8_t h8; } uh, ul;
        uh.s4[0] = va[cur][0]; uh.s4[1] = va[cur][1]; ul.s4[0] = va[cur][2]; ul.s4[1] = va[cur][3];
        acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ul.h8, ph, acc_o[db], 0, 0, 0);
        acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(uh.h8, pl, acc_o[db], 0, 0, 0);
        acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(uh.h8, ph, acc_o[db], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      });
    }
  }

  // ---- U[column][d] = acc / sum: lane holds d = 32 db + (e & 3) + 8 (e >> 2) + 4 hh of its column ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const int col = wave * 32 + l31;
  if (col < p.ncols) {
    const float inv = p.u_unscale / l_tot;
    float* dst = p.u + qrow * FD;
#pragma unroll
    for (int db = 0; db < 8; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = acc_o[db][4 * g + c] * inv;
        *reinterpret_cast<f32x4*>(dst + db * 32 + 8 * g + 4 * hh) = o;
      }
  }
}

}  // namespace

// keys: planes of [k_rows >= R * N, 256] (KB32, value * 2^keys_e); pek: planes of [N, 128]; qp: planes of [q_rows >= R * 96,
// 256] = q' per (RoI, column); tqx: planes of [q_rows, 128] = the block-diagonal projected queries (softmax scale inside);
// u: fp32 [R * 96, 256], rows of columns >= ncols are left untouched.
extern "C" int rsp_sam_t2i_fold(const uint16_t* keys_hi, const uint16_t* keys_lo, int64_t k_rows, int32_t keys_e,
                                const uint16_t* pek_hi, const uint16_t* pek_lo, int32_t pek_e, const uint16_t* qp_hi,
                                const uint16_t* qp_lo, int32_t qp_e, const uint16_t* tqx_hi, const uint16_t* tqx_lo,
                                int32_t tqx_e, int64_t q_rows, float* u, int32_t R, int32_t N, int32_t ncols,
                                rsp_stream_t stream) {
  if (!keys_hi || !keys_lo || !pek_hi || !pek_lo || !qp_hi || !qp_lo || !tqx_hi || !tqx_lo || !u || R < 0 || N <= 0 ||
      (N % FKT) || ncols <= 0 || ncols > 96 || (ncols & 7) || k_rows < (int64_t)R * N || q_rows < (int64_t)R * 96 ||
      k_rows * 512 > 0x7fffffffLL || (int64_t)N * 256 > 0x7fffffffLL)
    return RSP_EINVAL;
  if (R == 0) return RSP_OK;
  T2iFoldP p;
  p.khi = reinterpret_cast<const half_t*>(keys_hi); p.klo = reinterpret_cast<const half_t*>(keys_lo); p.k_rows = k_rows;
  p.phi = reinterpret_cast<const half_t*>(pek_hi); p.plo = reinterpret_cast<const half_t*>(pek_lo);
  p.qhi = reinterpret_cast<const half_t*>(qp_hi); p.qlo = reinterpret_cast<const half_t*>(qp_lo); p.q_rows = q_rows;
  p.thi = reinterpret_cast<const half_t*>(tqx_hi); p.tlo = reinterpret_cast<const half_t*>(tqx_lo);
  p.u = u; p.N = N; p.ncols = ncols;
  constexpr float LOG2E_F = 1.4426950408889634f;
  p.c_main = ldexpf(1.0f, -(keys_e + qp_e)) * LOG2E_F;
  p.c_pe = ldexpf(1.0f, -(pek_e + tqx_e)) * LOG2E_F;
  p.u_unscale = ldexpf(1.0f, -keys_e);                   // (the 2^14 of the probabilities cancels against their sum)
  // columns are ordered h * T + t: 32 consecutive ones touch at most 5 heads when T >= 7
  const bool five = ncols >= 56;
  const dim3 g(R), b(FNT);
  hipStream_t s = (hipStream_t)stream;
  if (five) hipLaunchKernelGGL((sam_t2i_fold_kernel<5>), g, b, 0, s, p);
  else hipLaunchKernelGGL((sam_t2i_fold_kernel<8>), g, b, 0, s, p);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}
