// SAM ViT attention for gfx950: flash-style (never materialises the TxT score
// or bias matrices), fp32 online softmax, fp16x3 split-precision MFMA for both
// QK^T and PV (see gemm.hip for the numerics), decomposed rel-pos bias added
// in registers.
//
// Reference semantics (HF:803-831, HF:761-801; vit_sam.py:117-157,202-221):
//   attn = softmax_fp32((q*scale) k^T + rel_h[q,kh] + rel_w[q,kw]) ; out = attn v
//   rel_h[q,kh] = q . Rh[qh-kh+S-1], rel_w[q,kw] = q . Rw[qw-kw+S-1]   (UNSCALED q)
//
// Layout trick: we compute the TRANSPOSED score tile S^T = K Q^T with
// v_mfma_f32_32x32x16_f16, so every lane owns ONE query column (q = lane&31)
// and 16 keys per 32-key block.  Row max / row sum are then per-lane scalars
// (one cross-half shuffle per tile), and the 8 consecutive accumulator
// registers [8*(s&1), 8*(s&1)+8) of block (s>>1) are exactly the B operand
// (P^T) of the PV MFMA for k-step s, provided the A operand (V^T) is read with
// the same key permutation  slot t<4 -> key 16s+4hh+t, t>=4 -> key 16s+8+4hh+t-4.
// No LDS round trip and no cross-lane traffic for P.
#include <type_traits>
#include "rsp_common.h"

namespace {

constexpr int KT = 64;        // keys per tile
constexpr int QB = 128;       // queries per block (4 waves x 32)
constexpr int VT_LD = 68;     // halves per V^T row (34 dwords: conflict-free b64 reads)
constexpr int EQ = 6, EK = 6, EV = 6;  // power-of-two operand scales (fp16 range)
constexpr float P_SCALE = 16384.0f;    // probabilities are scaled by 2^14 before the fp16 split

__device__ __forceinline__ void split8(const float* x, half8_t& hi, half8_t& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    half_t h, l;
    rsp_split1(x[i], h, l);
    hi[i] = h; lo[i] = l;
  }
}

struct AttnP {
  const float* q; const float* k; const float* v; const float* rel; float* out;
  const int32_t* kv_batch_map;  // optional: batch b reads K/V of batch kv_batch_map[b]
  const int32_t* q_batch_map;   // optional: batch b reads Q of batch q_batch_map[b]
  half_t* out_hi; half_t* out_lo; float out_pscale;  // optional fp16-plane copy of `out`, KB32 layout
  int64_t out_rows;                                  // rows (= B * Tq) of that [rows, nh*dh] matrix
  const uint8_t* mask;                               // optional [B, Tq, Tk] bytes, non-zero = blocked (all heads)
  int64_t q_bs, q_ts, q_hs;     // element strides: batch, token, head
  int64_t k_bs, k_ts, k_hs;
  int64_t v_bs, v_ts, v_hs;
  int64_t o_bs, o_ts, o_hs;
  int Tq, Tk, S, nh;
  float scale;
};

// REL: decomposed rel-pos bias (ViT; requires Tq == Tk == S*S): 0 = none, 1 = S == 64 (a key tile is one key row:
// rel_h is one scalar per tile, rel_w lives in registers), 2 = S <= 32 (both tables in LDS).
// MASK: honour p.mask.  Both are compile-time so that the inner loops stay branch-free.
template <int DH, int REL, bool MASK>
__global__ __launch_bounds__(256) void attn_kernel(const AttnP p) {
  constexpr bool HAS_REL = REL != 0;
  constexpr int DSTEPS = DH / 16;
  constexpr int DBLK = (DH + 31) / 32;
  constexpr int K_LD = DH + 8;
  constexpr int DCH = DH / 4;                          // float4 chunks per key row
  constexpr int MT_TOTAL = 16 * DCH;                   // 4key x 4d micro tiles per K/V tile
  constexpr int MT_PER_THREAD = (MT_TOTAL + 255) / 256;
  constexpr int REL_MAX_S = (REL == 2) ? 32 : 1;       // LDS rel table only for S < 64

  __shared__ __attribute__((aligned(16))) half_t sK[2][KT * K_LD];
  __shared__ __attribute__((aligned(16))) half_t sVt[2][DBLK * 32 * VT_LD];
  __shared__ float sRel[QB * (2 * REL_MAX_S + 1)];
  __shared__ int sKmap[REL_MAX_S * REL_MAX_S];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, l31 = lane & 31;
  const int bp = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * QB;
  const int ql = wave * 32 + l31;
  const int q = q0 + ql;
  const int T = p.Tq, TK = p.Tk, S = p.S, nh = p.nh;
  const float scale = p.scale;
  constexpr bool aligned = REL == 1;
  const int kvb = p.kv_batch_map ? p.kv_batch_map[bp] : bp;
  const int qbi = p.q_batch_map ? p.q_batch_map[bp] : bp;
  const float* q_b = p.q + (int64_t)qbi * p.q_bs + (int64_t)h * p.q_hs;
  const float* k_b = p.k + (int64_t)kvb * p.k_bs + (int64_t)h * p.k_hs;
  const float* v_b = p.v + (int64_t)kvb * p.v_bs + (int64_t)h * p.v_hs;
  const float* rel_b = HAS_REL ? p.rel + ((int64_t)bp * nh + h) * T * (2 * S) : nullptr;
  float* out = p.out;

  // ---- Q fragments (B operand of S^T = K Q^T), scaled and split once ----
  half8_t qh[DSTEPS], qlo[DSTEPS];
  {
    const float qs = scale * ldexpf(1.0f, EQ);
#pragma unroll
    for (int st = 0; st < DSTEPS; ++st) {
      float x[8];
      if (q < T) {
        const float* src = q_b + (int64_t)q * p.q_ts + st * 16 + hh * 8;
        const f32x4 a = *reinterpret_cast<const f32x4*>(src);
        const f32x4 b = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { x[i] = a[i] * qs; x[4 + i] = b[i] * qs; }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = 0.f;
      }
      split8(x, qh[st], qlo[st]);
    }
  }

  // ---- rel-pos bias sources ----
  float bw[2][16];  // aligned path: rel_w for this lane's 32 key columns (tile invariant)
  float bh_next = 0.f;   // aligned path: rel_h of the NEXT key tile, fetched a whole tile ahead of its use
  if constexpr (aligned) {
    bh_next = (q < T) ? rel_b[(int64_t)q * (2 * S)] : 0.f;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kw = 32 * blk + (r & 3) + 8 * (r >> 2) + 4 * hh;
        bw[blk][r] = (q < T) ? rel_b[(int64_t)q * (2 * S) + S + kw] : 0.f;
      }
  } else if constexpr (REL == 2) {
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int r = 0; r < 16; ++r) bw[blk][r] = 0.f;
    const int W2 = 2 * S;
    for (int idx = tid; idx < QB * W2; idx += 256) {
      const int r = idx / W2, j = idx - r * W2;
      const int qq = q0 + r;
      sRel[r * (W2 + 1) + j] = (qq < T) ? rel_b[(int64_t)qq * W2 + j] : 0.f;
    }
    for (int k = tid; k < T; k += 256) {
      const int kh = k / S;
      sKmap[k] = kh | ((k - kh * S) << 16);
    }
  }

  // ---- K/V staging registers (issue-early / write-late) ----
  f32x4 kreg[MT_PER_THREAD][4], vreg[MT_PER_THREAD][4];
  auto load_kv = [&](int kt) {
#pragma unroll
    for (int i = 0; i < MT_PER_THREAD; ++i) {
      const int mt = tid + 256 * i;
      if (mt < MT_TOTAL) {
        const int kg = mt / DCH, dc = mt - kg * DCH;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int key = kt * KT + kg * 4 + j;
          if (key < TK) {
            kreg[i][j] = *reinterpret_cast<const f32x4*>(k_b + (int64_t)key * p.k_ts + dc * 4);
            vreg[i][j] = *reinterpret_cast<const f32x4*>(v_b + (int64_t)key * p.v_ts + dc * 4);
          } else {
            kreg[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            vreg[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
      }
    }
  };
  auto store_kv = [&]() {
    const float ks = ldexpf(1.0f, EK), vs = ldexpf(1.0f, EV);
#pragma unroll
    for (int i = 0; i < MT_PER_THREAD; ++i) {
      const int mt = tid + 256 * i;
      if (mt < MT_TOTAL) {
        const int kg = mt / DCH, dc = mt - kg * DCH;
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // K: row = key, 4 consecutive d
          half4_t hi, lo;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            half_t a, b;
            rsp_split1(kreg[i][j][c] * ks, a, b);
            hi[c] = a; lo[c] = b;
          }
          const int off = (kg * 4 + j) * K_LD + dc * 4;
          *reinterpret_cast<half4_t*>(&sK[0][off]) = hi;
          *reinterpret_cast<half4_t*>(&sK[1][off]) = lo;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {  // V^T: row = d, 4 consecutive keys
          half4_t hi, lo;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            half_t a, b;
            rsp_split1(vreg[i][j][c] * vs, a, b);
            hi[j] = a; lo[j] = b;
          }
          const int off = (dc * 4 + c) * VT_LD + kg * 4;
          *reinterpret_cast<half4_t*>(&sVt[0][off]) = hi;
          *reinterpret_cast<half4_t*>(&sVt[1][off]) = lo;
        }
      }
    }
  };

  // zero the padded V^T rows (d >= DH) once
  if (DBLK * 32 > DH) {
    for (int idx = tid; idx < (DBLK * 32 - DH) * VT_LD; idx += 256) {
      sVt[0][DH * VT_LD + idx] = (half_t)0.f;
      sVt[1][DH * VT_LD + idx] = (half_t)0.f;
    }
  }

  f32x16 acc_o[DBLK];
#pragma unroll
  for (int db = 0; db < DBLK; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[db][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float s_unscale = ldexpf(1.0f, -(EQ + EK));
  const float LOG2E = 1.4426950408889634f;

  const int nt = (TK + KT - 1) / KT;
  load_kv(0);
  for (int kt = 0; kt < nt; ++kt) {
    store_kv();
    __syncthreads();
    if (kt + 1 < nt) load_kv(kt + 1);

    // ---- S^T = K Q^T ----
    f32x16 sc[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[blk][r] = 0.f;
#pragma unroll
      for (int st = 0; st < DSTEPS; ++st) {
        const int off = (blk * 32 + l31) * K_LD + st * 16 + hh * 8;
        const half8_t kh8 = *reinterpret_cast<const half8_t*>(&sK[0][off]);
        const half8_t kl8 = *reinterpret_cast<const half8_t*>(&sK[1][off]);
        sc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl8, qh[st], sc[blk], 0, 0, 0);
        sc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh8, qlo[st], sc[blk], 0, 0, 0);
        sc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh8, qh[st], sc[blk], 0, 0, 0);
      }
    }

    // ---- bias, mask, online softmax (per-lane query column) ----
    float bh_t = 0.f;
    if constexpr (aligned) {
      bh_t = bh_next;
      if (kt + 1 < nt && q < T) bh_next = rel_b[(int64_t)q * (2 * S) + kt + 1];
    }
    const int tk_lim = TK - kt * KT;      // keys of this tile that exist (>= KT except on the last tile)
    float tmax = -INFINITY;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kl = blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        float v = sc[blk][r] * s_unscale;
        if constexpr (aligned) {
          v += bh_t + bw[blk][r];
        } else if constexpr (REL == 2) {
          if (kl < tk_lim) {
            const int km = sKmap[kt * KT + kl];
            v += sRel[ql * (2 * S + 1) + (km & 0xffff)] + sRel[ql * (2 * S + 1) + S + (km >> 16)];
          }
        }
        if constexpr (MASK) {
          if (kl < tk_lim && q < T && p.mask[((int64_t)bp * T + q) * TK + kt * KT + kl]) v = -INFINITY;
        }
        sc[blk][r] = v;
      }
    if (tk_lim < KT) {   // wave-uniform: only the ragged last tile pays for the bounds test
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh >= tk_lim) sc[blk][r] = -INFINITY;
    }
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sc[blk][r]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    // fully masked so far (m_new == -inf): contribute nothing and keep the state (no inf - inf)
    const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((m_run - m_safe) * LOG2E);
    const float m_l2 = m_safe * LOG2E;
    float psum = 0.f;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        // v_exp_f32 directly: arguments are <= 0, results in [0, 1]; flushing results below 2^-126 is harmless
        const float pv = __builtin_amdgcn_exp2f(sc[blk][r] * LOG2E - m_l2) * P_SCALE;
        sc[blk][r] = pv;
        psum += pv;
      }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int db = 0; db < DBLK; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc_o[db][r] *= alpha;

    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float pf[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) pf[t] = sc[s >> 1][8 * (s & 1) + t];
      half8_t ph, pl;
      split8(pf, ph, pl);
#pragma unroll
      for (int db = 0; db < DBLK; ++db) {
        const int off = (db * 32 + l31) * VT_LD + 16 * s + 4 * hh;
        half8_t vh8, vl8;
        const half4_t a0 = *reinterpret_cast<const half4_t*>(&sVt[0][off]);
        const half4_t a1 = *reinterpret_cast<const half4_t*>(&sVt[0][off + 8]);
        const half4_t b0 = *reinterpret_cast<const half4_t*>(&sVt[1][off]);
        const half4_t b1 = *reinterpret_cast<const half4_t*>(&sVt[1][off + 8]);
#pragma unroll
        for (int t = 0; t < 4; ++t) { vh8[t] = a0[t]; vh8[4 + t] = a1[t]; vl8[t] = b0[t]; vl8[4 + t] = b1[t]; }
        acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl8, ph, acc_o[db], 0, 0, 0);
        acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh8, pl, acc_o[db], 0, 0, 0);
        acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh8, ph, acc_o[db], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- normalise and store: lane holds O[q][d], d = db*32 + (r&3) + 8*(r>>2) + 4*hh ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (q < T) {
    const float inv = ldexpf(1.0f, -EV) / l_tot;
    float* dst = out + (int64_t)bp * p.o_bs + (int64_t)q * p.o_ts + (int64_t)h * p.o_hs;
#pragma unroll
    for (int db = 0; db < DBLK; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = db * 32 + 8 * g + 4 * hh;
        if (d0 < DH) {
          f32x4 o;
#pragma unroll
          for (int c = 0; c < 4; ++c) o[c] = acc_o[db][4 * g + c] * inv;
          if (out) *reinterpret_cast<f32x4*>(dst + d0) = o;
          if (p.out_hi) {
            half4_t h4, l4;
#pragma unroll
            for (int c = 0; c < 4; ++c) { half_t a, b; rsp_split1(o[c] * p.out_pscale, a, b); h4[c] = a; l4[c] = b; }
            const int col = h * DH + d0;
            const int64_t eo = ((int64_t)(col >> 5) * p.out_rows + ((int64_t)bp * T + q)) * 32 + (col & 31);
            *reinterpret_cast<half4_t*>(p.out_hi + eo) = h4;
            *reinterpret_cast<half4_t*>(p.out_lo + eo) = l4;
          }
        }
      }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Windowed ViT layers (HF:900-972: 14x14 windows, T = 196 tokens): the whole K and V of one (window, head) fit in
// LDS as fp16 hi/lo, so a block stages them ONCE and its 7 waves (32 queries each) run all 7 key blocks without any
// further barrier.  The decomposed rel-pos bias of a query is 14 + 14 scalars kept in registers; key -> (kh, kw)
// is compile-time after unrolling.  Same arithmetic as attn_kernel (fp16x3 MFMA, fp32 online softmax).
template <int I, int N, class F>
__device__ __forceinline__ void static_for_w(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for_w<I + 1, N>(f);
  }
}

template <int DH>
__global__ __launch_bounds__(448) void attn_window_kernel(const AttnP p) {
  constexpr int S = 14, T = S * S;             // 196 tokens
  constexpr int NKB = (T + 31) / 32;           // 7 key blocks of 32
  constexpr int KP = NKB * 32;                 // 224 padded keys
  constexpr int NT = NKB * 64;                 // 7 waves
  constexpr int DSTEPS = DH / 16;
  constexpr int DBLK = (DH + 31) / 32;
  constexpr int K_LD = DH + 8;                 // halves per K row
  constexpr int V_LD = KP + 8;                 // halves per V^T row: 464 B = 29 x 16 (16-byte aligned, conflict-free b128 reads)
  constexpr int DCH = DH / 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
  half_t* sK0 = reinterpret_cast<half_t*>(wsm);            // [KP][K_LD] hi
  half_t* sK1 = sK0 + KP * K_LD;                            // lo
  half_t* sV0 = sK1 + KP * K_LD;                            // [DH][V_LD] hi (V transposed)
  half_t* sV1 = sV0 + DH * V_LD;                            // lo

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, l31 = lane & 31;
  const int bp = blockIdx.z, h = blockIdx.y;
  const int q = wave * 32 + l31;
  const int nh = p.nh;
  const float* q_b = p.q + (int64_t)bp * p.q_bs + (int64_t)h * p.q_hs;
  const float* k_b = p.k + (int64_t)bp * p.k_bs + (int64_t)h * p.k_hs;
  const float* v_b = p.v + (int64_t)bp * p.v_bs + (int64_t)h * p.v_hs;
  const float* rel_b = p.rel + ((int64_t)bp * nh + h) * T * (2 * S);

  // ---- stage K (row = key) and V^T (row = d) of the whole window, split to fp16 hi/lo; padded keys are zero ----
  {
    const float ks = ldexpf(1.0f, EK), vs = ldexpf(1.0f, EV);
    for (int mt = tid; mt < (KP / 4) * DCH; mt += NT) {
      const int kg = mt / DCH, dc = mt - kg * DCH;
      f32x4 kr[4], vr[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int key = kg * 4 + j;
        if (key < T) {
          kr[j] = *reinterpret_cast<const f32x4*>(k_b + (int64_t)key * p.k_ts + dc * 4);
          vr[j] = *reinterpret_cast<const f32x4*>(v_b + (int64_t)key * p.v_ts + dc * 4);
        } else {
          kr[j] = f32x4{0.f, 0.f, 0.f, 0.f};
          vr[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        half4_t hi, lo;
#pragma unroll
        for (int c = 0; c < 4; ++c) { half_t a, b; rsp_split1(kr[j][c] * ks, a, b); hi[c] = a; lo[c] = b; }
        const int off = (kg * 4 + j) * K_LD + dc * 4;
        *reinterpret_cast<half4_t*>(sK0 + off) = hi;
        *reinterpret_cast<half4_t*>(sK1 + off) = lo;
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        half4_t hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) { half_t a, b; rsp_split1(vr[j][c] * vs, a, b); hi[j] = a; lo[j] = b; }
        // key order inside every 16-key group: [0-3, 8-11, 4-7, 12-15] (4-key groups 1 and 2 swapped), so that the 8
        // keys one half wave multiplies per MFMA are one 16-byte chunk (see attn_global.hip)
        const int kgp = (kg & ~3) | ((kg & 1) << 1) | ((kg & 2) >> 1);
        const int off = (dc * 4 + c) * V_LD + kgp * 4;
        *reinterpret_cast<half4_t*>(sV0 + off) = hi;
        *reinterpret_cast<half4_t*>(sV1 + off) = lo;
      }
    }
  }

  // ---- Q fragments and this query's 14 + 14 bias scalars ----
  half8_t qh[DSTEPS], qlo[DSTEPS];
  {
    const float qs = p.scale * ldexpf(1.0f, EQ);
#pragma unroll
    for (int st = 0; st < DSTEPS; ++st) {
      float x[8];
      if (q < T) {
        const float* src = q_b + (int64_t)q * p.q_ts + st * 16 + hh * 8;
        const f32x4 a = *reinterpret_cast<const f32x4*>(src);
        const f32x4 b = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { x[i] = a[i] * qs; x[4 + i] = b[i] * qs; }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = 0.f;
      }
      split8(x, qh[st], qlo[st]);
    }
  }
  float bH[S], bW[S];
  {
    const float* rq = rel_b + (int64_t)(q < T ? q : 0) * (2 * S);
#pragma unroll
    for (int j = 0; j < S; j += 2) {   // rows are 28 floats = 112 B: 8-byte aligned
      const float2 a = *reinterpret_cast<const float2*>(rq + j);
      const float2 b = *reinterpret_cast<const float2*>(rq + S + j);
      bH[j] = a.x; bH[j + 1] = a.y; bW[j] = b.x; bW[j + 1] = b.y;
    }
  }
  __syncthreads();                      // the only barrier: K / V^T are complete

  f32x16 acc_o[DBLK];
#pragma unroll
  for (int db = 0; db < DBLK; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[db][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float s_unscale = ldexpf(1.0f, -(EQ + EK));
  const float LOG2E = 1.4426950408889634f;

  static_for_w<0, (NKB + 1) / 2>([&](auto tc) {
    constexpr int tile = decltype(tc)::value;               // 64 keys: blocks 2*tile, 2*tile + 1
    constexpr int NB = (2 * tile + 1 < NKB) ? 2 : 1;
    f32x16 sc[NB];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[blk][r] = 0.f;
#pragma unroll
      for (int st = 0; st < DSTEPS; ++st) {
        const int off = ((2 * tile + blk) * 32 + l31) * K_LD + st * 16 + hh * 8;
        const half8_t kh8 = *reinterpret_cast<const half8_t*>(sK0 + off);
        const half8_t kl8 = *reinterpret_cast<const half8_t*>(sK1 + off);
        sc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl8, qh[st], sc[blk], 0, 0, 0);
        sc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh8, qlo[st], sc[blk], 0, 0, 0);
        sc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh8, qh[st], sc[blk], 0, 0, 0);
      }
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        // key index for the two half waves; (kh, kw) and validity are compile-time constants
        const int key0 = (2 * tile + blk) * 32 + (r & 3) + 8 * (r >> 2), key1 = key0 + 4;
        const float b0 = key0 < T ? bH[key0 < T ? key0 / S : 0] + bW[key0 < T ? key0 % S : 0] : -INFINITY;
        const float b1 = key1 < T ? bH[key1 < T ? key1 / S : 0] + bW[key1 < T ? key1 % S : 0] : -INFINITY;
        const float v = sc[blk][r] * s_unscale + (hh ? b1 : b0);
        sc[blk][r] = v;
        tmax = fmaxf(tmax, v);
      }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);                 // finite: every tile has valid keys for hh == 0
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
    const float m_l2 = m_new * LOG2E;
    float psum = 0.f;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(sc[blk][r] * LOG2E - m_l2) * P_SCALE;
        sc[blk][r] = pv;
        psum += pv;
      }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int db = 0; db < DBLK; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc_o[db][r] *= alpha;
#pragma unroll
    for (int s = 0; s < 2 * NB; ++s) {
      float pf[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) pf[t] = sc[s >> 1][8 * (s & 1) + t];
      half8_t ph, pl;
      split8(pf, ph, pl);
#pragma unroll
      for (int db = 0; db < DBLK; ++db) {
        int row = db * 32 + l31;
        if (DBLK * 32 > DH && row >= DH) row = DH - 1;       // rows >= DH feed output rows nobody stores
        const int off = row * V_LD + 64 * tile + 16 * s + 8 * hh;
        const half8_t vh8 = *reinterpret_cast<const half8_t*>(sV0 + off);
        const half8_t vl8 = *reinterpret_cast<const half8_t*>(sV1 + off);
        acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl8, ph, acc_o[db], 0, 0, 0);
        acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh8, pl, acc_o[db], 0, 0, 0);
        acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh8, ph, acc_o[db], 0, 0, 0);
      }
    }
  });

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (q < T) {
    const float inv = ldexpf(1.0f, -EV) / l_tot;
    float* dst = p.out + (int64_t)bp * p.o_bs + (int64_t)q * p.o_ts + (int64_t)h * p.o_hs;
#pragma unroll
    for (int db = 0; db < DBLK; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = db * 32 + 8 * g + 4 * hh;
        if (d0 < DH) {
          f32x4 o;
#pragma unroll
          for (int c = 0; c < 4; ++c) o[c] = acc_o[db][4 * g + c] * inv;
          if (p.out) *reinterpret_cast<f32x4*>(dst + d0) = o;
          if (p.out_hi) {
            half4_t h4, l4;
#pragma unroll
            for (int c = 0; c < 4; ++c) { half_t a, b; rsp_split1(o[c] * p.out_pscale, a, b); h4[c] = a; l4[c] = b; }
            const int col = h * DH + d0;
            const int64_t eo = ((int64_t)(col >> 5) * p.out_rows + ((int64_t)bp * T + q)) * 32 + (col & 31);
            *reinterpret_cast<half4_t*>(p.out_hi + eo) = h4;
            *reinterpret_cast<half4_t*>(p.out_lo + eo) = l4;
          }
        }
      }
  }
}

template <int DH>
static int launch_attn_window(const AttnP& p, int B, hipStream_t s) {
  constexpr int KP = 224;
  const size_t smem = (size_t)2 * (KP * (DH + 8) + DH * (KP + 8)) * sizeof(half_t);
  // (a per-device attribute: set on every call, no cached "done" flag that would be wrong for a second device)
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_window_kernel<DH>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
    return RSP_ELAUNCH;
  hipLaunchKernelGGL((attn_window_kernel<DH>), dim3(1, p.nh, B), dim3(448), smem, s, p);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

}  // namespace

template <int DH, int REL, bool MASK>
static int launch_attn(const AttnP& p, int B, hipStream_t s) {
  dim3 grid((p.Tq + QB - 1) / QB, p.nh, B);
  hipLaunchKernelGGL((attn_kernel<DH, REL, MASK>), grid, dim3(256), 0, s, p);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

extern "C" int rsp_vit_attention_ex(const float* qkv, const float* rel, float* out, uint16_t* out_hi,
                                    uint16_t* out_lo, int32_t out_scale_log2, int32_t Bp, int32_t S,
                                    int32_t nh, int32_t dh, float scale, rsp_stream_t stream);

extern "C" int rsp_vit_attention(const float* qkv, const float* rel, float* out, int32_t Bp,
                                 int32_t S, int32_t nh, int32_t dh, float scale,
                                 rsp_stream_t stream) {
  if (!out) return RSP_EINVAL;
  return rsp_vit_attention_ex(qkv, rel, out, nullptr, nullptr, 0, Bp, S, nh, dh, scale, stream);
}

extern "C" int rsp_vit_attention_ex(const float* qkv, const float* rel, float* out, uint16_t* out_hi,
                                    uint16_t* out_lo, int32_t out_scale_log2, int32_t Bp, int32_t S,
                                    int32_t nh, int32_t dh, float scale, rsp_stream_t stream) {
  if (!qkv || !rel || Bp <= 0 || S <= 0 || nh <= 0) return RSP_EINVAL;
  if (!out && !(out_hi && out_lo)) return RSP_EINVAL;
  if (!(S == 64 || S <= 32)) return RSP_EINVAL;
  const int T = S * S;
  const int64_t D = (int64_t)nh * dh;
  AttnP p;
  p.q = qkv; p.k = qkv + D; p.v = qkv + 2 * D; p.rel = rel; p.out = out; p.kv_batch_map = nullptr; p.q_batch_map = nullptr;
  p.out_hi = reinterpret_cast<half_t*>(out_hi); p.out_lo = reinterpret_cast<half_t*>(out_lo);
  p.out_pscale = ldexpf(1.0f, out_scale_log2);
  p.mask = nullptr;
  p.out_rows = (int64_t)Bp * T;
  if (out_hi && ((nh * dh) & 31)) return RSP_EINVAL;
  p.q_bs = p.k_bs = p.v_bs = (int64_t)T * 3 * D;
  p.q_ts = p.k_ts = p.v_ts = 3 * D;
  p.q_hs = p.k_hs = p.v_hs = dh;
  p.o_bs = (int64_t)T * D; p.o_ts = D; p.o_hs = dh;
  p.Tq = T; p.Tk = T; p.S = S; p.nh = nh; p.scale = scale;
  hipStream_t s = (hipStream_t)stream;
  if (S == 14) {   // the SAM window size: whole-window-resident kernel
    if (dh == 64) return launch_attn_window<64>(p, Bp, s);
    if (dh == 80) return launch_attn_window<80>(p, Bp, s);
  }
  if (dh == 64) return S == 64 ? launch_attn<64, 1, false>(p, Bp, s) : launch_attn<64, 2, false>(p, Bp, s);
  if (dh == 80) return S == 64 ? launch_attn<80, 1, false>(p, Bp, s) : launch_attn<80, 2, false>(p, Bp, s);
  return RSP_EINVAL;
}

extern "C" int rsp_attention(const RspAttnDesc* d, rsp_stream_t stream) {
  if (!d || !d->q || !d->k || !d->v || (!d->out && !d->out_hi)) return RSP_EINVAL;
  if (d->B <= 0 || d->nh <= 0 || d->Tq <= 0 || d->Tk <= 0) return RSP_EINVAL;
  if ((d->q_ts & 3) || (d->k_ts & 3) || (d->v_ts & 3) || (d->o_ts & 3) || (d->q_hs & 3) ||
      (d->k_hs & 3) || (d->v_hs & 3) || (d->o_hs & 3))
    return RSP_EINVAL;  // float4 loads/stores
  AttnP p;
  p.q = d->q; p.k = d->k; p.v = d->v; p.rel = nullptr; p.out = d->out;
  p.kv_batch_map = d->kv_batch_map; p.q_batch_map = d->q_batch_map;
  p.out_hi = reinterpret_cast<half_t*>(d->out_hi); p.out_lo = reinterpret_cast<half_t*>(d->out_lo);
  p.out_pscale = ldexpf(1.0f, d->out_scale_log2); p.out_rows = (int64_t)d->B * d->Tq;
  p.mask = d->mask;
  if (p.out_hi) {   // plane copy assumes a dense [B*Tq, nh*dh] output matrix
    if (!p.out_lo || d->o_hs != d->dh || d->o_ts != (int64_t)d->nh * d->dh || d->o_bs != d->o_ts * d->Tq ||
        ((d->nh * d->dh) & 31))
      return RSP_EINVAL;
  }
  p.q_bs = d->q_bs; p.q_ts = d->q_ts; p.q_hs = d->q_hs;
  p.k_bs = d->k_bs; p.k_ts = d->k_ts; p.k_hs = d->k_hs;
  p.v_bs = d->v_bs; p.v_ts = d->v_ts; p.v_hs = d->v_hs;
  p.o_bs = d->o_bs; p.o_ts = d->o_ts; p.o_hs = d->o_hs;
  p.Tq = d->Tq; p.Tk = d->Tk; p.S = 0; p.nh = d->nh; p.scale = d->scale;
  hipStream_t s = (hipStream_t)stream;
  switch (d->dh) {
    case 16: return p.mask ? launch_attn<16, 0, true>(p, d->B, s) : launch_attn<16, 0, false>(p, d->B, s);
    case 32: return p.mask ? launch_attn<32, 0, true>(p, d->B, s) : launch_attn<32, 0, false>(p, d->B, s);
    case 64: return p.mask ? launch_attn<64, 0, true>(p, d->B, s) : launch_attn<64, 0, false>(p, d->B, s);
    default: return RSP_EINVAL;
  }
}
