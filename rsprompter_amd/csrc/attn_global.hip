// Global-attention layers of the SAM ViT (HF:803-831, S = 64 / 32: T = 4096 / 1024 tokens), DMA-fed variant of
// attn_kernel<DH, 1>:
//   1. vit_kv_split_kernel converts K and V of every (image, head) ONCE into fp16 hi/lo planes -- K as [key][dh],
//      V transposed as [d][key] -- instead of every 128-query block re-splitting all T keys (32x per element at
//      T = 4096; that conversion was a third of the attention kernel's VALU instructions, and the kernel is
//      VALU-issue bound: PMC, profiles/r1_pmc/attn_global_*.txt).
//   2. attn_global_kernel streams 64-key K / V^T tiles HBM -> LDS with global_load_lds_dwordx4 into a 2-deep ring
//      (one barrier per tile, no staging registers, no conversion), XOR-swizzled 128-byte rows (K rows of dh = 80
//      are pitched to 176 bytes through dummy chunks instead) so that the ds_read_b128 / b64 fragment reads are
//      conflict-free.  Arithmetic: S^T = K Q^T and O^T += V^T P^T as fp16x3 MFMA, fp32 online softmax, decomposed
//      rel-pos bias from registers (rel_w) and one scalar per key tile (rel_h).  (A 2-pass PV with P or V as ONE fp16
//      was measured and rejected: its error is 2^-12 max|V| per layer whenever the softmax is sharp -- 1.2e-3 on the
//      ViT-H + LoRA fixture against the 1e-3 budget)
//   3. The keys of every 16-key group of V^T are stored in the order [0-3, 8-11, 4-7, 12-15]: the 8 keys a half wave
//      multiplies in one MFMA (those of its P registers) are then ONE 16-byte chunk -> a single conflict-free
//      ds_read_b128 per plane instead of two 2-way conflicting ds_read_b64 (PMC r2 base: 37 % LDS conflict cycles).
//   4. dh = 80 (ViT-H): the 90 KB ring allows one block per CU, so the block is 8 waves (256 queries, 2 waves / SIMD)
//      instead of 4; blocks are numbered so that all query blocks of an (image, head) run on the SAME XCD and share
//      its L2 (K / V of one head = 2.6 MB; round-robin placement re-fetched them 9x from the memory side).
#include "rsp_common.h"

namespace {

constexpr int KT = 64;                 // keys per tile
constexpr int EQ = 6, EK = 6, EV = 6;  // power-of-two operand scales (same as attn.hip)
constexpr float P_SCALE = 16384.0f;
constexpr float LOG2E_C = 1.4426950408889634f;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ uint4 g_zero16[4];          // source of the dummy (padding) DMA chunks

__device__ __forceinline__ void split8(const float* x, half8_t& hi, half8_t& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    half_t h, l;
    rsp_split1(x[i], h, l);
    hi[i] = h; lo[i] = l;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// qkv [Bp*T, 3*nh*DH] fp32  ->  Kh/Kl [Bp*nh][T][DH], Vth/Vtl [Bp*nh][DH][T]   (fp16, value * 2^EK / 2^EV)
// block = (64-key tile, head, image): K rows are written as they come, V goes through an LDS transpose.
template <int DH>
__global__ __launch_bounds__(256) void vit_kv_split_kernel(const float* __restrict__ qkv, half_t* __restrict__ Kh,
                                                           half_t* __restrict__ Kl, half_t* __restrict__ Vth,
                                                           half_t* __restrict__ Vtl, int T, int nh) {
  constexpr int DCH = DH / 4;
  __shared__ half_t sV[2][DH][KT + 8];
  const int tid = threadIdx.x;
  const int k0 = blockIdx.x * KT, h = blockIdx.y, bp = blockIdx.z;
  const int64_t D3 = (int64_t)3 * nh * DH;
  const float* kb = qkv + ((int64_t)bp * T + k0) * D3 + (int64_t)(nh + h) * DH;
  const float* vb = kb + (int64_t)nh * DH;
  const int64_t bh = (int64_t)bp * nh + h;
  const float ks = ldexpf(1.0f, EK), vs = ldexpf(1.0f, EV);
  for (int u = tid; u < KT * DCH; u += 256) {
    const int key = u / DCH, dc = u - key * DCH;
    const f32x4 kv = *reinterpret_cast<const f32x4*>(kb + (int64_t)key * D3 + dc * 4);
    const f32x4 vv = *reinterpret_cast<const f32x4*>(vb + (int64_t)key * D3 + dc * 4);
    half4_t hi, lo;
#pragma unroll
    for (int c = 0; c < 4; ++c) { half_t a, b; rsp_split1(kv[c] * ks, a, b); hi[c] = a; lo[c] = b; }
    const int64_t ko = (bh * T + k0 + key) * DH + dc * 4;
    *reinterpret_cast<half4_t*>(Kh + ko) = hi;
    *reinterpret_cast<half4_t*>(Kl + ko) = lo;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      half_t a, b;
      rsp_split1(vv[c] * vs, a, b);
      const int kp = (key & ~12) | ((key & 4) << 1) | ((key & 8) >> 1);     // swap key bits 2 and 3 (see header, 3.)
      sV[0][dc * 4 + c][kp] = a;
      sV[1][dc * 4 + c][kp] = b;
    }
  }
  __syncthreads();
  for (int u = tid; u < DH * (KT / 8); u += 256) {       // 16-byte pieces of the transposed rows
    const int d = u / (KT / 8), c8 = u - d * (KT / 8);
    const int64_t vo = (bh * DH + d) * T + k0 + c8 * 8;
    *reinterpret_cast<half8_t*>(Vth + vo) = *reinterpret_cast<const half8_t*>(&sV[0][d][c8 * 8]);
    *reinterpret_cast<half8_t*>(Vtl + vo) = *reinterpret_cast<const half8_t*>(&sV[1][d][c8 * 8]);
  }
}

// ---------------------------------------------------------------------------------------------------------------
struct AttnGP {
  const float* q; const float* rel; float* out;
  const half_t* Kh; const half_t* Kl; const half_t* Vth; const half_t* Vtl;
  half_t* out_hi; half_t* out_lo; float out_pscale;
  int64_t out_rows;
  int64_t q_bs, q_ts, q_hs, o_bs, o_ts, o_hs;
  int T, S, nh;
  float scale;
};

// XCD-aware numbering (block b runs on XCD b % 8): every XCD gets a contiguous range of the logical order
__device__ __forceinline__ unsigned xcd_contig(unsigned bid, unsigned nblk) {
  const unsigned q = nblk >> 3, r = nblk & 7u;
  const unsigned xcd = bid & 7u, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int DH, int NW>
__global__ __launch_bounds__(NW * 64) void attn_global_kernel(const AttnGP p) {
  constexpr int NT = NW * 64;
  constexpr int QB = NW * 32;                         // queries per block
  constexpr int DSTEPS = DH / 16;
  constexpr int DBLK = (DH + 31) / 32;
  constexpr int KCH = DH / 8;                         // real 16-byte chunks per K row
  constexpr int KCPR = (KCH == 8) ? 8 : KCH + 1;      // chunks per LDS row (dh = 80: 11 -> 176-byte pitch)
  constexpr bool KSWZ = (KCH == 8);                   // 128-byte rows: XOR swizzle instead of a pitch
  constexpr int K_UNITS = KT * KCPR;                  // 16-byte units per K plane tile
  constexpr int V_UNITS = DH * 8;                     // V^T tile: DH rows x 128 bytes
  constexpr int TILE_UNITS = 2 * K_UNITS + 2 * V_UNITS;
  constexpr int NDMA = (TILE_UNITS + NT - 1) / NT;    // DMA instructions per thread per tile
  constexpr int BUF_BYTES = NDMA * NT * 16;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2][BUF_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, l31 = lane & 31;
  const int T = p.T, S = p.S, nh = p.nh;
  const int nqb = T / QB;
  const unsigned lb = xcd_contig(blockIdx.x, gridDim.x);      // logical order: query block fastest, then head, image
  const int bp = (int)(lb / (unsigned)(nqb * nh));
  const int h = (int)(lb / (unsigned)nqb) - bp * nh;
  const int q0 = (int)(lb % (unsigned)nqb) * QB;
  const int q = q0 + wave * 32 + l31;
  const int64_t bh = (int64_t)bp * nh + h;
  const float* q_b = p.q + (int64_t)bp * p.q_bs + (int64_t)h * p.q_hs;
  const float* rel_b = p.rel + bh * T * (2 * S);

  // ---- per-thread DMA slots: unit u = i*256 + tid of the tile image [K_hi | K_lo | V^T_hi | V^T_lo] ----
  const unsigned char* dsrc[NDMA];    // source of tile 0 (bytes); dummy units point at the zero page
  int dstep[NDMA];                    // byte advance per key tile
#pragma unroll
  for (int i = 0; i < NDMA; ++i) {
    const int u = i * NT + tid;
    dsrc[i] = reinterpret_cast<const unsigned char*>(g_zero16);
    dstep[i] = 0;
    if (u < 2 * K_UNITS) {
      const int pl = u / K_UNITS, v = u - pl * K_UNITS;
      const int row = v / KCPR, pc = v - row * KCPR;
      const int c = KSWZ ? (pc ^ ((row >> 1) & 7)) : pc;        // logical chunk held at physical position pc
      if (c < KCH) {
        const half_t* base = pl == 0 ? p.Kh : p.Kl;
        dsrc[i] = reinterpret_cast<const unsigned char*>(base + (bh * T + row) * DH) + c * 16;
        dstep[i] = KT * DH * 2;
      }
    } else if (u < TILE_UNITS) {
      const int w = u - 2 * K_UNITS;
      const int pl = w / V_UNITS, v = w - pl * V_UNITS;
      const int row = v >> 3, pc = v & 7;
      const int c = pc ^ ((row >> 1) & 7);
      const half_t* base = pl == 0 ? p.Vth : p.Vtl;
      dsrc[i] = reinterpret_cast<const unsigned char*>(base + (bh * DH + row) * T) + c * 16;
      dstep[i] = KT * 2;
    }
  }
  auto issue_tile = [&](int kt, int buf) {
    unsigned char* lbase = &smem[buf][0];
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      const unsigned char* src = dsrc[i] + (int64_t)kt * dstep[i];
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lbase + (i * NT + wave * 64) * 16), 16, 0, 0);
    }
  };

  const int nt = T / KT;
  issue_tile(0, 0);

  // ---- Q fragments (B operand of S^T = K Q^T), scaled and split once ----
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
  // rel-pos: a key tile is (part of) one key row kh; rel_w for this lane's key columns lives in registers
  const int tiles_per_row = S / KT > 0 ? S / KT : 1;     // S = 64: 1
  const int rows_per_tile = KT / S > 0 ? KT / S : 1;     // S = 32: 2 key rows per tile
  float bw[2][16];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kl = 32 * blk + (r & 3) + 8 * (r >> 2) + 4 * hh;
      bw[blk][r] = (q < T) ? rel_b[(int64_t)q * (2 * S) + S + (kl % S)] * LOG2E_C : 0.f;   // log2 domain
    }
  (void)tiles_per_row;
  // rel_h of the key row(s) of a tile: blk b of tile kt covers key row (kt*64 + 32 b) / S
  auto load_bh = [&](int kt, float& b0, float& b1) {
    const int kh0 = (kt * KT) / S, kh1 = (kt * KT + 32) / S;
    b0 = (q < T) ? rel_b[(int64_t)q * (2 * S) + kh0] * LOG2E_C : 0.f;
    b1 = (rows_per_tile > 1 && q < T) ? rel_b[(int64_t)q * (2 * S) + kh1] * LOG2E_C : b0;
  };
  float bhn0, bhn1;
  load_bh(0, bhn0, bhn1);

  f32x16 acc_o[DBLK];
#pragma unroll
  for (int db = 0; db < DBLK; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[db][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float s_unscale2 = ldexpf(1.0f, -(EQ + EK)) * LOG2E_C;

  int buf = 0;
  for (int kt = 0; kt < nt; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's part of tile kt has landed
    __builtin_amdgcn_s_barrier();                          // ... everybody's; buffer buf^1 is drained
    if (kt + 1 < nt) issue_tile(kt + 1, buf ^ 1);
    const unsigned char* sb = &smem[buf][0];
    const half_t* sK0 = reinterpret_cast<const half_t*>(sb);
    const half_t* sK1 = reinterpret_cast<const half_t*>(sb + K_UNITS * 16);
    const half_t* sV0 = reinterpret_cast<const half_t*>(sb + 2 * K_UNITS * 16);
    const half_t* sV1 = reinterpret_cast<const half_t*>(sb + (2 * K_UNITS + V_UNITS) * 16);

    // ---- S^T = K Q^T ----
    f32x16 sc[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[blk][r] = 0.f;
      const int row = blk * 32 + l31;
#pragma unroll
      for (int st = 0; st < DSTEPS; ++st) {
        const int c = st * 2 + hh;
        const int off = row * (KCPR * 8) + ((KSWZ ? (c ^ ((row >> 1) & 7)) : c) << 3);     // halves
        const half8_t kh8 = *reinterpret_cast<const half8_t*>(sK0 + off);
        const half8_t kl8 = *reinterpret_cast<const half8_t*>(sK1 + off);
        sc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl8, qh[st], sc[blk], 0, 0, 0);
        sc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh8, qlo[st], sc[blk], 0, 0, 0);
        sc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh8, qh[st], sc[blk], 0, 0, 0);
      }
    }

    // ---- bias, online softmax (per-lane query column) ----
    const float bh0 = bhn0, bh1 = bhn1;
    if (kt + 1 < nt) load_bh(kt + 1, bhn0, bhn1);         // a whole tile ahead of its use
    // everything in the log2 domain: u = s * (2^-12 log2 e) + rel_w' ; the per-block rel_h' term and the running
    // maximum enter as ONE scalar per block inside the exponent (softmax is shift invariant), the 2^14 scale of P too
    float tm0 = -INFINITY, tm1 = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sc[0][r] = fmaf(sc[0][r], s_unscale2, bw[0][r]);
      sc[1][r] = fmaf(sc[1][r], s_unscale2, bw[1][r]);
      tm0 = fmaxf(tm0, sc[0][r]);
      tm1 = fmaxf(tm1, sc[1][r]);
    }
    float tmax = fmaxf(tm0 + bh0, tm1 + bh1);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);                              // log2 units
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);           // 0 on the first tile
    const float k0 = bh0 - m_new + 14.0f, k1 = bh1 - m_new + 14.0f;      // 14 = log2(P_SCALE)
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p0 = __builtin_amdgcn_exp2f(sc[0][r] + k0);
      const float p1 = __builtin_amdgcn_exp2f(sc[1][r] + k1);
      sc[0][r] = p0; sc[1][r] = p1;
      psum += p0 + p1;
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int db = 0; db < DBLK; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc_o[db][r] *= alpha;

    // ---- O^T += V^T P^T as fp16x3 (P hi + lo straight from the score registers, V hi + lo) ----
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float pf[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) pf[t] = sc[s >> 1][8 * (s & 1) + t];
      half8_t ph, pl;
      split8(pf, ph, pl);
#pragma unroll
      for (int db = 0; db < DBLK; ++db) {
        int row = db * 32 + l31;
        if (DBLK * 32 > DH && row >= DH) row = DH - 1;       // rows >= DH feed output rows nobody stores
        // the 8 keys of this half wave (16s + 4hh + {0..3}, 16s + 8 + 4hh + {0..3}) are chunk 2s + hh of the row
        const int off = row * 64 + (((2 * s + hh) ^ ((row >> 1) & 7)) << 3);
        const half8_t vh8 = *reinterpret_cast<const half8_t*>(sV0 + off);
        const half8_t vl8 = *reinterpret_cast<const half8_t*>(sV1 + off);
        acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl8, ph, acc_o[db], 0, 0, 0);
        acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh8, pl, acc_o[db], 0, 0, 0);
        acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh8, ph, acc_o[db], 0, 0, 0);
      }
    }
    buf ^= 1;
  }

  // ---- normalise and store: lane holds O[q][d], d = db*32 + (r&3) + 8*(r>>2) + 4*hh ----
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
int launch_global(const float* qkv, const float* rel, void* ws, float* out, uint16_t* out_hi, uint16_t* out_lo,
                  int out_scale_log2, int Bp, int S, int nh, float scale, hipStream_t s) {
  const int T = S * S;
  const int64_t D = (int64_t)nh * DH;
  const int64_t plane = (int64_t)Bp * nh * T * DH;     // halves per plane
  half_t* Kh = reinterpret_cast<half_t*>(ws);
  half_t* Kl = Kh + plane;
  half_t* Vth = Kl + plane;
  half_t* Vtl = Vth + plane;
  hipLaunchKernelGGL((vit_kv_split_kernel<DH>), dim3(T / KT, nh, Bp), dim3(256), 0, s, qkv, Kh, Kl, Vth, Vtl, T, nh);
  RSP_CHECK_LAUNCH();
  AttnGP p;
  p.q = qkv; p.rel = rel; p.out = out; p.Kh = Kh; p.Kl = Kl; p.Vth = Vth; p.Vtl = Vtl;
  p.out_hi = reinterpret_cast<half_t*>(out_hi); p.out_lo = reinterpret_cast<half_t*>(out_lo);
  p.out_pscale = ldexpf(1.0f, out_scale_log2);
  p.out_rows = (int64_t)Bp * T;
  p.q_bs = (int64_t)T * 3 * D; p.q_ts = 3 * D; p.q_hs = DH;
  p.o_bs = (int64_t)T * D; p.o_ts = D; p.o_hs = DH;
  p.T = T; p.S = S; p.nh = nh; p.scale = scale;
  // dh = 80: the ring is 90 KB -> one block per CU, so make it 8 waves; dh = 64: 64 KB -> two 4-wave blocks per CU
  constexpr int NW = (DH > 64) ? 8 : 4;
  if (T % (NW * 32)) return RSP_EINVAL;
  hipLaunchKernelGGL((attn_global_kernel<DH, NW>), dim3((unsigned)(T / (NW * 32)) * nh * Bp), dim3(NW * 64), 0, s, p);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

}  // namespace

extern "C" int64_t rsp_vit_attention_global_ws_bytes(int32_t Bp, int32_t S, int32_t nh, int32_t dh) {
  return (int64_t)4 * Bp * nh * S * S * dh * (int64_t)sizeof(half_t);
}

extern "C" int rsp_vit_attention_global(const float* qkv, const float* rel, void* workspace, float* out,
                                        uint16_t* out_hi, uint16_t* out_lo, int32_t out_scale_log2, int32_t Bp,
                                        int32_t S, int32_t nh, int32_t dh, float scale, rsp_stream_t stream) {
  if (!qkv || !rel || !workspace || Bp <= 0 || nh <= 0) return RSP_EINVAL;
  if (!out && !(out_hi && out_lo)) return RSP_EINVAL;
  if ((out_hi == nullptr) != (out_lo == nullptr)) return RSP_EINVAL;
  if (!(S == 64 || S == 32)) return RSP_EINVAL;          // T % 256 == 0 and whole key rows per 32-key block
  if (out_hi && ((nh * dh) & 31)) return RSP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (dh == 64) return launch_global<64>(qkv, rel, workspace, out, out_hi, out_lo, out_scale_log2, Bp, S, nh, scale, s);
  if (dh == 80) return launch_global<80>(qkv, rel, workspace, out, out_hi, out_lo, out_scale_log2, Bp, S, nh, scale, s);
  return RSP_EINVAL;
}
