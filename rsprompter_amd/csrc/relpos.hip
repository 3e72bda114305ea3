// hipcc-flags: -fno-slp-vectorize
// Decomposed relative-position terms of the SAM ViT attention (HF:761-801; vit_sam.py:202-221):
//   rel[bh, t, 0:S] = q . Rh[qh - kh + S-1],  rel[bh, t, S:2S] = q . Rw[qw - kw + S-1]     (UNSCALED q)
// Own translation unit because it wants the SLP vectorizer off: packing the FMAs of two table rows into
// v_pk_fma_f32 makes the compiler keep {q, q} pairs of the whole query (hundreds of registers, spills).
#include "rsp_common.h"

namespace {

// rel[bh, t, 0:S] = q . Rh[qh - kh + S-1], rel[bh, t, S:2S] = q . Rw[qw - kw + S-1]
// plain fp32 FMA dot products (same arithmetic class as the reference einsum).
// Block = 64 queries of one (window, head); thread = (query, quarter of the 2S outputs).  The query lives in
// registers, both tables in LDS (rows padded to DH+4 floats: 16-byte aligned, conflict-free float4 reads), so one
// output costs DH FMAs + DH/4 LDS reads instead of 2*DH LDS reads.
template <int DH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8))) void vit_relpos_kernel(const float* __restrict__ qkv,
                                                         const float* __restrict__ rph,
                                                         const float* __restrict__ rpw,
                                                         float* __restrict__ rel, int T, int S,
                                                         int nh, int64_t rows_total) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int LD = DH + 4;
  const int nrow = 2 * S - 1;
  float* sH = smem;                 // [nrow][LD]
  float* sW = sH + nrow * LD;       // [nrow][LD]
  const int tid = threadIdx.x;
  const int h = blockIdx.y;
  // 64 consecutive rows of the [Bp*T] token axis: windows of T = 196 tokens leave no ragged last block
  const int64_t g = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  for (int idx = tid; idx < nrow * (DH / 4); idx += 256) {
    const int r = idx / (DH / 4), c = idx - r * (DH / 4);
    *reinterpret_cast<f32x4*>(sH + r * LD + 4 * c) = *reinterpret_cast<const f32x4*>(rph + (int64_t)r * DH + 4 * c);
    *reinterpret_cast<f32x4*>(sW + r * LD + 4 * c) = *reinterpret_cast<const f32x4*>(rpw + (int64_t)r * DH + 4 * c);
  }
  const bool qok = g < rows_total;
  const int bp = qok ? (int)(g / T) : 0;
  const int q = qok ? (int)(g - (int64_t)bp * T) : 0;
  const int part = tid >> 6;                      // wave-uniform: 0,1 -> rel_h halves, 2,3 -> rel_w halves
  const int64_t tok_stride = (int64_t)3 * nh * DH;
  f32x4 qv[DH / 4];
  {
    const float* src = qkv + ((int64_t)bp * T + q) * tok_stride + (int64_t)h * DH;
#pragma unroll
    for (int c = 0; c < DH / 4; ++c) qv[c] = *reinterpret_cast<const f32x4*>(src + 4 * c);
  }
  __syncthreads();
  if (!qok) return;
  const int qy = q / S, qx = q - qy * S;
  const int half = (((S + 1) / 2) + 1) & ~1;        // even split point: 8-byte stores stay aligned
  const int jb = (part & 1) * half, je = (part & 1) ? S : half;
  const int tab0 = (part < 2) ? 0 : nrow * LD;      // float index of the table inside smem
  const int pos = (part < 2) ? qy : qx;
  float* dst = rel + (((int64_t)bp * nh + h) * T + q) * (2 * S) + (part < 2 ? 0 : S);
  if ((S & 7) == 0) {     // global layers (S = 64): 16-byte stores, a thread fills whole 128-byte lines
#pragma unroll 1
    for (int j = jb; j < je; j += 4) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ro = tab0 + (pos - (j + e) + S - 1) * LD;
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < DH / 4; ++c) {
          const f32x4 t4 = *reinterpret_cast<const f32x4*>(&smem[ro + 4 * c]);
          acc = fmaf(qv[c][0], t4[0], acc);
          acc = fmaf(qv[c][1], t4[1], acc);
          acc = fmaf(qv[c][2], t4[2], acc);
          acc = fmaf(qv[c][3], t4[3], acc);
        }
        o[e] = acc;
        if (e == 1) __builtin_amdgcn_sched_barrier(0);   // two rows' LDS reads in flight at a time, not four
      }
      *reinterpret_cast<f32x4*>(dst + j) = o;
    }
  } else if ((S & 1) == 0) {   // windows (S = 14): 8-byte stores
#pragma unroll 1
    for (int j = jb; j < je; j += 2) {
      float o[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int ro = tab0 + (pos - (j + e) + S - 1) * LD;
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < DH / 4; ++c) {
          const f32x4 t4 = *reinterpret_cast<const f32x4*>(&smem[ro + 4 * c]);
          acc = fmaf(qv[c][0], t4[0], acc);
          acc = fmaf(qv[c][1], t4[1], acc);
          acc = fmaf(qv[c][2], t4[2], acc);
          acc = fmaf(qv[c][3], t4[3], acc);
        }
        o[e] = acc;
      }
      *reinterpret_cast<float2*>(dst + j) = make_float2(o[0], o[1]);
    }
  } else {
#pragma unroll 1
    for (int j = jb; j < je; ++j) {
      const int ro = tab0 + (pos - j + S - 1) * LD;
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < DH / 4; ++c) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(&smem[ro + 4 * c]);
        acc = fmaf(qv[c][0], t4[0], acc);
        acc = fmaf(qv[c][1], t4[1], acc);
        acc = fmaf(qv[c][2], t4[2], acc);
        acc = fmaf(qv[c][3], t4[3], acc);
      }
      dst[j] = acc;
    }
  }
}

}  // namespace

extern "C" int rsp_vit_relpos(const float* qkv, const float* rel_pos_h, const float* rel_pos_w,
                              float* rel, int32_t Bp, int32_t S, int32_t nh, int32_t dh,
                              rsp_stream_t stream) {
  if (!qkv || !rel_pos_h || !rel_pos_w || !rel || Bp <= 0 || S <= 0 || S > 64 || nh <= 0)
    return RSP_EINVAL;
  const int T = S * S;
  const int64_t rows_total = (int64_t)Bp * T;
  dim3 grid((unsigned)((rows_total + 63) / 64), nh, 1);
  const size_t smem = (size_t)(2 * (2 * S - 1)) * (dh + 4) * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  if (dh == 64) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&vit_relpos_kernel<64>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((vit_relpos_kernel<64>), grid, dim3(256), smem, s, qkv, rel_pos_h, rel_pos_w, rel, T, S, nh, rows_total);
  } else if (dh == 80) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&vit_relpos_kernel<80>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((vit_relpos_kernel<80>), grid, dim3(256), smem, s, qkv, rel_pos_h, rel_pos_w, rel, T, S, nh, rows_total);
  } else {
    return RSP_EINVAL;
  }
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

