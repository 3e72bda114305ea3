// Evaluation hand-off: COCO run-length encoding of the predicted masks on the device, so that
// results leave the GPU as a few KB per instance instead of H*W bytes.
// Reference: encode_mask_results (mmdet/structures/mask/utils.py:38-53) = pycocotools.mask.encode on the
// Fortran-ordered mask, i.e. cocoapi maskApi.c rleEncode: run lengths of the COLUMN-major pixel stream, first run
// counts zeros.  The compression of the counts to the ASCII string (rleToString) is a few hundred integers per
// instance; round 3 moved it onto the device as well (rsp_rle_to_string below), so that the multi-GPU result exchange
// carries finished strings and no rank loops over instances in Python.
#include "rsp_common.h"

namespace {

constexpr int RLE_THREADS = 1024;
constexpr int RLE_MAX_W = 8192;

// one block per mask; thread t owns columns t, t + 1024, ... so that at every step the block reads one row segment
// of the row-major mask contiguously.  Pass 1 counts the value changes per column, a block scan turns them into
// per-column output offsets, pass 2 writes the change positions j = x*H + y, the last step takes differences.
__global__ __launch_bounds__(RLE_THREADS) void mask_rle_kernel(const uint8_t* __restrict__ masks, int H, int W,
                                                               uint32_t* __restrict__ pos_ws,
                                                               uint32_t* __restrict__ counts,
                                                               int32_t* __restrict__ n_counts, int cap) {
  __shared__ int colcnt[RLE_MAX_W];
  __shared__ int part[RLE_THREADS];
  __shared__ int s_total;
  const int m = blockIdx.x, tid = threadIdx.x;
  const uint8_t* mk = masks + (int64_t)m * H * W;
  uint32_t* pos = pos_ws + (int64_t)m * cap;
  uint32_t* out = counts + (int64_t)m * cap;
  auto prev_of_col = [&](int x) -> int { return x == 0 ? 0 : (mk[(int64_t)(H - 1) * W + x - 1] != 0); };
  for (int x = tid; x < W; x += RLE_THREADS) {
    int prev = prev_of_col(x), c = 0;
    for (int y = 0; y < H; ++y) {
      const int v = mk[(int64_t)y * W + x] != 0;
      c += (v != prev);
      prev = v;
    }
    colcnt[x] = c;
  }
  __syncthreads();
  // exclusive scan over the W column counts: per-thread chunks + Hillis-Steele over the 1024 partial sums
  const int C = (W + RLE_THREADS - 1) / RLE_THREADS;
  int mine = 0;
  for (int i = 0; i < C; ++i) { const int x = tid * C + i; if (x < W) mine += colcnt[x]; }
  part[tid] = mine;
  __syncthreads();
  for (int o = 1; o < RLE_THREADS; o <<= 1) {
    const int v = tid >= o ? part[tid - o] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  if (tid == RLE_THREADS - 1) s_total = part[tid];
  {
    int run = part[tid] - mine;                      // exclusive prefix of this thread's chunk
    for (int i = 0; i < C; ++i) {
      const int x = tid * C + i;
      if (x < W) { const int c = colcnt[x]; colcnt[x] = run; run += c; }
    }
  }
  __syncthreads();
  const int ntrans = s_total;
  if (ntrans + 1 > cap) {                            // caller retries with a larger capacity
    if (tid == 0) n_counts[m] = -(ntrans + 1);
    return;
  }
  for (int x = tid; x < W; x += RLE_THREADS) {
    int prev = prev_of_col(x), k = colcnt[x];
    for (int y = 0; y < H; ++y) {
      const int v = mk[(int64_t)y * W + x] != 0;
      if (v != prev) pos[k++] = (uint32_t)(x * H + y);
      prev = v;
    }
  }
  __syncthreads();
  const uint32_t N = (uint32_t)H * (uint32_t)W;
  for (int i = tid; i <= ntrans; i += RLE_THREADS) {
    const uint32_t lo = i == 0 ? 0u : pos[i - 1];
    const uint32_t hi = i == ntrans ? N : pos[i];
    out[i] = hi - lo;
  }
  if (tid == 0) n_counts[m] = ntrans + 1;
}

// Round 4: the same algorithm on CELLS of 4 columns x H / G rows (W % 4 == 0, W <= 4096).  A value change is local
// (v[y] != v[y - 1], the first row of a column against the last row of the previous one), so any partition of a column's
// rows can be counted independently once each cell reads the value in front of it; the prefix scan then runs over the
// cells in column-major order (x, row group).  Every thread reads 4 bytes of a row at a time -- a wave 256 contiguous
// bytes instead of 64 -- and walks H / G rows instead of H: with one block per mask and byte loads the codec of a step's
// 800 masks kept a side stream busy for ~25 ms and cost the compute stream 7 ms per step (round 4,
// tests/test_gpu_dist.py::test_bench_step_loop_exchange_costs_no_gpu_time).
__global__ __launch_bounds__(RLE_THREADS) void mask_rle4_kernel(const uint8_t* __restrict__ masks, int H, int W, int G,
                                                                uint32_t* __restrict__ pos_ws,
                                                                uint32_t* __restrict__ counts,
                                                                int32_t* __restrict__ n_counts, int cap) {
  __shared__ int cellcnt[RLE_MAX_W];                   // [x * G + g], W * G <= 8192
  __shared__ int part[RLE_THREADS];
  __shared__ int s_total;
  const int m = blockIdx.x, tid = threadIdx.x;
  const uint8_t* mk = masks + (int64_t)m * H * W;
  uint32_t* pos = pos_ws + (int64_t)m * cap;
  uint32_t* out = counts + (int64_t)m * cap;
  const int nq = W >> 2, ncell = nq * G;
  const int Hg = (H + G - 1) / G;
  // the value in front of row y0 of column x (column-major stream): row y0 - 1, or the last row of column x - 1, or 0
  auto front = [&](int x, int y0) -> int {
    if (y0 > 0) return mk[(int64_t)(y0 - 1) * W + x] != 0;
    return x == 0 ? 0 : (mk[(int64_t)(H - 1) * W + x - 1] != 0);
  };
  for (int cell = tid; cell < ncell; cell += RLE_THREADS) {
    const int cq = cell % nq, g = cell / nq;          // consecutive threads: consecutive column quads of one row group
    const int y0 = g * Hg, y1 = min(H, y0 + Hg);
    int prev[4], c[4] = {0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 4; ++e) prev[e] = y0 < y1 ? front(4 * cq + e, y0) : 0;
    // rows in batches of 8: the eight loads are independent of the running state and go out together (one load per row
    // and iteration made the walk a chain of ~500 ns memory latencies: 261 us per 100 masks)
    for (int yb = y0; yb < y1; yb += 8) {
      uint32_t w8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) w8[j] = *reinterpret_cast<const uint32_t*>(mk + (int64_t)min(yb + j, y1 - 1) * W + 4 * cq);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (yb + j < y1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int v = ((w8[j] >> (8 * e)) & 0xffu) != 0;
            c[e] += (v != prev[e]);
            prev[e] = v;
          }
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) cellcnt[(4 * cq + e) * G + g] = c[e];
  }
  __syncthreads();
  const int NE = W * G;
  const int C = (NE + RLE_THREADS - 1) / RLE_THREADS;
  int mine = 0;
  for (int i = 0; i < C; ++i) { const int x = tid * C + i; if (x < NE) mine += cellcnt[x]; }
  part[tid] = mine;
  __syncthreads();
  for (int o = 1; o < RLE_THREADS; o <<= 1) {
    const int v = tid >= o ? part[tid - o] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  if (tid == RLE_THREADS - 1) s_total = part[tid];
  {
    int run = part[tid] - mine;                      // exclusive prefix of this thread's chunk
    for (int i = 0; i < C; ++i) {
      const int x = tid * C + i;
      if (x < NE) { const int c = cellcnt[x]; cellcnt[x] = run; run += c; }
    }
  }
  __syncthreads();
  const int ntrans = s_total;
  if (ntrans + 1 > cap) {                            // caller retries with a larger capacity
    if (tid == 0) n_counts[m] = -(ntrans + 1);
    return;
  }
  for (int cell = tid; cell < ncell; cell += RLE_THREADS) {
    const int cq = cell % nq, g = cell / nq;
    const int y0 = g * Hg, y1 = min(H, y0 + Hg);
    int prev[4], k[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { prev[e] = y0 < y1 ? front(4 * cq + e, y0) : 0; k[e] = cellcnt[(4 * cq + e) * G + g]; }
    for (int yb = y0; yb < y1; yb += 8) {
      uint32_t w8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) w8[j] = *reinterpret_cast<const uint32_t*>(mk + (int64_t)min(yb + j, y1 - 1) * W + 4 * cq);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (yb + j < y1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int v = ((w8[j] >> (8 * e)) & 0xffu) != 0;
            if (v != prev[e]) pos[k[e]++] = (uint32_t)((4 * cq + e) * H + yb + j);
            prev[e] = v;
          }
        }
      }
    }
  }
  __syncthreads();
  const uint32_t N = (uint32_t)H * (uint32_t)W;
  for (int i = tid; i <= ntrans; i += RLE_THREADS) {
    const uint32_t lo = i == 0 ? 0u : pos[i - 1];
    const uint32_t hi = i == ntrans ? N : pos[i];
    out[i] = hi - lo;
  }
  if (tid == 0) n_counts[m] = ntrans + 1;
}

// ---- counts -> COCO ASCII string (cocoapi maskApi.c rleToString; structures/mask/utils.py:38-53 hands these strings to
// CocoMetric): count i is delta-coded against count i-2 from i = 3 on, then written 5 bits per character, least
// significant group first, bit 5 = "more follows", + 48.  A 32-bit value needs at most 7 characters.
__device__ __forceinline__ int rle_encode_one(const uint32_t* __restrict__ c, int i, unsigned char* ch) {
  int x = (int)c[i];
  if (i > 2) x -= (int)c[i - 2];
  int n = 0;
  bool more = true;
  while (more) {
    int v = x & 0x1f;
    x >>= 5;                                          // arithmetic: deltas may be negative
    more = (v & 0x10) ? (x != -1) : (x != 0);
    if (more) v |= 0x20;
    ch[n++] = (unsigned char)(v + 48);
  }
  return n;
}

constexpr int STR_THREADS = 256;

// one block per instance: length of its string (n_counts[m] <= 0 -- capacity overflow of rsp_mask_rle -- gives 0)
__global__ __launch_bounds__(STR_THREADS) void rle_strlen_kernel(const uint32_t* __restrict__ counts,
                                                                 const int32_t* __restrict__ n_counts, int cap,
                                                                 int32_t* __restrict__ lens) {
  __shared__ int part[STR_THREADS / 64];
  const int m = blockIdx.x, tid = threadIdx.x;
  const int n = max(n_counts[m], 0);
  const uint32_t* c = counts + (int64_t)m * cap;
  int mine = 0;
  unsigned char ch[8];
  for (int i = tid; i < n; i += STR_THREADS) mine += rle_encode_one(c, i, ch);
  mine = (int)rsp_wave_sum((float)mine);              // < 2^24 per wave: exact in fp32
  if ((tid & 63) == 0) part[tid >> 6] = mine;
  __syncthreads();
  if (tid == 0) {
    int t = 0;
    for (int w = 0; w < STR_THREADS / 64; ++w) t += part[w];
    lens[m] = t;
  }
}

// exclusive scan of the k lengths -> offs[0..k] (one block; k is a few thousand at most)
__global__ __launch_bounds__(1024) void rle_stroffs_kernel(const int32_t* __restrict__ lens, int k, int64_t* __restrict__ offs) {
  __shared__ long long part[1024];
  const int tid = threadIdx.x;
  const int C = (k + 1023) / 1024;
  long long mine = 0;
  for (int i = 0; i < C; ++i) { const int j = tid * C + i; if (j < k) mine += lens[j]; }
  part[tid] = mine;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const long long v = tid >= o ? part[tid - o] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  long long run = part[tid] - mine;
  for (int i = 0; i < C; ++i) { const int j = tid * C + i; if (j < k) { offs[j] = run; run += lens[j]; } }
  if (tid == 1023) offs[k] = part[tid];
}

// one block per instance: the characters of chunk after chunk of 256 counts, positions from a block scan
__global__ __launch_bounds__(STR_THREADS) void rle_strwrite_kernel(const uint32_t* __restrict__ counts,
                                                                   const int32_t* __restrict__ n_counts, int cap,
                                                                   const int64_t* __restrict__ offs,
                                                                   uint8_t* __restrict__ flat, int64_t flat_cap) {
  __shared__ int sc[STR_THREADS];
  __shared__ int s_base;
  const int m = blockIdx.x, tid = threadIdx.x;
  const int n = max(n_counts[m], 0);
  if (offs[m + 1] > flat_cap) return;                 // does not fit: the caller sees offs[k] > flat_cap and retries
  const uint32_t* c = counts + (int64_t)m * cap;
  uint8_t* out = flat + offs[m];
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += STR_THREADS) {
    const int i = i0 + tid;
    unsigned char ch[8];
    const int len = i < n ? rle_encode_one(c, i, ch) : 0;
    sc[tid] = len;
    __syncthreads();
    for (int o = 1; o < STR_THREADS; o <<= 1) {
      const int v = tid >= o ? sc[tid - o] : 0;
      __syncthreads();
      sc[tid] += v;
      __syncthreads();
    }
    const int pos = s_base + sc[tid] - len;
    for (int j = 0; j < len; ++j) out[pos + j] = ch[j];
    __syncthreads();
    if (tid == STR_THREADS - 1) s_base += sc[tid];
    __syncthreads();
  }
}

}  // namespace

extern "C" int rsp_rle_to_string(const uint32_t* counts, const int32_t* n_counts, int32_t k, int32_t cap, int32_t* lens,
                                 int64_t* offs, uint8_t* flat, int64_t flat_cap, rsp_stream_t stream) {
  if (!counts || !n_counts || !lens || !offs || !flat || k < 0 || cap < 2 || flat_cap < 0) return RSP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (k > 0) {
    hipLaunchKernelGGL(rle_strlen_kernel, dim3(k), dim3(STR_THREADS), 0, s, counts, n_counts, cap, lens);
    RSP_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(rle_stroffs_kernel, dim3(1), dim3(1024), 0, s, lens, k, offs);
  RSP_CHECK_LAUNCH();
  if (k > 0) {
    hipLaunchKernelGGL(rle_strwrite_kernel, dim3(k), dim3(STR_THREADS), 0, s, counts, n_counts, cap, offs, flat, flat_cap);
    RSP_CHECK_LAUNCH();
  }
  return RSP_OK;
}

extern "C" int rsp_mask_rle(const uint8_t* masks, int32_t k, int32_t H, int32_t W, void* workspace, uint32_t* counts,
                            int32_t* n_counts, int32_t cap, rsp_stream_t stream) {
  if (!masks || !workspace || !counts || !n_counts || k < 0 || H <= 0 || W <= 0 || W > RLE_MAX_W || cap < 2 ||
      (int64_t)H * W > 0x7fffffffLL)
    return RSP_EINVAL;
  if (k == 0) return RSP_OK;
  if ((W & 3) == 0 && W <= 4096 && (reinterpret_cast<uintptr_t>(masks) & 3) == 0) {
    int G = RLE_THREADS / (W >> 2);                   // row groups: as many cells as threads, W * G <= RLE_MAX_W
    if (G < 1) G = 1;
    while (G > 1 && (W * G > RLE_MAX_W || G > H)) G >>= 1;
    hipLaunchKernelGGL(mask_rle4_kernel, dim3(k), dim3(RLE_THREADS), 0, (hipStream_t)stream, masks, H, W, G,
                       reinterpret_cast<uint32_t*>(workspace), counts, n_counts, cap);
    RSP_CHECK_LAUNCH();
    return RSP_OK;
  }
  hipLaunchKernelGGL(mask_rle_kernel, dim3(k), dim3(RLE_THREADS), 0, (hipStream_t)stream, masks, H, W,
                     reinterpret_cast<uint32_t*>(workspace), counts, n_counts, cap);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}
