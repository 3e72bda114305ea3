// Evaluation hand-off: COCO run-length encoding of the predicted masks on the device, so that
// results leave the GPU as a few KB per instance instead of H*W bytes.
// Reference: encode_mask_results (mmdet/structures/mask/utils.py:38-53) = pycocotools.mask.encode on the
// Fortran-ordered mask, i.e. cocoapi maskApi.c rleEncode: run lengths of the COLUMN-major pixel stream, first run
// counts zeros.  The compression of the counts to the ASCII string (rleToString) is a few hundred integers per
// instance and stays on the host (rsprompter_amd/rle.py).
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

}  // namespace

extern "C" int rsp_mask_rle(const uint8_t* masks, int32_t k, int32_t H, int32_t W, void* workspace, uint32_t* counts,
                            int32_t* n_counts, int32_t cap, rsp_stream_t stream) {
  if (!masks || !workspace || !counts || !n_counts || k < 0 || H <= 0 || W <= 0 || W > RLE_MAX_W || cap < 2 ||
      (int64_t)H * W > 0x7fffffffLL)
    return RSP_EINVAL;
  if (k == 0) return RSP_OK;
  hipLaunchKernelGGL(mask_rle_kernel, dim3(k), dim3(RLE_THREADS), 0, (hipStream_t)stream, masks, H, W,
                     reinterpret_cast<uint32_t*>(workspace), counts, n_counts, cap);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}
