// Proposal / detection selection on the GPU: deterministic top-k, box decoding,
// greedy NMS with mmcv's batched (coordinate-offset) semantics.
//
// Reference semantics reproduced bit-for-bit in index space (given identical inputs):
//   RPNHead._predict_by_feat_single / _bbox_post_process   rpn_head.py:134-304
//   delta2bbox                                            delta_xywh_bbox_coder.py:264-361
//   BBoxHead._predict_by_feat_single + multiclass_nms      bbox_head.py:476-571, bbox_nms.py:12-105
//   mmcv.ops.nms / batched_nms (un-vendored): stable score-descending order,
//   suppress when IoU > thr (strict), boxes_for_nms = boxes + id * (max_coord + 1) in fp32.
// Canonical tie order everywhere: score descending, original position ascending.
// Compiled with -ffp-contract=off so that box / IoU arithmetic matches the scalar CPU code.
#include "rsp_common.h"

namespace {

__device__ __forceinline__ uint32_t f2ord(float f) {  // order-preserving float -> uint
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
  const uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(u);
}

// In-LDS bitonic sort of n (power of two) 64-bit keys, DESCENDING, by all threads of the block.
__device__ void bitonic_sort_desc(unsigned long long* keys, int n) {
  for (int k = 2; k <= n; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = keys[i], b = keys[ixj];
          const bool desc = ((i & k) == 0);
          if (desc ? (a < b) : (a > b)) { keys[i] = b; keys[ixj] = a; }
        }
      }
      __syncthreads();
    }
  }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---------------------------------------------------------------------------------------
// RPN top-k per (image, level).  head: [B*HW, ld] rows, objectness logits in cols [0, A).
// Candidate index c = pos * A + a   (== permute(1,2,0).reshape(-1) order, rpn_head.py:194-197).
// ---------------------------------------------------------------------------------------
struct TopkP {
  const float* head[5];
  int HW[5];
  int ld, A, k, num_levels;
  int32_t* sel_idx;    // [B, L, k]
  float* sel_score;    // [B, L, k]
  int32_t* sel_cnt;    // [B, L]
};

constexpr int TK_THREADS = 1024;
constexpr int TK_MAXK = 1024;

__global__ __launch_bounds__(TK_THREADS) void rpn_topk_kernel(const TopkP p) {
  __shared__ unsigned int hist[256];
  __shared__ unsigned long long skeys[TK_MAXK];
  __shared__ int eq_list[TK_MAXK];
  __shared__ unsigned int s_prefix, s_remaining, s_gt_slot, s_eq_cnt;
  const int lvl = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int HW = p.HW[lvl], A = p.A, k = p.k;
  const int n = HW * A;
  const float* base = p.head[lvl] + (int64_t)b * HW * p.ld;
  int32_t* out_idx = p.sel_idx + ((int64_t)b * p.num_levels + lvl) * k;
  float* out_score = p.sel_score + ((int64_t)b * p.num_levels + lvl) * k;
  auto score_at = [&](int c) -> float {
    const int pos = c / A, a = c - pos * A;
    return sigmoidf_(base[(int64_t)pos * p.ld + a]);
  };
  // every candidate's ordered key, eight candidates per thread and trip with their loads issued together: a block walks
  // up to 196 608 candidates (the 256 x 256 level, A = 3) five times with 1024 threads, and one load per trip left the
  // kernel waiting for memory latency 960 times per thread (round 5: 1.16 ms per step before).  The visiting order is
  // free: the histogram and the unordered collection do not depend on it.
  auto scan_keys = [&](auto&& f) {
    constexpr int U = 8;
    int c = tid;
    for (; c + (U - 1) * TK_THREADS < n; c += U * TK_THREADS) {
      float raw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int cc = c + u * TK_THREADS, pos = cc / A;
        raw[u] = base[(int64_t)pos * p.ld + (cc - pos * A)];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) f(c + u * TK_THREADS, f2ord(sigmoidf_(raw[u])));
    }
    for (; c < n; c += TK_THREADS) f(c, f2ord(score_at(c)));
  };
  if (n <= k) {  // nms_pre >= n: keep everything in natural order (rpn_head.py:205)
    for (int c = tid; c < n; c += TK_THREADS) { out_idx[c] = c; out_score[c] = score_at(c); }
    if (tid == 0) p.sel_cnt[b * p.num_levels + lvl] = n;
    return;
  }
  // ---- 4-pass radix select of the k-th largest ordered key ----
  if (tid == 0) { s_prefix = 0; s_remaining = k; }
  __syncthreads();
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    for (int i = tid; i < 256; i += TK_THREADS) hist[i] = 0;
    __syncthreads();
    const uint32_t prefix = s_prefix;
    const uint32_t himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
    scan_keys([&](int, uint32_t u) {
      if ((u & himask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u);
    });
    __syncthreads();
    if (tid == 0) {
      unsigned int rem = s_remaining, acc = 0;
      int bin = 255;
      for (; bin >= 0; --bin) {
        if (acc + hist[bin] >= rem) break;
        acc += hist[bin];
      }
      s_prefix = prefix | ((uint32_t)bin << shift);
      s_remaining = rem - acc;  // how many to take from inside this bin
    }
    __syncthreads();
  }
  const uint32_t kth = s_prefix;
  const int n_eq_take = (int)s_remaining;  // >= 1
  const int n_gt = k - n_eq_take;
  if (tid == 0) { s_gt_slot = 0; s_eq_cnt = 0; }
  __syncthreads();
  // ---- collect: all > kth (unordered), and candidates == kth ----
  scan_keys([&](int c, uint32_t u) {
    if (u > kth) {
      const unsigned int s = atomicAdd(&s_gt_slot, 1u);
      skeys[s] = ((unsigned long long)u << 32) | (uint32_t)(0xffffffffu - (uint32_t)c);
    } else if (u == kth) {
      const unsigned int s = atomicAdd(&s_eq_cnt, 1u);
      if (s < (unsigned)TK_MAXK) eq_list[s] = c;
    }
  });
  __syncthreads();
  const int n_eq = (int)s_eq_cnt;
  if (n_eq <= TK_MAXK) {
    // take the n_eq_take smallest indices among the ties: rank by counting (n_eq is tiny)
    for (int i = tid; i < n_eq; i += TK_THREADS) {
      const int c = eq_list[i];
      int rank = 0;
      for (int j = 0; j < n_eq; ++j) rank += (eq_list[j] < c);
      if (rank < n_eq_take)
        skeys[n_gt + rank] = ((unsigned long long)kth << 32) | (uint32_t)(0xffffffffu - (uint32_t)c);
    }
  } else {
    // many ties (saturated scores): take the first n_eq_take tied candidates in index order with a block-wide
    // ordered compaction, 1024 candidates per round (ballot + wave prefix, wave offsets through LDS)
    __shared__ unsigned int wcnt[TK_THREADS / 64];
    __shared__ unsigned int s_taken;
    if (tid == 0) s_taken = 0;
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    for (int c0 = 0; c0 < n; c0 += TK_THREADS) {
      const unsigned int taken = s_taken;
      if ((int)taken >= n_eq_take) break;
      const int c = c0 + tid;
      const bool flag = c < n && f2ord(score_at(c)) == kth;
      const unsigned long long bal = __ballot(flag);
      if (lane == 0) wcnt[wave] = (unsigned int)__popcll(bal);
      __syncthreads();
      unsigned int off = taken, total = 0;
      for (int w = 0; w < TK_THREADS / 64; ++w) {
        if (w < wave) off += wcnt[w];
        total += wcnt[w];
      }
      if (flag) {
        const unsigned int rank = off + (unsigned int)__popcll(bal & ((1ull << lane) - 1ull));
        if ((int)rank < n_eq_take)
          skeys[n_gt + rank] = ((unsigned long long)kth << 32) | (uint32_t)(0xffffffffu - (uint32_t)c);
      }
      __syncthreads();
      if (tid == 0) s_taken = taken + total;
      __syncthreads();
    }
  }
  for (int i = k + tid; i < TK_MAXK; i += TK_THREADS) skeys[i] = 0ull;
  __syncthreads();
  bitonic_sort_desc(skeys, TK_MAXK);
  for (int i = tid; i < k; i += TK_THREADS) {
    const unsigned long long key = skeys[i];
    out_idx[i] = (int32_t)(0xffffffffu - (uint32_t)(key & 0xffffffffull));
    out_score[i] = ord2f((uint32_t)(key >> 32));
  }
  if (tid == 0) p.sel_cnt[b * p.num_levels + lvl] = k;
}

// ---------------------------------------------------------------------------------------
// delta2bbox (delta_xywh_bbox_coder.py:264-361): every operation is its own fp32 rounding step
// as in the reference's eager tensor expression (the build has -ffp-contract=off), so that the
// decoded boxes -- and with them the NMS decisions -- are bit-identical.
// ---------------------------------------------------------------------------------------
struct BoxCoder {
  float mean[4], std[4];
  float max_ratio, ctr_clamp;
  int clip_border, add_ctr_clamp;
};

__device__ __forceinline__ void decode_box(const float a[4], const float d[4], const BoxCoder& c,
                                           float img_h, float img_w, float out[4]) {
  const float dx = d[0] * c.std[0] + c.mean[0], dy = d[1] * c.std[1] + c.mean[1];
  float dw = d[2] * c.std[2] + c.mean[2], dh = d[3] * c.std[3] + c.mean[3];
  const float px = (a[0] + a[2]) * 0.5f, py = (a[1] + a[3]) * 0.5f;
  const float pw = a[2] - a[0], ph = a[3] - a[1];
  float dxw = pw * dx, dyh = ph * dy;
  if (c.add_ctr_clamp) {  // :340-342: centre shift clamped to +-ctr_clamp pixels, sizes only from above
    dxw = fminf(fmaxf(dxw, -c.ctr_clamp), c.ctr_clamp);
    dyh = fminf(fmaxf(dyh, -c.ctr_clamp), c.ctr_clamp);
    dw = fminf(dw, c.max_ratio);
    dh = fminf(dh, c.max_ratio);
  } else {
    dw = fminf(fmaxf(dw, -c.max_ratio), c.max_ratio);
    dh = fminf(fmaxf(dh, -c.max_ratio), c.max_ratio);
  }
  const float gx = px + dxw, gy = py + dyh;
  const float gw = pw * expf(dw), gh = ph * expf(dh);
  const float x1 = gx - gw * 0.5f, y1 = gy - gh * 0.5f, x2 = gx + gw * 0.5f, y2 = gy + gh * 0.5f;
  if (c.clip_border) {
    out[0] = fminf(fmaxf(x1, 0.f), img_w);
    out[1] = fminf(fmaxf(y1, 0.f), img_h);
    out[2] = fminf(fmaxf(x2, 0.f), img_w);
    out[3] = fminf(fmaxf(y2, 0.f), img_h);
  } else {
    out[0] = x1; out[1] = y1; out[2] = x2; out[3] = y2;
  }
}

__host__ bool coder_from_abi(const RspBoxCoder* in, BoxCoder* c) {
  if (!in || !(in->max_ratio > 0.f) || (in->add_ctr_clamp && !(in->ctr_clamp >= 0.f))) return false;
  for (int i = 0; i < 4; ++i) { c->mean[i] = in->means[i]; c->std[i] = in->stds[i]; }
  c->max_ratio = in->max_ratio; c->ctr_clamp = in->ctr_clamp;
  c->clip_border = in->clip_border != 0; c->add_ctr_clamp = in->add_ctr_clamp != 0;
  return true;
}

// block-wide exclusive scan of one int per thread (blockDim.x <= 1024); returns the exclusive
// prefix, *total receives the block sum.  `tmp` is 17 ints of LDS.
__device__ int block_excl_scan(int v, int* tmp, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) tmp[wave] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    int acc = 0;
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 0; w < nw; ++w) { const int t = tmp[w]; tmp[w] = acc; acc += t; }
    tmp[16] = acc;
  }
  __syncthreads();
  const int res = tmp[wave] + incl - v;
  *total = tmp[16];
  __syncthreads();
  return res;
}

struct RpnDecodeP {
  const float* head[5];
  int HW[5], Wl[5];
  float stride[5];
  const float* base_anchors;  // [L, A, 4]
  const int32_t* sel_idx; const float* sel_score; const int32_t* sel_cnt;  // from rpn_topk
  const float* img_hw;        // [B, 2] (h, w) of img_shape
  int ld, A, k, num_levels, cap;  // cap = candidate capacity per image (>= L*k)
  BoxCoder coder; float min_size;
  float* cand_boxes;   // [B, cap, 4]
  float* cand_scores;  // [B, cap]
  int32_t* cand_ids;   // [B, cap]  level id
  int32_t* cand_src;   // [B, cap]  anchor index within the level (diagnostics / parity tests)
  int32_t* cand_cnt;   // [B]
};

__global__ __launch_bounds__(1024) void rpn_decode_kernel(const RpnDecodeP p) {
  __shared__ int tmp[17];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float img_h = p.img_hw[2 * b], img_w = p.img_hw[2 * b + 1];
  int base_out = 0;
  for (int lvl = 0; lvl < p.num_levels; ++lvl) {
    const int cnt = p.sel_cnt[b * p.num_levels + lvl];
    for (int j0 = 0; j0 < cnt; j0 += blockDim.x) {
      const int j = j0 + tid;
      bool valid = false;
      float box[4] = {0.f, 0.f, 0.f, 0.f};
      float sc = 0.f;
      int c = 0;
      if (j < cnt) {
        c = p.sel_idx[((int64_t)b * p.num_levels + lvl) * p.k + j];
        sc = p.sel_score[((int64_t)b * p.num_levels + lvl) * p.k + j];
        const int pos = c / p.A, a = c - pos * p.A;
        const int y = pos / p.Wl[lvl], x = pos - y * p.Wl[lvl];
        const float* ba = p.base_anchors + ((int64_t)lvl * p.A + a) * 4;
        const float shx = (float)x * p.stride[lvl], shy = (float)y * p.stride[lvl];
        const float an[4] = {ba[0] + shx, ba[1] + shy, ba[2] + shx, ba[3] + shy};
        const float* dp = p.head[lvl] + ((int64_t)b * p.HW[lvl] + pos) * p.ld + p.A + a * 4;
        const float d[4] = {dp[0], dp[1], dp[2], dp[3]};
        decode_box(an, d, p.coder, img_h, img_w, box);
        valid = p.min_size < 0.f || ((box[2] - box[0]) > p.min_size && (box[3] - box[1]) > p.min_size);
      }
      int total;
      const int off = block_excl_scan(valid ? 1 : 0, tmp, &total);
      if (valid) {
        const int64_t o = (int64_t)b * p.cap + base_out + off;
        p.cand_boxes[o * 4 + 0] = box[0]; p.cand_boxes[o * 4 + 1] = box[1];
        p.cand_boxes[o * 4 + 2] = box[2]; p.cand_boxes[o * 4 + 3] = box[3];
        p.cand_scores[o] = sc;
        p.cand_ids[o] = lvl;
        p.cand_src[o] = c;
      }
      base_out += total;
    }
  }
  if (tid == 0) p.cand_cnt[b] = base_out;
}

// ---------------------------------------------------------------------------------------
// R-CNN box head post-processing: softmax, per-class decode, score threshold (bbox_head.py:
// 525-571, bbox_nms.py:40-76).  head: [sum_n, ld] with cols [0, nc] = cls logits,
// [nc+1, nc+1+4nc) = per-class deltas.  candidate order: (roi, class) row-major.
// ---------------------------------------------------------------------------------------
struct BboxPostP {
  const float* head; int ld;
  const float* rois;           // [sum_n, 5]
  const int32_t* roi_start;    // [B+1] prefix of rois per image
  const float* img_hw;         // [B, 2]
  int nc, cap;
  float score_thr;
  BoxCoder coder;
  float* cand_boxes; float* cand_scores; int32_t* cand_ids; int32_t* cand_src; int32_t* cand_cnt;
};

__global__ __launch_bounds__(1024) void bbox_post_kernel(const BboxPostP p) {
  __shared__ int tmp[17];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int r0 = p.roi_start[b], n = p.roi_start[b + 1] - r0;
  const float img_h = p.img_hw[2 * b], img_w = p.img_hw[2 * b + 1];
  const int total_c = n * p.nc;
  int base_out = 0;
  for (int i0 = 0; i0 < total_c; i0 += blockDim.x) {
    const int i = i0 + tid;
    bool valid = false;
    float box[4] = {0.f, 0.f, 0.f, 0.f};
    float sc = 0.f;
    int cls = 0;
    if (i < total_c) {
      const int r = i / p.nc;
      cls = i - r * p.nc;
      const float* row = p.head + (int64_t)(r0 + r) * p.ld;
      float m = row[0];
      for (int j = 1; j <= p.nc; ++j) m = fmaxf(m, row[j]);
      float s = 0.f;
      for (int j = 0; j <= p.nc; ++j) s += expf(row[j] - m);
      sc = expf(row[cls] - m) / s;
      valid = sc > p.score_thr;
      if (valid) {
        const float* roi = p.rois + (int64_t)(r0 + r) * 5 + 1;
        const float an[4] = {roi[0], roi[1], roi[2], roi[3]};
        const float* dp = row + p.nc + 1 + cls * 4;
        const float d[4] = {dp[0], dp[1], dp[2], dp[3]};
        decode_box(an, d, p.coder, img_h, img_w, box);
      }
    }
    int total;
    const int off = block_excl_scan(valid ? 1 : 0, tmp, &total);
    if (valid && base_out + off < p.cap) {
      const int64_t o = (int64_t)b * p.cap + base_out + off;
      p.cand_boxes[o * 4 + 0] = box[0]; p.cand_boxes[o * 4 + 1] = box[1];
      p.cand_boxes[o * 4 + 2] = box[2]; p.cand_boxes[o * 4 + 3] = box[3];
      p.cand_scores[o] = sc;
      p.cand_ids[o] = cls;
      p.cand_src[o] = i;
    }
    base_out += total;
  }
  if (tid == 0) p.cand_cnt[b] = base_out < p.cap ? base_out : p.cap;
}

// ---------------------------------------------------------------------------------------
// Batched NMS (three kernels): prepare (max coordinate, stable sort, id offsets) ->
// 64x64 IoU bit-mask tiles -> sequential greedy reduction by one wave per image.
// ---------------------------------------------------------------------------------------
struct NmsP {
  const float* boxes; const float* scores; const int32_t* ids; const int32_t* cnt;
  int cap;            // candidates capacity per image
  int nsort;          // power of two >= max count
  float iou_thr; int max_out;
  float* sboxes;      // [B, cap, 4] offset boxes in sorted order
  int32_t* sorder;    // [B, cap]    original position of sorted element
  unsigned long long* gkeys;  // [B, nsort] sort keys in memory when nsort > NMS_LDS_KEYS, else null
  unsigned long long* mask;  // [B, cap, words]
  int words;
  int32_t* keep;      // [B, max_out] original positions, score order
  int32_t* keep_cnt;  // [B]
};

constexpr int NMS_LDS_KEYS = 16384;  // 128 KB of LDS; larger candidate sets sort in memory (same block, same network)
__global__ __launch_bounds__(1024) void nms_prepare_kernel(const NmsP p) {
  extern __shared__ unsigned long long skeys_lds[];
  unsigned long long* skeys_dyn = p.gkeys ? p.gkeys + (int64_t)blockIdx.x * p.nsort : skeys_lds;
  __shared__ float red[16];
  __shared__ float s_max;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = p.cnt[b];
  const float* boxes = p.boxes + (int64_t)b * p.cap * 4;
  const float* scores = p.scores + (int64_t)b * p.cap;
  const int32_t* ids = p.ids + (int64_t)b * p.cap;
  float m = -INFINITY;
  for (int i = tid; i < n * 4; i += blockDim.x) m = fmaxf(m, boxes[i]);
  m = rsp_wave_max(m);
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) {
    float mm = red[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) mm = fmaxf(mm, red[w]);
    s_max = mm;
  }
  for (int i = tid; i < p.nsort; i += blockDim.x) {
    unsigned long long key = 0ull;
    if (i < n) key = ((unsigned long long)f2ord(scores[i]) << 32) | (uint32_t)(0xffffffffu - (uint32_t)i);
    skeys_dyn[i] = key;
  }
  __syncthreads();
  bitonic_sort_desc(skeys_dyn, p.nsort);
  const float off_unit = s_max + 1.0f;  // max_coordinate + 1
  for (int i = tid; i < n; i += blockDim.x) {
    const int src = (int)(0xffffffffu - (uint32_t)(skeys_dyn[i] & 0xffffffffull));
    const float off = (float)ids[src] * off_unit;
    float* sb = p.sboxes + ((int64_t)b * p.cap + i) * 4;
    sb[0] = boxes[src * 4 + 0] + off; sb[1] = boxes[src * 4 + 1] + off;
    sb[2] = boxes[src * 4 + 2] + off; sb[3] = boxes[src * 4 + 3] + off;
    p.sorder[(int64_t)b * p.cap + i] = src;
  }
}

__global__ __launch_bounds__(64) void nms_mask_kernel(const NmsP p) {
  const int b = blockIdx.z;
  const int n = p.cnt[b];
  const int nblk = (n + 63) / 64;
  __shared__ float cbx[64][4];
  const int t = threadIdx.x;
  const float* sb = p.sboxes + (int64_t)b * p.cap * 4;
  // the grid covers min(words, 256)^2 tile positions; larger candidate sets stride over the rest (uniform loops)
  for (int rb = blockIdx.y; rb < nblk; rb += gridDim.y) {
    for (int cb = blockIdx.x; cb < nblk; cb += gridDim.x) {
      if (cb < rb) continue;
      __syncthreads();
      const int cj = cb * 64 + t;
      if (cj < n) { cbx[t][0] = sb[cj * 4]; cbx[t][1] = sb[cj * 4 + 1]; cbx[t][2] = sb[cj * 4 + 2]; cbx[t][3] = sb[cj * 4 + 3]; }
      __syncthreads();
      const int ri = rb * 64 + t;
      if (ri >= n) continue;
      const float x1 = sb[ri * 4], y1 = sb[ri * 4 + 1], x2 = sb[ri * 4 + 2], y2 = sb[ri * 4 + 3];
      const float ia = (x2 - x1) * (y2 - y1);
      unsigned long long bits = 0ull;
      const int ncol = min(64, n - cb * 64);
      const int start = (rb == cb) ? t + 1 : 0;
      for (int j = start; j < ncol; ++j) {
        const float xx1 = fmaxf(x1, cbx[j][0]), yy1 = fmaxf(y1, cbx[j][1]);
        const float xx2 = fminf(x2, cbx[j][2]), yy2 = fminf(y2, cbx[j][3]);
        const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
        const float inter = w * h;
        const float ja = (cbx[j][2] - cbx[j][0]) * (cbx[j][3] - cbx[j][1]);
        const float ovr = inter / (ia + ja - inter);
        if (ovr > p.iou_thr) bits |= (1ull << j);
      }
      p.mask[((int64_t)b * p.cap + ri) * p.words + cb] = bits;
    }
  }
}

// NMS_MAXW removal words per lane: 64 lanes * NMS_MAXW words * 64 bits candidates (4: 16384, the RSPrompter sizes; 32: 131072)
constexpr int NMS_MAXW_LARGE = 32;
template <int NMS_MAXW>
__global__ __launch_bounds__(64) void nms_reduce_kernel(const NmsP p) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int n = p.cnt[b];
  const int nblk = (n + 63) / 64;
  unsigned long long remv[NMS_MAXW];
#pragma unroll
  for (int i = 0; i < NMS_MAXW; ++i) remv[i] = 0ull;
  const unsigned long long* mask = p.mask + (int64_t)b * p.cap * p.words;
  const int32_t* sorder = p.sorder + (int64_t)b * p.cap;
  int32_t* keep = p.keep + (int64_t)b * p.max_out;
  int kept = 0;
  // A block's inputs -- its diagonal mask word and its sort-order entry per row -- are requested one block AHEAD, and the
  // rows a block keeps are propagated four at a time with their loads in flight together: the walk is serial by nature
  // (one wave per image), and round 5's counters showed it 81 % of its cycles waiting for one load after the other.
  auto load_blk = [&](int bk, unsigned long long& dm, int& so) {
    const int r = bk * 64 + lane;
    const bool ok = bk < nblk && r < n;
    dm = ok ? mask[(int64_t)r * p.words + bk] : 0ull;
    so = ok ? sorder[r] : 0;
  };
  unsigned long long dmask;
  int sord;
  load_blk(0, dmask, sord);
  for (int blk = 0; blk < nblk && kept < p.max_out; ++blk) {
    unsigned long long dnext;
    int snext;
    load_blk(blk + 1, dnext, snext);
    // removal word of this block lives in lane (blk & 63), slot (blk >> 6)
    unsigned long long mine = 0ull;
#pragma unroll
    for (int i = 0; i < NMS_MAXW; ++i) if (i == (blk >> 6)) mine = remv[i];
    const unsigned int lo = __shfl((unsigned int)(mine & 0xffffffffull), blk & 63, 64);
    const unsigned int hi = __shfl((unsigned int)(mine >> 32), blk & 63, 64);
    unsigned long long cur = ((unsigned long long)hi << 32) | lo;
    const int rows_here = min(64, n - blk * 64);
    unsigned long long kept_bits = 0ull;
    for (int t = 0; t < rows_here; ++t) {
      const unsigned int dlo = __shfl((unsigned int)(dmask & 0xffffffffull), t, 64);
      const unsigned int dhi = __shfl((unsigned int)(dmask >> 32), t, 64);
      const int so = __shfl(sord, t, 64);
      if (!((cur >> t) & 1ull)) {
        if (kept < p.max_out) {
          kept_bits |= (1ull << t);
          if (lane == 0) keep[kept] = so;
          ++kept;
          cur |= ((unsigned long long)dhi << 32) | dlo;
        }
      }
    }
    if (kept >= p.max_out) break;
    // propagate the kept rows of this block to all later words
    constexpr int CH = NMS_MAXW <= 4 ? 4 : 1;              // rows per batch of loads (register budget of the large form)
    unsigned long long kb = kept_bits;
    while (kb) {
      int tt[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        tt[j] = -1;
        if (kb) { tt[j] = __builtin_ctzll(kb); kb &= kb - 1ull; }
      }
      unsigned long long v[CH][NMS_MAXW];
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const unsigned long long* mrow = mask + (int64_t)(blk * 64 + max(tt[j], 0)) * p.words;
#pragma unroll
        for (int i = 0; i < NMS_MAXW; ++i) {
          const int w = lane + 64 * i;
          v[j][i] = (tt[j] >= 0 && w > blk && w < nblk) ? mrow[w] : 0ull;
        }
      }
#pragma unroll
      for (int j = 0; j < CH; ++j)
#pragma unroll
        for (int i = 0; i < NMS_MAXW; ++i) remv[i] |= v[j][i];
    }
    dmask = dnext;
    sord = snext;
  }
  if (lane == 0) p.keep_cnt[b] = kept;
}

// out rows gathered from candidate arrays through keep lists
__global__ void nms_gather_kernel(const float* boxes, const float* scores, const int32_t* ids,
                                  const int32_t* src, int cap, const int32_t* keep,
                                  const int32_t* keep_cnt, int max_out, float* out_boxes,
                                  float* out_scores, int32_t* out_ids, int32_t* out_src) {
  const int b = blockIdx.x;
  const int n = keep_cnt[b];
  for (int i = threadIdx.x; i < max_out; i += blockDim.x) {
    const int64_t o = (int64_t)b * max_out + i;
    if (i < n) {
      const int64_t s = (int64_t)b * cap + keep[o];
      out_boxes[o * 4 + 0] = boxes[s * 4 + 0]; out_boxes[o * 4 + 1] = boxes[s * 4 + 1];
      out_boxes[o * 4 + 2] = boxes[s * 4 + 2]; out_boxes[o * 4 + 3] = boxes[s * 4 + 3];
      out_scores[o] = scores[s];
      out_ids[o] = ids[s];
      if (out_src) out_src[o] = src ? src[s] : keep[o];
    } else {
      out_boxes[o * 4 + 0] = out_boxes[o * 4 + 1] = out_boxes[o * 4 + 2] = out_boxes[o * 4 + 3] = 0.f;
      out_scores[o] = 0.f;
      out_ids[o] = -1;
      if (out_src) out_src[o] = -1;
    }
  }
}

}  // namespace

extern "C" int rsp_rpn_topk(const RspRpnDesc* d, int32_t B, int32_t* sel_idx, float* sel_score,
                            int32_t* sel_cnt, rsp_stream_t stream) {
  if (!d || !sel_idx || !sel_score || !sel_cnt || B <= 0 || d->num_levels < 1 || d->num_levels > 5 ||
      d->nms_pre <= 0 || d->nms_pre > TK_MAXK || d->A <= 0)
    return RSP_EINVAL;
  TopkP p;
  for (int i = 0; i < 5; ++i) {
    const int j = i < d->num_levels ? i : 0;
    p.head[i] = d->head[j]; p.HW[i] = d->H[j] * d->W[j];
    if (!p.head[i]) return RSP_EINVAL;
  }
  p.ld = d->ld; p.A = d->A; p.k = d->nms_pre; p.num_levels = d->num_levels;
  p.sel_idx = sel_idx; p.sel_score = sel_score; p.sel_cnt = sel_cnt;
  hipLaunchKernelGGL(rpn_topk_kernel, dim3(d->num_levels, B), dim3(TK_THREADS), 0, (hipStream_t)stream, p);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

extern "C" int rsp_rpn_decode(const RspRpnDesc* d, int32_t B, const int32_t* sel_idx,
                              const float* sel_score, const int32_t* sel_cnt, const float* img_hw,
                              int32_t cap, float* cand_boxes, float* cand_scores, int32_t* cand_ids,
                              int32_t* cand_src, int32_t* cand_cnt, rsp_stream_t stream) {
  if (!d || !sel_idx || !sel_score || !sel_cnt || !img_hw || !cand_boxes || !cand_scores || !cand_ids ||
      !cand_src || !cand_cnt || B <= 0 || cap < d->num_levels * d->nms_pre || !d->base_anchors)
    return RSP_EINVAL;
  RpnDecodeP p;
  for (int i = 0; i < 5; ++i) {
    const int j = i < d->num_levels ? i : 0;
    p.head[i] = d->head[j]; p.HW[i] = d->H[j] * d->W[j]; p.Wl[i] = d->W[j]; p.stride[i] = d->stride[j];
  }
  p.base_anchors = d->base_anchors;
  p.sel_idx = sel_idx; p.sel_score = sel_score; p.sel_cnt = sel_cnt; p.img_hw = img_hw;
  p.ld = d->ld; p.A = d->A; p.k = d->nms_pre; p.num_levels = d->num_levels; p.cap = cap;
  if (!coder_from_abi(&d->coder, &p.coder)) return RSP_EINVAL;
  p.min_size = d->min_bbox_size;
  p.cand_boxes = cand_boxes; p.cand_scores = cand_scores; p.cand_ids = cand_ids; p.cand_src = cand_src;
  p.cand_cnt = cand_cnt;
  hipLaunchKernelGGL(rpn_decode_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, p);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

extern "C" int rsp_bbox_post(const float* head, int32_t ld, const float* rois, const int32_t* roi_start,
                             const float* img_hw, int32_t B, int32_t num_classes, float score_thr,
                             const RspBoxCoder* coder, int32_t cap, float* cand_boxes,
                             float* cand_scores, int32_t* cand_ids, int32_t* cand_src, int32_t* cand_cnt,
                             rsp_stream_t stream) {
  if (!head || !rois || !roi_start || !img_hw || !coder || !cand_boxes || !cand_scores || !cand_ids ||
      !cand_src || !cand_cnt || B <= 0 || num_classes <= 0 || cap <= 0)
    return RSP_EINVAL;
  BboxPostP p;
  p.head = head; p.ld = ld; p.rois = rois; p.roi_start = roi_start; p.img_hw = img_hw;
  p.nc = num_classes; p.cap = cap; p.score_thr = score_thr;
  if (!coder_from_abi(coder, &p.coder)) return RSP_EINVAL;
  p.cand_boxes = cand_boxes; p.cand_scores = cand_scores; p.cand_ids = cand_ids; p.cand_src = cand_src;
  p.cand_cnt = cand_cnt;
  hipLaunchKernelGGL(bbox_post_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, p);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

extern "C" int64_t rsp_nms_workspace_bytes(int32_t B, int32_t cap) {
  const int64_t words = (cap + 63) / 64;
  int64_t nsort = 1;
  while (nsort < cap) nsort <<= 1;
  const int64_t keys = nsort > 16384 ? (int64_t)B * nsort * sizeof(unsigned long long) : 0;  // NMS_LDS_KEYS
  return (int64_t)B * cap * (4 * sizeof(float) + sizeof(int32_t) + words * sizeof(unsigned long long)) + keys + 256;
}

extern "C" int rsp_batched_nms(const float* boxes, const float* scores, const int32_t* ids,
                               const int32_t* src, const int32_t* cnt, int32_t B, int32_t cap,
                               float iou_thr, int32_t max_out, void* workspace, int32_t* keep,
                               int32_t* keep_cnt, float* out_boxes, float* out_scores, int32_t* out_ids,
                               int32_t* out_src, rsp_stream_t stream) {
  if (!boxes || !scores || !ids || !cnt || !workspace || !keep || !keep_cnt || !out_boxes || !out_scores ||
      !out_ids || B <= 0 || cap <= 0 || cap > 64 * 64 * NMS_MAXW_LARGE || max_out <= 0)
    return RSP_EINVAL;
  int nsort = 1;
  while (nsort < cap) nsort <<= 1;
  NmsP p;
  p.boxes = boxes; p.scores = scores; p.ids = ids; p.cnt = cnt; p.cap = cap; p.nsort = nsort;
  p.iou_thr = iou_thr; p.max_out = max_out;
  p.words = (cap + 63) / 64;
  char* ws = (char*)workspace;
  p.sboxes = (float*)ws; ws += (int64_t)B * cap * 4 * sizeof(float);
  p.mask = (unsigned long long*)ws; ws += (int64_t)B * cap * p.words * sizeof(unsigned long long);
  p.sorder = (int32_t*)ws; ws += (((int64_t)B * cap * sizeof(int32_t)) + 7) / 8 * 8;
  p.gkeys = nsort > NMS_LDS_KEYS ? (unsigned long long*)ws : nullptr;
  p.keep = keep; p.keep_cnt = keep_cnt;
  hipStream_t s = (hipStream_t)stream;
  const size_t smem = p.gkeys ? 0 : (size_t)nsort * sizeof(unsigned long long);
  // always the LDS path's maximum: the attribute is per function, not per launch (concurrent callers of either path)
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&nms_prepare_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)(NMS_LDS_KEYS * sizeof(unsigned long long)));
  if (e != hipSuccess) return RSP_ELAUNCH;
  hipLaunchKernelGGL(nms_prepare_kernel, dim3(B), dim3(1024), smem, s, p);
  const int gw = p.words < 256 ? p.words : 256;
  hipLaunchKernelGGL(nms_mask_kernel, dim3(gw, gw, B), dim3(64), 0, s, p);
  if (p.words <= 64 * 4) hipLaunchKernelGGL(nms_reduce_kernel<4>, dim3(B), dim3(64), 0, s, p);
  else hipLaunchKernelGGL(nms_reduce_kernel<NMS_MAXW_LARGE>, dim3(B), dim3(64), 0, s, p);
  hipLaunchKernelGGL(nms_gather_kernel, dim3(B), dim3(256), 0, s, boxes, scores, ids, src, cap, keep, keep_cnt,
                     max_out, out_boxes, out_scores, out_ids, out_src);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}
