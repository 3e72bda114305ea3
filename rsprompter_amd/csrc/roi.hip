// RoIAlign over a channels-last FPN pyramid (mmcv.ops.RoIAlign, pool_mode='avg',
// aligned=True, adaptive sampling grid) fused with SingleRoIExtractor's level mapping
// (single_level_roi_extractor.py:44-119) and with the input-independent extra positional
// encoding of RSPrompterAnchorRoIPromptHead (models.py:1566-1574): bilinear sampling is
// linear, so roi_align(x + pe) = roi_align(x) + roi_align(pe) and x + pe is never written.
//
// One wave per output bin; a lane owns 4 consecutive channels (C == 256 => one float4 per
// lane, 1 KiB coalesced per corner fetch).  HBM/L2 gather bound.
#include "rsp_common.h"

namespace {

struct RoiP {
  const float* feat[4];   // per level [B, H, W, C]
  const float* pe[4];     // per level [H, W, C] or null
  int H[4], W[4];
  float scale[4];         // 1/stride
  const float* rois;      // [K, 5] (batch, x1, y1, x2, y2)
  float* out;             // [K, P, P, C]
  int K, P, C, num_levels, finest_scale;
};

__global__ __launch_bounds__(256) void roi_align_kernel(const RoiP p) {
  const int lane = threadIdx.x & 63;
  const int64_t bin = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int P = p.P, C = p.C;
  if (bin >= (int64_t)p.K * P * P) return;
  const int k = (int)(bin / (P * P));
  const int pr = (int)(bin - (int64_t)k * P * P);
  const int py = pr / P, px = pr - py * P;
  const float* r = p.rois + 5 * k;
  const int b = (int)r[0];
  // map_roi_levels: floor(log2(sqrt(w*h)/finest + 1e-6)) clamped to [0, L-1]
  const float sc = sqrtf((r[3] - r[1]) * (r[4] - r[2]));
  float lf = floorf(log2f(sc / (float)p.finest_scale + 1e-6f));
  lf = fminf(fmaxf(lf, 0.f), (float)(p.num_levels - 1));
  const int lvl = (int)lf;
  const int H = p.H[lvl], W = p.W[lvl];
  const float s = p.scale[lvl];
  const float x1 = r[1] * s - 0.5f, y1 = r[2] * s - 0.5f;
  const float x2 = r[3] * s - 0.5f, y2 = r[4] * s - 0.5f;
  const float rw = x2 - x1, rh = y2 - y1;
  const float bin_h = rh / (float)P, bin_w = rw / (float)P;
  const int gh = (int)ceilf(rh / (float)P), gw = (int)ceilf(rw / (float)P);
  const float count = (float)((gh * gw) > 1 ? (gh * gw) : 1);
  const float* f = p.feat[lvl] + (int64_t)b * H * W * C;
  const float* pe = p.pe[lvl];
  for (int c = lane * 4; c < C; c += 256) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int iy = 0; iy < gh; ++iy) {
      const float yy = y1 + py * bin_h + (iy + 0.5f) * bin_h / (float)gh;
      for (int ix = 0; ix < gw; ++ix) {
        const float xx = x1 + px * bin_w + (ix + 0.5f) * bin_w / (float)gw;
        float y = yy, x = xx;
        if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) continue;
        if (y <= 0.f) y = 0.f;
        if (x <= 0.f) x = 0.f;
        int yl = (int)y, xl = (int)x, yh, xh;
        if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else { yh = yl + 1; }
        if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else { xh = xl + 1; }
        const float ly = y - yl, lx = x - xl, hy = 1.f - ly, hx = 1.f - lx;
        const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
        const int64_t o1 = ((int64_t)yl * W + xl) * C + c, o2 = ((int64_t)yl * W + xh) * C + c;
        const int64_t o3 = ((int64_t)yh * W + xl) * C + c, o4 = ((int64_t)yh * W + xh) * C + c;
        f32x4 v1 = *reinterpret_cast<const f32x4*>(f + o1);
        f32x4 v2 = *reinterpret_cast<const f32x4*>(f + o2);
        f32x4 v3 = *reinterpret_cast<const f32x4*>(f + o3);
        f32x4 v4 = *reinterpret_cast<const f32x4*>(f + o4);
        if (pe) {
          const f32x4 e1 = *reinterpret_cast<const f32x4*>(pe + o1);
          const f32x4 e2 = *reinterpret_cast<const f32x4*>(pe + o2);
          const f32x4 e3 = *reinterpret_cast<const f32x4*>(pe + o3);
          const f32x4 e4 = *reinterpret_cast<const f32x4*>(pe + o4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { v1[j] += e1[j]; v2[j] += e2[j]; v3[j] += e3[j]; v4[j] += e4[j]; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += w1 * v1[j] + w2 * v2[j] + w3 * v3[j] + w4 * v4[j];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = acc[j] / count;
    *reinterpret_cast<f32x4*>(p.out + bin * C + c) = acc;
  }
}

}  // namespace

extern "C" int rsp_roi_align(const RspRoiAlignDesc* d, rsp_stream_t stream) {
  if (!d || !d->rois || !d->out || d->num_levels < 1 || d->num_levels > 4 || d->P <= 0 ||
      d->C <= 0 || (d->C & 3) || d->K < 0)
    return RSP_EINVAL;
  if (d->K == 0) return RSP_OK;
  RoiP p;
  for (int i = 0; i < 4; ++i) {
    const int j = i < d->num_levels ? i : d->num_levels - 1;
    if (!d->feat[j]) return RSP_EINVAL;
    p.feat[i] = d->feat[j]; p.pe[i] = d->pe[j]; p.H[i] = d->H[j]; p.W[i] = d->W[j];
    p.scale[i] = d->spatial_scale[j];
  }
  p.rois = d->rois; p.out = d->out; p.K = d->K; p.P = d->P; p.C = d->C;
  p.num_levels = d->num_levels; p.finest_scale = d->finest_scale;
  const int64_t bins = (int64_t)d->K * d->P * d->P;
  hipLaunchKernelGGL(roi_align_kernel, dim3((unsigned)((bins + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}
