/*
 * CPU oracle (TEST INFRASTRUCTURE ONLY) for the two mmcv operators the anchor path
 * calls and whose source is NOT under /root/reference (mmcv==2.1.0, un-vendored;
 * README.md:131).  Restated from the operators' published semantics
 *  -- PARITY UNPINNED at this boundary.
 *
 * Call sites in the reference:
 *   RoIAlign:    mmdet/models/roi_heads/roi_extractors/base_roi_extractor.py:41-68
 *                (RoIAlign(output_size, spatial_scale=1/stride, sampling_ratio=0,
 *                 pool_mode='avg', aligned=True)), used at
 *                single_level_roi_extractor.py:65-119
 *   nms:         mmdet/models/dense_heads/rpn_head.py:285-286 and
 *                mmdet/models/layers/bbox_nms.py:96 via mmcv.ops.batched_nms
 *
 * Built by oracle/build.py with plain gcc into oracle/_build/liboracle_ops.so.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* feat: [N, C, H, W] fp32 contiguous; rois: [K, 5] (batch_idx, x1, y1, x2, y2);
 * out: [K, C, ph, pw].  avg pooling, aligned flag, adaptive grid if sampling_ratio<=0. */
void oracle_roi_align(const float* feat, int N, int C, int H, int W, const float* rois, int K,
                      int ph, int pw, float spatial_scale, int sampling_ratio, int aligned,
                      float* out) {
  (void)N;
  const float offset = aligned ? 0.5f : 0.0f;
  for (int k = 0; k < K; ++k) {
    const float* r = rois + 5 * k;
    const int b = (int)r[0];
    const float x1 = r[1] * spatial_scale - offset, y1 = r[2] * spatial_scale - offset;
    const float x2 = r[3] * spatial_scale - offset, y2 = r[4] * spatial_scale - offset;
    float rw = x2 - x1, rh = y2 - y1;
    if (!aligned) { rw = fmaxf(rw, 1.0f); rh = fmaxf(rh, 1.0f); }
    const float bin_h = rh / (float)ph, bin_w = rw / (float)pw;
    const int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)ph);
    const int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)pw);
    const float count = (float)((gh * gw) > 1 ? (gh * gw) : 1);
    for (int c = 0; c < C; ++c) {
      const float* f = feat + ((size_t)b * C + c) * H * W;
      for (int py = 0; py < ph; ++py)
        for (int px = 0; px < pw; ++px) {
          float acc = 0.f;
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
              acc += w1 * f[yl * W + xl] + w2 * f[yl * W + xh] + w3 * f[yh * W + xl] + w4 * f[yh * W + xh];
            }
          }
          out[(((size_t)k * C + c) * ph + py) * pw + px] = acc / count;
        }
    }
  }
}

/* Greedy NMS.  boxes [n,4] fp32; `order` = indices sorted by score descending (ties already
 * resolved by the caller: stable, lower index first).  Suppress when IoU > thr (strict), offset=0.
 * Returns the number kept; keep[] receives the kept ORIGINAL indices in score order. */
int oracle_nms(const float* boxes, const int64_t* order, int n, float thr, int64_t* keep) {
  unsigned char* dead = (unsigned char*)calloc((size_t)n + 1, 1);
  float* area = (float*)malloc(sizeof(float) * ((size_t)n + 1));
  for (int i = 0; i < n; ++i) {
    const float* b = boxes + 4 * i;
    area[i] = (b[2] - b[0]) * (b[3] - b[1]);
  }
  int nk = 0;
  for (int _i = 0; _i < n; ++_i) {
    if (dead[_i]) continue;
    const int64_t i = order[_i];
    keep[nk++] = i;
    const float* bi = boxes + 4 * i;
    const float ia = area[i];
    for (int _j = _i + 1; _j < n; ++_j) {
      if (dead[_j]) continue;
      const int64_t j = order[_j];
      const float* bj = boxes + 4 * j;
      const float xx1 = fmaxf(bi[0], bj[0]), yy1 = fmaxf(bi[1], bj[1]);
      const float xx2 = fminf(bi[2], bj[2]), yy2 = fminf(bi[3], bj[3]);
      const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
      const float inter = w * h;
      const float ovr = inter / (ia + area[j] - inter);
      if (ovr > thr) dead[_j] = 1;
    }
  }
  free(dead);
  free(area);
  return nk;
}
