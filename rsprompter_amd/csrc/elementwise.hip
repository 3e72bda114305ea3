// HBM-bound data-movement kernels of the path (pre-process, patchify, pooling).
#include "rsp_common.h"

namespace {

// out[(b*gh + py)*gw + px, c*p*p + ky*p + kx] = img[b, c, py*p + ky, px*p + kx]
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img,
                                                       float* __restrict__ out, int B, int C,
                                                       int H, int W, int p, int64_t total4) {
  const int gh = H / p, gw = W / p;
  const int K = C * p * p;
  const int k4n = K / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / k4n;
    const int k = (int)(i - row * k4n) * 4;
    const int c = k / (p * p);
    const int rem = k - c * p * p;
    const int ky = rem / p, kx = rem - ky * p;
    const int px = (int)(row % gw);
    const int64_t t = row / gw;
    const int py = (int)(t % gh);
    const int b = (int)(t / gh);
    const float* src = img + (((int64_t)b * C + c) * H + (py * p + ky)) * W + px * p + kx;
    *reinterpret_cast<f32x4*>(out + row * K + k) = *reinterpret_cast<const f32x4*>(src);
  }
}

// DetDataPreprocessor: dst[b, c, y, x] = (src[c', y, x] - mean[c]) / std[c] inside the image,
// pad_value outside; c' = 2 - c when swap_rb (BGR -> RGB happens BEFORE normalisation, so
// mean/std index the output channel).  One launch per image (sources are separate tensors).
template <typename T>
__global__ __launch_bounds__(256) void preprocess_kernel(const T* __restrict__ src,
                                                         float* __restrict__ dst, int H, int W,
                                                         int Hp, int Wp, float m0, float m1,
                                                         float m2, float s0, float s1, float s2,
                                                         int swap_rb, float pad_value) {
  const int64_t total = (int64_t)3 * Hp * Wp;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % Wp);
    const int64_t t = i / Wp;
    const int y = (int)(t % Hp);
    const int c = (int)(t / Hp);
    float v = pad_value;
    if (y < H && x < W) {
      const int cs = swap_rb ? 2 - c : c;
      const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
      const float sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
      v = ((float)src[((int64_t)cs * H + y) * W + x] - mean) / sd;
    }
    dst[i] = v;
  }
}

}  // namespace

static inline int grid_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

extern "C" int rsp_patchify(const float* img, float* out, int32_t B, int32_t C, int32_t H,
                            int32_t W, int32_t patch, rsp_stream_t stream) {
  if (!img || !out || B <= 0 || C <= 0 || patch <= 0 || (patch & 3) || H % patch || W % patch)
    return RSP_EINVAL;
  const int64_t total4 = (int64_t)B * (H / patch) * (W / patch) * C * patch * patch / 4;
  hipLaunchKernelGGL(patchify_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, img,
                     out, B, C, H, W, patch, total4);
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}

extern "C" int rsp_preprocess(const void* src, int32_t src_is_u8, float* dst, int32_t H, int32_t W,
                              int32_t Hp, int32_t Wp, const float* mean3, const float* std3,
                              int32_t swap_rb, float pad_value, rsp_stream_t stream) {
  if (!src || !dst || !mean3 || !std3 || H <= 0 || W <= 0 || Hp < H || Wp < W) return RSP_EINVAL;
  const int64_t total = (int64_t)3 * Hp * Wp;
  hipStream_t s = (hipStream_t)stream;
  if (src_is_u8) {
    hipLaunchKernelGGL((preprocess_kernel<uint8_t>), dim3(grid_for(total)), dim3(256), 0, s,
                       (const uint8_t*)src, dst, H, W, Hp, Wp, mean3[0], mean3[1], mean3[2], std3[0],
                       std3[1], std3[2], swap_rb, pad_value);
  } else {
    hipLaunchKernelGGL((preprocess_kernel<float>), dim3(grid_for(total)), dim3(256), 0, s,
                       (const float*)src, dst, H, W, Hp, Wp, mean3[0], mean3[1], mean3[2], std3[0],
                       std3[1], std3[2], swap_rb, pad_value);
  }
  RSP_CHECK_LAUNCH();
  return RSP_OK;
}
