// FP16_OVFL probe (round 5): does MODE.FP16_OVFL (hwreg MODE bit 23) saturate the results of v_cvt_f16_f32 and of gfx950's
// packed v_cvt_pk_f16_f32 at +-65504 instead of +-inf?  If it does, the two v_med3_f32 clamps per value of the plane split
// (rsp_common.h, rsp_split4) can go.  Prints the conversions of a few values with the bit clear and set.
// hipcc --offload-arch=gfx950 -O2 -o /tmp/f16_ovfl_probe tools/probes/f16_ovfl_probe.hip && /tmp/f16_ovfl_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ void probe(const float* in, float* out, int n, int set) {
  if (set) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);        // hwreg(HW_REG_MODE, 23, 1) = 1
  const int i = threadIdx.x;
  if (i >= n) return;
  float a = in[i], b = in[(i + 1) % n];
  asm volatile("" : "+v"(a), "+v"(b));
  const _Float16 s = (_Float16)a;                                            // v_cvt_f16_f32
  const f32x2 ab = {a, b};
  const half2_t pk = __builtin_convertvector(ab, half2_t);                   // v_cvt_pk_f16_f32
  out[3 * i + 0] = (float)s;
  out[3 * i + 1] = (float)pk[0];
  out[3 * i + 2] = (float)pk[1];
}

int main() {
  const float h[] = {1.0f, 65504.0f, 65519.0f, 65520.0f, 70000.0f, 1e6f, 1e30f, -65520.0f, -1e6f, INFINITY, -INFINITY, NAN, 6e-8f, 3e-8f};
  const int n = sizeof(h) / sizeof(h[0]);
  float *din, *dout;
  hipMalloc(&din, sizeof(h)); hipMalloc(&dout, 3 * sizeof(h));
  hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
  for (int set = 0; set < 2; ++set) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, din, dout, n, set);
    float o[3 * 32];
    hipMemcpy(o, dout, 3 * sizeof(h), hipMemcpyDeviceToHost);
    printf("FP16_OVFL = %d\n", set);
    for (int i = 0; i < n; ++i)
      printf("  x = %-12g  cvt_f16 -> %-10g  cvt_pk lane lo -> %-10g  hi (x[i+1] = %g) -> %g\n", h[i], o[3 * i], o[3 * i + 1], h[(i + 1) % n], o[3 * i + 2]);
  }
  return 0;
}
