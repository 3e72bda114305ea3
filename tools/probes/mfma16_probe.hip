// Probe of v_mfma_f32_16x16x32_f16 (gfx950): which A / B element each (lane, slot) supplies and which D element each
// (lane, register) receives.  Random small integers go in as halves, the host recomputes D = A B under the hypothesis
//   A[m = l % 16][k = 8 (l / 16) + j],  B[k = 8 (l / 16) + j][n = l % 16],  D[m = 4 (l / 16) + r][n = l % 16]
// and under the transposed-D alternative, and prints which one the device agrees with (round 5: the folded attention as six
// waves of 16 columns needs this layout; the lane-level emulator must not be taught a layout nobody has compared).
//   hipcc --offload-arch=gfx950 -O2 -o mfma16_probe tools/probes/mfma16_probe.hip && ./mfma16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, const float* b, float* d) {
  const int l = threadIdx.x;
  half8 av, bv;
  for (int j = 0; j < 8; ++j) { av[j] = (_Float16)a[l * 8 + j]; bv[j] = (_Float16)b[l * 8 + j]; }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}
int main() {
  float ha[512], hb[512], hd[256];
  srand(7);
  for (int i = 0; i < 512; ++i) { ha[i] = (float)(rand() % 7 - 3); hb[i] = (float)(rand() % 5 - 2); }
  float *da, *db, *dd;
  hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dd, sizeof(hd));
  hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
  hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
  float A[16][32], B[32][16];
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 8; ++j) { A[l % 16][8 * (l / 16) + j] = ha[l * 8 + j]; B[8 * (l / 16) + j][l % 16] = hb[l * 8 + j]; }
  int bad0 = 0, bad1 = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int m0 = 4 * (l / 16) + r, n0 = l % 16;          // hypothesis
      const int m1 = l % 16, n1 = 4 * (l / 16) + r;          // transposed D
      float s0 = 0.f, s1 = 0.f;
      for (int kk = 0; kk < 32; ++kk) { s0 += A[m0][kk] * B[kk][n0]; s1 += A[m1][kk] * B[kk][n1]; }
      bad0 += s0 != hd[l * 4 + r]; bad1 += s1 != hd[l * 4 + r];
    }
  printf("v_mfma_f32_16x16x32_f16: A[l%%16][8(l/16)+j] B[8(l/16)+j][l%%16] with D[4(l/16)+r][l%%16]: %d mismatches; with D transposed: %d\n",
         bad0, bad1);
  printf("%s\n", bad0 == 0 ? "LAYOUT CONFIRMED" : (bad1 == 0 ? "D IS TRANSPOSED" : "NEITHER: dump and derive"));
  if (bad0 && bad1)
    for (int l = 0; l < 64; ++l) printf("lane %2d: %g %g %g %g\n", l, hd[4 * l], hd[4 * l + 1], hd[4 * l + 2], hd[4 * l + 3]);
  return 0;
}
