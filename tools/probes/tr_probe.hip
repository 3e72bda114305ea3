// Probe of ds_read_b64_tr_b16 (gfx950 LDS transpose read): LDS holds lds[i] = i (16-bit); every lane passes its own
// address and the program prints which source elements each lane received.  hipcc --offload-arch=gfx950 -O2 -o tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(const int* addr, unsigned short* out) {
  __shared__ unsigned short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int a = addr[threadIdx.x];
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + a));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)r[j];
}
int main() {
  int h_addr[64]; unsigned short h_out[256];
  int* d_addr; unsigned short* d_out;
  hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
  for (int pat = 0; pat < 3; ++pat) {
    for (int l = 0; l < 64; ++l) {
      const int g = l >> 4, i = l & 15;
      if (pat == 0) h_addr[l] = l * 4;                                   // 8 contiguous bytes per lane, lane-linear
      else if (pat == 1) h_addr[l] = g * 1000 + (i >> 2) * 100 + (i & 3) * 4;   // [4 rows][16 cols], row pitch 100
      else h_addr[l] = g * 1000 + i * 100;                               // 16 rows, 4 contiguous elements each
    }
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("pattern %d\n", pat);
    for (int l = 0; l < 64; ++l)
      printf("  lane %2d addr %4d -> %4d %4d %4d %4d\n", l, h_addr[l], h_out[4 * l], h_out[4 * l + 1], h_out[4 * l + 2], h_out[4 * l + 3]);
  }
  return 0;
}
