// Probe of the raw-buffer range check on gfx950 (what csrc/gemm_s2.hip's branch-free epilogue relies on):
//   hipcc --offload-arch=gfx950 -O2 -o buf_probe buf_probe.hip && ./buf_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned* buf, unsigned* res, int nrec, int soff) {
  const int t = threadIdx.x;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf, 0, nrec, 0x00020000);
  __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 0, 0x00020000);
  __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc((unsigned*)nullptr, 0, 0, 0x00020000);
  // 0: in-range load with soffset;  1: voffset in range, voffset + soffset beyond num_records;  2: voffset = 0x80000000
  res[t * 8 + 0] = __builtin_amdgcn_raw_buffer_load_b32(r, t * 4, soff, 0);
  res[t * 8 + 1] = __builtin_amdgcn_raw_buffer_load_b32(r, t * 4, nrec, 0);
  res[t * 8 + 2] = __builtin_amdgcn_raw_buffer_load_b32(r, 0x80000000u + t * 4, soff, 0);
  res[t * 8 + 3] = __builtin_amdgcn_raw_buffer_load_b32(r0, t * 4, 0, 0);       // num_records = 0, valid base
  res[t * 8 + 4] = __builtin_amdgcn_raw_buffer_load_b32(rn, t * 4, 0, 0);       // null base, 0 records
  res[t * 8 + 5] = __builtin_amdgcn_raw_buffer_load_b32(r, nrec - 4 + t * 4, 0, 0);   // straddles the end
  // stores: 6: num_records = 0 -> must be dropped; 7: negative soffset with num_records = 0
  __builtin_amdgcn_raw_buffer_store_b32(0xdead0000u + t, r0, 1024 + t * 4, 0, 0);
  __builtin_amdgcn_raw_buffer_store_b32(0xbeef0000u + t, r0, 2048 + t * 4, -64, 0);
  __builtin_amdgcn_raw_buffer_store_b32(0xfeed0000u + t, r, 0x80000000u + 3072 + t * 4, 0, 0);
  __builtin_amdgcn_raw_buffer_store_b32(0x600d0000u + t, r, 3584 + t * 4, 0, 0);          // in range: must land
}
int main() {
  const int n = 4096;
  std::vector<unsigned> h(n);
  for (int i = 0; i < n; ++i) h[i] = 1000 + i;
  unsigned *d, *r;
  hipMalloc(&d, n * 4 * 2); hipMalloc(&r, 64 * 8 * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  hipMemset(d + n, 0, n * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, r, n * 4, 256);
  std::vector<unsigned> o(64 * 8), b(n);
  hipMemcpy(o.data(), r, o.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), d, n * 4, hipMemcpyDeviceToHost);
  printf("lane 0/1/5: load+soff %u %u %u (expect 1064 1065 1069) | voff ok, voff+soff beyond: %u %u %u (0 = soffset is range-checked, else 1000+4096.. garbage) | voff 2^31: %u %u | nrec0: %u %u | null: %u %u | straddle: %u %u %u\n",
         o[0], o[8], o[40], o[1], o[9], o[41], o[2], o[10], o[3], o[11], o[4], o[12], o[5], o[13], o[21]);
  printf("stores: nrec0 [256]=%u (1256 = dropped) | nrec0 neg soff [496]=%u [512]=%u | voff 2^31 [768]=%u | in range [896]=%x (600d0000)\n",
         b[256], b[496], b[512], b[768], b[896]);
  return 0;
}
