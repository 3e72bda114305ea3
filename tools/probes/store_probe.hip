// Store-path probe (round 5): how fast can one CU's waves store, by instruction width and by how contiguous the 64 lanes'
// pieces are?  The ping-pong GEMM's exposed epilogue writes a 256 x 256 fp32 tile in ~18 k cycles = 14 B/clk per CU
// (73 cycles per 1 KiB store instruction), whether 8 or 256 CUs run.  Patterns (every lane stores W bytes per instruction):
//   0: 64 lanes x 16 B = 1 KiB contiguous            1: 4 rows x 256 B (row stride 20 KiB: the GEMM's fp32 epilogue)
//   2: 32 rows x 32 B (the untransposed MFMA layout)  3: 64 lanes x 8 B = 512 B contiguous
//   4: 8 pieces x 64 B (b64; the GEMM's plane epilogue: 4 rows x 2 K-blocks)
//   5: 16 rows x 64 B contiguous-by-row pairs (b128, 4 lanes per 64-B plane row, 16 consecutive rows = 1 KiB contiguous)
// hipcc --offload-arch=gfx950 -O2 -o /tmp/store_probe tools/probes/store_probe.hip && /tmp/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int PAT>
__global__ __launch_bounds__(512) void store_kernel(unsigned char* out, int iters, size_t per_block) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned char* base = out + (size_t)blockIdx.x * per_block + (size_t)wave * (per_block / 8);
  const u32x4 v4 = {1u, 2u, 3u, (unsigned)lane};
  const u32x2 v2 = {1u, (unsigned)lane};
  const size_t ROW = 20480;
  for (int it = 0; it < iters; ++it) {
    unsigned char* p = base + (size_t)(it & 63) * 1024;       // revisit a 64 KiB window per wave: no TLB / footprint effects
    if constexpr (PAT == 0) *reinterpret_cast<u32x4*>(p + lane * 16) = v4;
    if constexpr (PAT == 1) *reinterpret_cast<u32x4*>(base + (size_t)((it & 15) * 4 + (lane >> 4)) * ROW + (lane & 15) * 16) = v4;
    if constexpr (PAT == 2) *reinterpret_cast<u32x4*>(base + (size_t)(lane & 31) * ROW + ((it & 31) * 2 + (lane >> 5)) * 16) = v4;
    if constexpr (PAT == 3) *reinterpret_cast<u32x2*>(p + lane * 8) = v2;
    if constexpr (PAT == 4) *reinterpret_cast<u32x2*>(base + (size_t)((lane >> 3) & 1) * (ROW * 8) + (size_t)((it & 15) * 4 + (lane >> 4)) * 64 + (lane & 7) * 8) = v2;
    if constexpr (PAT == 5) *reinterpret_cast<u32x4*>(p + (lane >> 2) * 64 + (lane & 3) * 16) = v4;
  }
}

int main() {
  const int nblk = 256, iters = 4096;
  const size_t per_block = 8ull * 32 * 20480;          // 5 MiB per block
  unsigned char* d;
  hipMalloc(&d, (nblk + 2) * per_block);        // (+ slack: a wave of the last block walks 1.3 MiB of rows)
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[6] = {"b128 1 KiB contiguous", "b128 4 rows x 256 B", "b128 32 rows x 32 B", "b64 512 B contiguous",
                          "b64 8 pieces x 64 B", "b128 16 rows x 64 B (1 KiB contiguous)"};
  for (int blocks : {256, 8}) {
    for (int pat = 0; pat < 6; ++pat) {
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        switch (pat) {
          case 0: hipLaunchKernelGGL(store_kernel<0>, dim3(blocks), dim3(512), 0, 0, d, iters, per_block); break;
          case 1: hipLaunchKernelGGL(store_kernel<1>, dim3(blocks), dim3(512), 0, 0, d, iters, per_block); break;
          case 2: hipLaunchKernelGGL(store_kernel<2>, dim3(blocks), dim3(512), 0, 0, d, iters, per_block); break;
          case 3: hipLaunchKernelGGL(store_kernel<3>, dim3(blocks), dim3(512), 0, 0, d, iters, per_block); break;
          case 4: hipLaunchKernelGGL(store_kernel<4>, dim3(blocks), dim3(512), 0, 0, d, iters, per_block); break;
          case 5: hipLaunchKernelGGL(store_kernel<5>, dim3(blocks), dim3(512), 0, 0, d, iters, per_block); break;
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const double bytes_per_instr = (pat == 3 || pat == 4) ? 512.0 : 1024.0;
      const double instr = 8.0 * iters;                                   // per block (= per CU)
      const double us = best * 1e3;
      printf("%3d blocks  %-40s %8.1f us  %6.1f ns per store instruction per CU  %6.1f GB/s per CU  %7.2f TB/s total\n", blocks,
             names[pat], us, us * 1e3 / instr, instr * bytes_per_instr / us * 1e-3, blocks * instr * bytes_per_instr / us * 1e-6);
    }
  }
  return 0;
}
