// Round 6 probe 2: the `v_pk_mul_f32 ... op_sel:[0,1]` chain of pk_opsel_probe.hip next to waves that run the OTHER
// ingredients of the failing kernel.  In sam_upscale2_kernel the two waves of a SIMD run the same code at different places:
// while one multiplies with op_sel:[0,1], its neighbour may be issuing MFMAs into AGPR accumulators, reading them back with
// v_accvgpr_read, running v_exp_f32, packed fmas with SGPR-pair constants (the GELU), LDS fragment reads, 16-byte global loads
// or ds_bpermute.  Here every wave alternates between the chain (phase X: 8 links, issue slot given up around every
// instruction, results checked against exact integer arithmetic) and a disturber block (phase Y) assembled from those
// ingredients by a bit mask; odd waves start with Y, so that SIMD neighbours are in opposite phases.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/pk_opsel_probe2 tools/probes/pk_opsel_probe2.hip && /tmp/pk_opsel_probe2
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define NOP "s_nop 0\n"
#define LINK                                                                                          \
  "v_mov_b32 v60, %[a]\n" NOP "v_mov_b32 v61, %[b]\n" NOP                                             \
  "v_pk_mul_f32 v[62:63], v[60:61], v[52:53] op_sel:[0,1]\n" NOP                                      \
  "v_pk_add_f32 v[56:57], v[56:57], v[62:63]\n" NOP

#define CHAIN                                                                                                          \
  asm volatile("v_mov_b32 v52, %[h0]\nv_mov_b32 v53, %[h1]\nv_mov_b32 v56, 0\nv_mov_b32 v57, 0\n" LINK LINK LINK LINK  \
               LINK LINK LINK LINK "s_nop 7\nv_mov_b32 %[o0], v56\nv_mov_b32 %[o1], v57\n"                             \
               : [o0] "=&v"(o0), [o1] "=&v"(o1)                                                                        \
               : [a] "v"(a), [b] "v"(b), [h0] "v"(h0), [h1] "v"(h1)                                                    \
               : "v52", "v53", "v56", "v57", "v60", "v61", "v62", "v63")

#define MF "v_mfma_f32_32x32x16_f16 a[0:15], v[66:69], v[70:73], a[0:15]\n"
#define D_MFMA "v_mfma_f32_32x32x16_f16 a[0:15], v[66:69], v[70:73], 0\n" MF MF MF MF MF "v_mfma_f32_32x32x16_f16 v[100:115], v[66:69], v[70:73], 0\n" MF MF
#define D_ACC  "s_nop 7\ns_nop 7\nv_accvgpr_read_b32 v84, a0\nv_accvgpr_read_b32 v85, a1\nv_accvgpr_read_b32 v86, a2\nv_accvgpr_read_b32 v87, a3\n" \
               "v_mul_f32 v84, %[sf], v84\nv_mul_f32 v85, %[sf], v85\nv_accvgpr_read_b32 v86, a4\nv_accvgpr_read_b32 v87, a5\n"
#define D_EXP  "v_exp_f32 v88, -v84\nv_exp_f32 v89, -v85\nv_exp_f32 v90, -v86\nv_exp_f32 v91, -v87\nv_exp_f32 v88, -v88\nv_exp_f32 v89, -v89\n"
#define D_PKF  "v_pk_fma_f32 v[92:93], v[84:85], %[sc], v[86:87] op_sel_hi:[1,0,0]\nv_pk_fma_f32 v[92:93], v[92:93], v[84:85], %[sc] op_sel_hi:[1,1,0]\n" \
               "v_pk_mul_f32 v[94:95], v[84:85], -0.5 op_sel_hi:[1,0]\nv_pk_fma_f32 v[92:93], v[94:95], v[88:89], v[92:93]\n"
#define D_LDS  "ds_read_b128 v[74:77], %[q]\nds_read_b128 v[78:81], %[q] offset:64\ns_waitcnt lgkmcnt(0)\n"
#define D_VMEM "global_load_dwordx4 v[116:119], %[p], off\nglobal_load_dwordx4 v[120:123], %[p], off offset:32\n"
#define D_BPERM "ds_bpermute_b32 v96, %[bp], v84\nds_bpermute_b32 v97, %[bp], v85\ns_waitcnt lgkmcnt(0)\n"

#define DISTURB(TXT)                                                                                                      \
  asm volatile("v_mov_b32 v66, 0\nv_mov_b32 v67, 0\nv_mov_b32 v68, 0\nv_mov_b32 v69, 0\nv_mov_b32 v70, 0\nv_mov_b32 v71, 0\n" \
               "v_mov_b32 v72, 0\nv_mov_b32 v73, 0\nv_mov_b32 v84, %[a]\nv_mov_b32 v85, %[b]\nv_mov_b32 v86, %[a]\n"        \
               "v_mov_b32 v87, %[b]\nv_mov_b32 v88, 0\nv_mov_b32 v89, 0\n" TXT "s_waitcnt vmcnt(0) lgkmcnt(0)\n"              \
               :: [a] "v"(a), [b] "v"(b), [sf] "s"(0.5f), [sc] "s"(sc), [q] "v"(lp), [p] "v"(gp), [bp] "v"(bp)                \
               : "memory", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78",    \
                 "v79", "v80", "v81", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", \
                 "v96", "v97", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110",    \
                 "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123",  \
                 "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15")

template <int MASK>
__device__ __forceinline__ void disturb(float a, float b, unsigned long long sc, unsigned lp, const float* gp, unsigned bp) {
  if constexpr (MASK & 1) DISTURB(D_MFMA);
  if constexpr (MASK & 2) DISTURB(D_ACC);
  if constexpr (MASK & 4) DISTURB(D_EXP);
  if constexpr (MASK & 8) DISTURB(D_PKF);
  if constexpr (MASK & 16) DISTURB(D_LDS);
  if constexpr (MASK & 32) DISTURB(D_VMEM);
  if constexpr (MASK & 64) DISTURB(D_BPERM);
  if constexpr ((MASK & 127) == 127) DISTURB(D_MFMA D_EXP D_PKF D_LDS D_VMEM D_ACC D_PKF D_EXP D_BPERM);      // everything interleaved in one block
}

template <int MASK>
__global__ __launch_bounds__(256) void probe(const float* in, unsigned* bad, int iters) {
  const int l = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ float sm[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) sm[i] = in[i];
  __syncthreads();
  const float* gp = in + ((threadIdx.x * 4 + blockIdx.x * 64) & 1016);
  const unsigned lp = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)(sm + ((threadIdx.x * 4) & 1000));
  const unsigned bp = (unsigned)((l ^ 32) * 4);
  const unsigned long long sc = 0x3f0000003e800000ull;            // the pair (0.25, 0.5)
  unsigned nlo = 0, nhi = 0;
  const bool y_first = ((wave + blockIdx.x) & 1) != 0;
  for (int it = 0; it < iters; ++it) {
    const float a = in[(l * 4 + it) & 1023], b = in[(l * 4 + it + 1) & 1023];
    const float h0 = in[(l + it * 3 + 2) & 1023], h1 = in[(l + it * 5 + 7) & 1023];
    float o0, o1;
    if (y_first) disturb<MASK>(a, b, sc, lp, gp, bp);
    CHAIN;
    if (!y_first) disturb<MASK>(a, b, sc, lp, gp, bp);
    const float hx = (MASK & 256) ? h0 : h1;                      // bit 8: self-test of the checker (expects the wrong register)
    nlo += (o0 != 8.f * a * hx);
    nhi += (o1 != 8.f * b * hx);
  }
  if (nlo) atomicAdd(&bad[l >> 4], nlo);
  if (nhi) atomicAdd(&bad[4 + (l >> 4)], nhi);
}

template <int MASK>
void run(const float* din, unsigned* dbad, int blocks, int iters) {
  (void)hipMemset(dbad, 0, 8 * sizeof(unsigned));
  hipLaunchKernelGGL((probe<MASK>), dim3(blocks), dim3(256), 0, 0, din, dbad, iters);
  const hipError_t e1 = hipGetLastError(), e2 = hipDeviceSynchronize();
  if (e1 != hipSuccess || e2 != hipSuccess) printf("LAUNCH FAILED: %s / %s\n", hipGetErrorString(e1), hipGetErrorString(e2));
  unsigned h[8];
  (void)hipMemcpy(h, dbad, sizeof(h), hipMemcpyDeviceToHost);
  printf("disturber mask %3d, %5d blocks of 256: low sums wrong per lane quarter %u %u %u %u, high sums %u %u %u %u  (of %llu sums each)\n",
         MASK, blocks, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], (unsigned long long)64 * blocks * iters);
}

int main() {
  std::vector<float> h(1024);
  srand(5);
  for (auto& w : h) w = (float)(rand() % 13 - 6);
  float* din; unsigned* dbad;
  (void)hipMalloc(&din, h.size() * 4); (void)hipMalloc(&dbad, 8 * sizeof(unsigned));
  (void)hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int iters = 4000;
  for (int blocks : {512, 2048}) {                  // 2 blocks per CU = 2 waves per SIMD at 124 registers; many rounds of them
    run<256>(din, dbad, blocks, iters);               // checker self-test: must report (almost) every sum wrong
    run<127>(din, dbad, blocks, iters);
    run<1>(din, dbad, blocks, iters);
    run<2 | 1>(din, dbad, blocks, iters);
    run<4>(din, dbad, blocks, iters);
    run<8>(din, dbad, blocks, iters);
    run<16>(din, dbad, blocks, iters);
    run<32>(din, dbad, blocks, iters);
    run<64>(din, dbad, blocks, iters);
    run<0>(din, dbad, blocks, iters);
  }
  (void)hipFree(din); (void)hipFree(dbad);
  return 0;
}
