// Round 6 probe: is a VALU write to the SOURCE registers of a v_mfma_f32_32x32x16_f16 safe right behind it?
//
// Why: the one GPU test that went red on the driver's box in round 5 (multimask mask decoder) was traced to
// sam_upscale2_kernel producing 16 wrong outputs (pixels 16..31 of one wave tile, one sub-pixel) about once in 5-1500 runs,
// only with several blocks co-resident on a CU, only for some instruction schedules.  hipcc's schedule has, right behind
// the LAST MFMA of a dependent chain   v_mfma a[0:15], v[0:3], v[130:133], a[0:15]   a VALU instruction that overwrites
// v[0:1]  (v_lshlrev_b64 v[0:1], ...): a write-after-read the compiler regards as safe (sources are read at issue).
// This probe issues exactly that pattern with fixed registers:
//     CHAIN dependent MFMAs (same A, B; each waits for its predecessor's accumulator) ; DIST x s_nop 0 ;
//     v_mov_b32 into two of the last MFMA's source registers ; wait ; read the accumulators
// and compares with the same chain without the overwrite, for every (DIST, CHAIN, which half of A / B), with 1, 2 and 4
// waves per SIMD.  A non-zero count = the hardware read the source AFTER the younger VALU instruction wrote it.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_war_probe tools/probes/mfma_war_probe.hip && /tmp/mfma_war_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define NOP0 ""
#define NOP1 "s_nop 0\n"
#define NOP2 "s_nop 1\n"
#define NOP4 "s_nop 3\n"
#define NOP8 "s_nop 7\n"
#define NOP16 "s_nop 15\n"
#define MF0 "v_mfma_f32_32x32x16_f16 a[0:15], v[50:53], v[54:57], 0\n"
#define MFC "v_mfma_f32_32x32x16_f16 a[0:15], v[50:53], v[54:57], a[0:15]\n"
#define CHAIN1 MF0
#define CHAIN2 MF0 MFC
#define CHAIN4 MF0 MFC MFC MFC
#define CLOB_NONE ""
#define CLOB_ALO "v_mov_b32 v50, %[g]\nv_mov_b32 v51, %[g]\n"
#define CLOB_AHI "v_mov_b32 v52, %[g]\nv_mov_b32 v53, %[g]\n"
#define CLOB_BLO "v_mov_b32 v54, %[g]\nv_mov_b32 v55, %[g]\n"
#define CLOB_BHI "v_mov_b32 v56, %[g]\nv_mov_b32 v57, %[g]\n"

#define SEQ(CH, NOPS, CLOB)                                                                                         \
  asm volatile("v_mov_b32 v50, %[a0]\nv_mov_b32 v51, %[a1]\nv_mov_b32 v52, %[a2]\nv_mov_b32 v53, %[a3]\n"      \
               "v_mov_b32 v54, %[b0]\nv_mov_b32 v55, %[b1]\nv_mov_b32 v56, %[b2]\nv_mov_b32 v57, %[b3]\n"      \
               "s_nop 7\n" CH NOPS CLOB "s_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\n"                                 \
               "v_accvgpr_read_b32 %[o0], a0\nv_accvgpr_read_b32 %[o1], a1\nv_accvgpr_read_b32 %[o2], a2\n"        \
               "v_accvgpr_read_b32 %[o3], a3\nv_accvgpr_read_b32 %[o4], a4\nv_accvgpr_read_b32 %[o5], a5\n"        \
               "v_accvgpr_read_b32 %[o6], a6\nv_accvgpr_read_b32 %[o7], a7\nv_accvgpr_read_b32 %[o8], a8\n"        \
               "v_accvgpr_read_b32 %[o9], a9\nv_accvgpr_read_b32 %[o10], a10\nv_accvgpr_read_b32 %[o11], a11\n"    \
               "v_accvgpr_read_b32 %[o12], a12\nv_accvgpr_read_b32 %[o13], a13\nv_accvgpr_read_b32 %[o14], a14\n"  \
               "v_accvgpr_read_b32 %[o15], a15\n"                                                                  \
               : [o0] "=&v"(o[0]), [o1] "=&v"(o[1]), [o2] "=&v"(o[2]), [o3] "=&v"(o[3]), [o4] "=&v"(o[4]),          \
                 [o5] "=&v"(o[5]), [o6] "=&v"(o[6]), [o7] "=&v"(o[7]), [o8] "=&v"(o[8]), [o9] "=&v"(o[9]),          \
                 [o10] "=&v"(o[10]), [o11] "=&v"(o[11]), [o12] "=&v"(o[12]), [o13] "=&v"(o[13]), [o14] "=&v"(o[14]), \
                 [o15] "=&v"(o[15])                                                                                 \
               : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [b0] "v"(b[0]), [b1] "v"(b[1]),    \
                 [b2] "v"(b[2]), [b3] "v"(b[3]), [g] "v"(g)                                                         \
               : "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "a0", "a1", "a2", "a3", "a4", "a5", \
                 "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15")

// bad[0..15]: mismatches per accumulator register; bad[16 + q]: per lane quarter (the lane = output column n % 32, hh)
template <int CHAIN, int DIST, int WHICH>
__global__ void probe(const uint32_t* in, unsigned* bad, int iters) {
  const int l = threadIdx.x & 63;
  uint32_t a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[l * 8 + i]; b[i] = in[l * 8 + 4 + i]; }
  const uint32_t g = 0x46004600u;          // two halves of 6.0: finite, changes every product it enters
  float ref[16], o[16];
  {
    if constexpr (CHAIN == 1) SEQ(CHAIN1, NOP0, CLOB_NONE);
    if constexpr (CHAIN == 2) SEQ(CHAIN2, NOP0, CLOB_NONE);
    if constexpr (CHAIN == 4) SEQ(CHAIN4, NOP0, CLOB_NONE);
    for (int r = 0; r < 16; ++r) ref[r] = o[r];
  }
  unsigned nb[16] = {};
  for (int it = 0; it < iters; ++it) {
#define RUN(CH, NOPS)                                      \
    if constexpr (WHICH == 0) SEQ(CH, NOPS, CLOB_ALO);     \
    if constexpr (WHICH == 1) SEQ(CH, NOPS, CLOB_AHI);     \
    if constexpr (WHICH == 2) SEQ(CH, NOPS, CLOB_BLO);     \
    if constexpr (WHICH == 3) SEQ(CH, NOPS, CLOB_BHI);
#define RUND(CH)                                 \
    if constexpr (DIST == 0) { RUN(CH, NOP0) }   \
    if constexpr (DIST == 1) { RUN(CH, NOP1) }   \
    if constexpr (DIST == 2) { RUN(CH, NOP2) }   \
    if constexpr (DIST == 4) { RUN(CH, NOP4) }   \
    if constexpr (DIST == 8) { RUN(CH, NOP8) }   \
    if constexpr (DIST == 16) { RUN(CH, NOP16) }
    if constexpr (CHAIN == 1) { RUND(CHAIN1) }
    if constexpr (CHAIN == 2) { RUND(CHAIN2) }
    if constexpr (CHAIN == 4) { RUND(CHAIN4) }
    for (int r = 0; r < 16; ++r) nb[r] += (o[r] != ref[r]);
  }
  unsigned tot = 0;
  for (int r = 0; r < 16; ++r) { if (nb[r]) atomicAdd(&bad[r], nb[r]); tot += nb[r]; }
  if (tot) atomicAdd(&bad[16 + (l >> 4)], tot);
}

template <int CHAIN, int DIST, int WHICH>
void run(const uint32_t* din, unsigned* dbad, int threads, int iters) {
  hipMemset(dbad, 0, 20 * sizeof(unsigned));
  hipLaunchKernelGGL((probe<CHAIN, DIST, WHICH>), dim3(256 * 2), dim3(threads), 0, 0, din, dbad, iters);
  unsigned h[20];
  hipMemcpy(h, dbad, sizeof(h), hipMemcpyDeviceToHost);
  unsigned tot = 0;
  for (int r = 0; r < 16; ++r) tot += h[r];
  static const char* names[4] = {"A[0:1]", "A[2:3]", "B[0:1]", "B[2:3]"};
  printf("chain %d dist %2d overwrite %s, %4d threads/block: %10u mismatches of %llu", CHAIN, DIST, names[WHICH], threads, tot,
         (unsigned long long)512 * threads * 16 * iters);
  if (tot) {
    printf("  per register:");
    for (int r = 0; r < 16; ++r) printf(" %u", h[r]);
    printf("  per lane quarter: %u %u %u %u", h[16], h[17], h[18], h[19]);
  }
  printf("\n");
}

template <int CHAIN, int DIST>
void run_which(const uint32_t* din, unsigned* dbad, int threads, int iters) {
  run<CHAIN, DIST, 0>(din, dbad, threads, iters);
  run<CHAIN, DIST, 1>(din, dbad, threads, iters);
  run<CHAIN, DIST, 2>(din, dbad, threads, iters);
  run<CHAIN, DIST, 3>(din, dbad, threads, iters);
}

int main() {
  std::vector<uint32_t> h(64 * 8);
  srand(3);
  auto half_of = [](int v) -> uint32_t {          // small integers as fp16 bit patterns
    _Float16 x = (_Float16)(float)v;
    uint16_t u;
    __builtin_memcpy(&u, &x, 2);
    return u;
  };
  for (auto& w : h) w = half_of(rand() % 7 - 3) | (half_of(rand() % 5 - 2) << 16);
  uint32_t* din; unsigned* dbad;
  hipMalloc(&din, h.size() * 4); hipMalloc(&dbad, 20 * sizeof(unsigned));
  hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int iters = 400;
  for (int threads : {256, 512, 1024}) {          // 1, 2, 4 waves per SIMD
    run_which<1, 0>(din, dbad, threads, iters);
    run_which<2, 0>(din, dbad, threads, iters);
    run_which<4, 0>(din, dbad, threads, iters);
    run_which<2, 1>(din, dbad, threads, iters);
    run_which<2, 2>(din, dbad, threads, iters);
    run_which<2, 4>(din, dbad, threads, iters);
    run_which<2, 8>(din, dbad, threads, iters);
    run_which<4, 4>(din, dbad, threads, iters);
    run_which<4, 8>(din, dbad, threads, iters);
    run_which<4, 16>(din, dbad, threads, iters);
  }
  return 0;
}
