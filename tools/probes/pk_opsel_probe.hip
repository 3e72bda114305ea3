// Round 6 probe: does `v_pk_mul_f32 ... op_sel:[0,1]` give a wrong low result when several waves share a SIMD?
//
// Why: the ISA bisect of the failing round-5 sam_upscale2_kernel (tools/probes/up2_isa_bisect.py) says
//   * every wrong output is the exact sum MINUS ONE addend, always channel 8 g + 5: element 1 of a group of four, the one
//     product hipcc's SLP packing forms as      v_pk_mul_f32 v[2:3], v[82:83], v[8:9] op_sel:[0,1]     (low result = src0.lo x
//     src1.HI; elements 0, 2, 3 use op_sel_hi:[1,0] forms), always the LOW result, always lanes 48..63;
//   * it needs two waves per SIMD (96 KB of LDS in the kernel descriptor: 0 failures with the same instruction stream);
//   * an s_nop behind every v_pk_* -- the wave gives up its issue slot there, its neighbour's instruction goes in between --
//     turns 1 wrong launch in 50-700 into EVERY launch wrong (4 500 wrong values per launch).
// This probe issues that instruction form in isolation and inside the kernel's accumulation pattern (v_mov into the source
// pair, v_pk_mul, v_pk_add into a running pair), with and without the s_nops, with 1 / 2 / 3 waves per SIMD, on small
// integers (every product and sum exact in fp32), and counts results that differ from the arithmetic, per lane quarter and
// per result half.  MODE 3 is the control: the same chain with the op_sel_hi:[1,0] form only.
//
// RESULT (profiles/r6_multimask_root_cause/isa_bisect_11_*, isa_bisect_12_*): modes 0-11 -- the chain alone, behind MFMAs, next
// to memory traffic, source pair written by v_mov or by a load -- never fail.  Modes 12, 13, 15 -- an MFMA of the SAME wave
// issued inside every link, an s_nop 0 behind every instruction -- lose the LOW product in lanes 48..63 with two or three waves
// per SIMD (up to 9 % of the sums; 0 with one wave per SIMD), the high product never.  The same links with the op_sel_hi:[1,0]
// form (16), with two v_mul_f32 instead of the packed multiply (17), without the s_nops (14), with the wait state only behind
// the MFMA (18) or only around the multiply (19): 0.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/pk_opsel_probe tools/probes/pk_opsel_probe.hip && /tmp/pk_opsel_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

// one link of the chain: (v60, v61) <- (a, b); product pair v[62:63]; running sums v[56:57]
#define LINK_OPSEL(NOP)                                                                              \
  "v_mov_b32 v60, %[a]\n" NOP "v_mov_b32 v61, %[b]\n" NOP                                            \
  "v_pk_mul_f32 v[62:63], v[60:61], v[52:53] op_sel:[0,1]\n" NOP                                     \
  "v_pk_add_f32 v[56:57], v[56:57], v[62:63]\n" NOP
#define LINK_LOLO(NOP)                                                                               \
  "v_mov_b32 v60, %[a]\n" NOP "v_mov_b32 v61, %[b]\n" NOP                                            \
  "v_pk_mul_f32 v[62:63], v[60:61], v[52:53] op_sel_hi:[1,0]\n" NOP                                  \
  "v_pk_add_f32 v[56:57], v[56:57], v[62:63]\n" NOP
// the in-place form of the kernel:  v_pk_mul_f32 v[8:9], v[32:33], v[8:9] op_sel:[0,1]  (destination = second source)
#define LINK_INPLACE(NOP)                                                                            \
  "v_mov_b32 v60, %[a]\n" NOP "v_mov_b32 v61, %[b]\n" NOP                                            \
  "v_mov_b32 v64, v52\nv_mov_b32 v65, v53\n" NOP                                                      \
  "v_pk_mul_f32 v[64:65], v[60:61], v[64:65] op_sel:[0,1]\n" NOP                                     \
  "v_pk_add_f32 v[56:57], v[56:57], v[64:65]\n" NOP
// ... with memory traffic landing in the register file meanwhile: a 16-byte global load and an LDS read issued in front of
// every link (their destinations v[74:81] are not otherwise used), all waited for at the end
#define LINK_MEM(NOP)                                                                                \
  "global_load_dwordx4 v[74:77], %[p], off\nds_read_b128 v[78:81], %[q]\n"                           \
  "v_mov_b32 v60, %[a]\n" NOP "v_mov_b32 v61, %[b]\n" NOP                                            \
  "v_pk_mul_f32 v[62:63], v[60:61], v[52:53] op_sel:[0,1]\n" NOP                                     \
  "v_pk_add_f32 v[56:57], v[56:57], v[62:63]\n" NOP
// ... with a matrix instruction issued inside EVERY link (the kernel's scheduler interleaves the next sub-pixel's MFMAs with this
// sub-pixel's sums; without its MFMAs the stressed kernel does not fail: up2_isa_bisect.py nomfma+pk, 0 of 300)
#define LINK_MFMA_A(NOP) "v_mfma_f32_32x32x16_f16 a[0:15], v[66:69], v[70:73], a[0:15]\n" NOP LINK_OPSEL(NOP)
#define LINK_MFMA_V(NOP) "v_mfma_f32_32x32x16_f16 v[100:115], v[66:69], v[70:73], v[100:115]\n" NOP LINK_OPSEL(NOP)
// minimal-set variants of the reproducing modes 12 / 13: the control form, the two-v_mul_f32 workaround, the wait state only
// behind the MFMA
#define LINK_MFMA_V_LOLO(NOP) "v_mfma_f32_32x32x16_f16 v[100:115], v[66:69], v[70:73], v[100:115]\n" NOP LINK_LOLO(NOP)
#define LINK_MFMA_V_MUL(NOP)                                                                          \
  "v_mfma_f32_32x32x16_f16 v[100:115], v[66:69], v[70:73], v[100:115]\n" NOP                          \
  "v_mov_b32 v60, %[a]\n" NOP "v_mov_b32 v61, %[b]\n" NOP                                            \
  "v_mul_f32 v62, v60, v53\n" NOP "v_mul_f32 v63, v61, v53\n" NOP                                    \
  "v_pk_add_f32 v[56:57], v[56:57], v[62:63]\n" NOP
#define LINK_MFMA_V_NOP1 "v_mfma_f32_32x32x16_f16 v[100:115], v[66:69], v[70:73], v[100:115]\ns_nop 0\n" LINK_OPSEL("")
#define LINK_MFMA_V_NOPSEL                                                                            \
  "v_mfma_f32_32x32x16_f16 v[100:115], v[66:69], v[70:73], v[100:115]\n"                              \
  "v_mov_b32 v60, %[a]\nv_mov_b32 v61, %[b]\ns_nop 0\n"                                              \
  "v_pk_mul_f32 v[62:63], v[60:61], v[52:53] op_sel:[0,1]\ns_nop 0\n"                                \
  "v_pk_add_f32 v[56:57], v[56:57], v[62:63]\n"
#define ZERO_V100 "v_mov_b32 v100, 0\nv_mov_b32 v101, 0\nv_mov_b32 v102, 0\nv_mov_b32 v103, 0\nv_mov_b32 v104, 0\nv_mov_b32 v105, 0\nv_mov_b32 v106, 0\nv_mov_b32 v107, 0\nv_mov_b32 v108, 0\nv_mov_b32 v109, 0\nv_mov_b32 v110, 0\nv_mov_b32 v111, 0\nv_mov_b32 v112, 0\nv_mov_b32 v113, 0\nv_mov_b32 v114, 0\nv_mov_b32 v115, 0\n"
#define MFMAS "v_mfma_f32_32x32x16_f16 a[0:15], v[66:69], v[70:73], 0\nv_mfma_f32_32x32x16_f16 a[0:15], v[66:69], v[70:73], a[0:15]\n"

#define BODY(PRE, L)                                                                                                  \
  asm volatile("v_mov_b32 v52, %[h0]\nv_mov_b32 v53, %[h1]\nv_mov_b32 v56, 0\nv_mov_b32 v57, 0\n"                     \
               "v_mov_b32 v66, 0\nv_mov_b32 v67, 0\nv_mov_b32 v68, 0\nv_mov_b32 v69, 0\n"                             \
               "v_mov_b32 v70, 0\nv_mov_b32 v71, 0\nv_mov_b32 v72, 0\nv_mov_b32 v73, 0\n" PRE L L L L L L L L          \
               "s_waitcnt vmcnt(0) lgkmcnt(0)\ns_nop 7\nv_mov_b32 %[o0], v56\nv_mov_b32 %[o1], v57\n"                      \
               : [o0] "=&v"(o0), [o1] "=&v"(o1)                                                                        \
               : [a] "v"(a), [b] "v"(b), [h0] "v"(h0), [h1] "v"(h1), [p] "v"(gp), [q] "v"(lp)                          \
               : "memory", "v52", "v53", "v56", "v57", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69",   \
                 "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81",                                             \
                 "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", \
                 "v114", "v115",                                                                                     \
                 "v70", "v71", "v72", "v73", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10",     \
                 "a11", "a12", "a13", "a14", "a15")

// the same chain, but the source pair v[52:53] is written by a GLOBAL LOAD (as the hyper values of the kernel are) instead of v_mov:
// in the kernel, copying those two registers onto themselves with v_mov right in front of the multiply cut the failures 1500-fold
#define BODY_VMEM(PRE, L)                                                                                             \
  asm volatile("global_load_dwordx4 v[52:55], %[p], off\nv_mov_b32 v56, 0\nv_mov_b32 v57, 0\n"                         \
               "v_mov_b32 v66, 0\nv_mov_b32 v67, 0\nv_mov_b32 v68, 0\nv_mov_b32 v69, 0\n"                             \
               "v_mov_b32 v70, 0\nv_mov_b32 v71, 0\nv_mov_b32 v72, 0\nv_mov_b32 v73, 0\ns_waitcnt vmcnt(0)\n" PRE L L L L L L L L \
               "s_waitcnt vmcnt(0) lgkmcnt(0)\ns_nop 7\nv_mov_b32 %[o0], v56\nv_mov_b32 %[o1], v57\n"                      \
               : [o0] "=&v"(o0), [o1] "=&v"(o1)                                                                        \
               : [a] "v"(a), [b] "v"(b), [p] "v"(gp), [q] "v"(lp)                                                      \
               : "memory", "v52", "v53", "v54", "v55", "v56", "v57", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", \
                 "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81",   \
                 "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", \
                 "v114", "v115",                                                                                     \
                 "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15")

// bad[0 + q]: wrong LOW sums per lane quarter, bad[4 + q]: wrong HIGH sums per lane quarter
template <int MODE>
__global__ void probe(const float* in, unsigned* bad, int iters) {
  const int l = threadIdx.x & 63;
  __shared__ float sm[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) sm[i] = in[i];
  __syncthreads();
  const float* gp = in + ((threadIdx.x * 4 + blockIdx.x * 64) & 1020);
  const unsigned lp = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)(sm + ((threadIdx.x * 4) & 1020));
  unsigned nlo = 0, nhi = 0;
  for (int it = 0; it < iters; ++it) {
    const float a = in[(l * 4 + it) & 1023], b = in[(l * 4 + it + 1) & 1023];
    const float h0 = in[(l + it * 3 + 2) & 1023], h1 = in[(l + it * 5 + 7) & 1023];
    float o0, o1;
    if constexpr (MODE == 0) BODY("", LINK_OPSEL(""));                    // the kernel's pattern, back to back
    if constexpr (MODE == 1) BODY("", LINK_OPSEL("s_nop 0\n"));           // ... with the issue slot given up behind every instruction
    if constexpr (MODE == 2) BODY(MFMAS, LINK_OPSEL("s_nop 0\n"));        // ... behind two MFMAs (the neighbour's pk code meets them)
    if constexpr (MODE == 3) BODY("", LINK_LOLO("s_nop 0\n"));            // control: op_sel_hi:[1,0] only
    if constexpr (MODE == 4) BODY("", LINK_INPLACE("s_nop 0\n"));         // destination = second source
    if constexpr (MODE == 5) BODY(MFMAS, LINK_INPLACE("s_nop 0\n"));
    if constexpr (MODE == 6) BODY("", LINK_MEM("s_nop 0\n"));             // loads and LDS reads returning meanwhile
    if constexpr (MODE == 7) BODY(MFMAS, LINK_MEM("s_nop 0\n"));
    if constexpr (MODE == 8) BODY_VMEM("", LINK_OPSEL("s_nop 0\n"));      // source pair written by a global load
    if constexpr (MODE == 9) BODY_VMEM(MFMAS, LINK_OPSEL("s_nop 0\n"));
    if constexpr (MODE == 10) BODY_VMEM("", LINK_MEM("s_nop 0\n"));       // ... with more loads / LDS reads in flight
    if constexpr (MODE == 11) BODY_VMEM(MFMAS, LINK_INPLACE("s_nop 0\n"));
    if constexpr (MODE == 12) BODY_VMEM(MFMAS, LINK_MFMA_A("s_nop 0\n"));   // an MFMA into AGPRs in every link, source pair from a load
    if constexpr (MODE == 13) BODY_VMEM(ZERO_V100, LINK_MFMA_V("s_nop 0\n"));
    if constexpr (MODE == 14) BODY_VMEM(MFMAS, LINK_MFMA_A(""));             // ... without the s_nops
    if constexpr (MODE == 15) BODY(ZERO_V100, LINK_MFMA_V("s_nop 0\n"));         // as 13, the source pair written by v_mov
    if constexpr (MODE == 16) BODY_VMEM(ZERO_V100, LINK_MFMA_V_LOLO("s_nop 0\n")); // as 13, control form op_sel_hi:[1,0] (expects x src1.lo)
    if constexpr (MODE == 17) BODY_VMEM(ZERO_V100, LINK_MFMA_V_MUL("s_nop 0\n")); // as 13, two v_mul_f32 instead of the packed multiply
    if constexpr (MODE == 18) BODY_VMEM(ZERO_V100, LINK_MFMA_V_NOP1);              // as 13, ONE wait state, right behind the MFMA
    if constexpr (MODE == 19) BODY_VMEM(ZERO_V100, LINK_MFMA_V_NOPSEL);            // as 13, wait states only around the packed multiply
    const float hs = (MODE >= 8 && MODE != 15) ? gp[1] : h1;                            // modes 8+: v53 = the second float of the loaded 16 bytes
    const float hl = (MODE == 16) ? gp[0] : h0;                           // the controls multiply with src1.lo
    const float w0 = (MODE == 3 || MODE == 16) ? 8.f * a * hl : 8.f * a * hs;   // low sums: a x src1.HI
    const float w1 = (MODE == 3 || MODE == 16) ? 8.f * b * hl : 8.f * b * hs;   // high sums: b x src1.HI
    nlo += (o0 != w0);
    nhi += (o1 != w1);
  }
  if (nlo) atomicAdd(&bad[l >> 4], nlo);
  if (nhi) atomicAdd(&bad[4 + (l >> 4)], nhi);
}

template <int MODE>
void run(const float* din, unsigned* dbad, int threads, int blocks, int iters, const char* what) {
  hipMemset(dbad, 0, 8 * sizeof(unsigned));
  hipLaunchKernelGGL((probe<MODE>), dim3(blocks), dim3(threads), 0, 0, din, dbad, iters);
  const hipError_t e1 = hipGetLastError(), e2 = hipDeviceSynchronize();
  if (e1 != hipSuccess || e2 != hipSuccess) printf("LAUNCH FAILED: %s / %s\n", hipGetErrorString(e1), hipGetErrorString(e2));
  unsigned h[8];
  hipMemcpy(h, dbad, sizeof(h), hipMemcpyDeviceToHost);
  printf("mode %d (%s), %4d threads x %4d blocks: low sums wrong per lane quarter %u %u %u %u, high sums %u %u %u %u  (of %llu sums each)\n",
         MODE, what, threads, blocks, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], (unsigned long long)threads / 4 * blocks * iters);
}

int main() {
  std::vector<float> h(1024);
  srand(5);
  for (auto& w : h) w = (float)(rand() % 13 - 6);           // small integers: 8 products of two of them sum exactly
  float* din; unsigned* dbad;
  hipMalloc(&din, h.size() * 4); hipMalloc(&dbad, 8 * sizeof(unsigned));
  hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int iters = 20000;
  struct { int threads, blocks; } shapes[] = {{256, 256}, {256, 512}, {512, 256}, {256, 2048}};   // 1, 2, 2 waves per SIMD; many blocks (1024 threads do not fit the registers of the MFMA modes)
  for (auto s : shapes) {
    run<0>(din, dbad, s.threads, s.blocks, iters, "op_sel:[0,1] chain, back to back");
    run<1>(din, dbad, s.threads, s.blocks, iters, "op_sel:[0,1] chain, s_nop behind every instruction");
    run<2>(din, dbad, s.threads, s.blocks, iters, "... behind two MFMAs");
    run<3>(din, dbad, s.threads, s.blocks, iters, "control: op_sel_hi:[1,0] chain, s_nop behind every instruction");
    run<4>(din, dbad, s.threads, s.blocks, iters, "in-place form (dst = src1), s_nop behind every instruction");
    run<5>(din, dbad, s.threads, s.blocks, iters, "in-place form behind two MFMAs");
    run<6>(din, dbad, s.threads, s.blocks, iters, "op_sel:[0,1] chain with a global load and an LDS read in flight per link");
    run<7>(din, dbad, s.threads, s.blocks, iters, "... behind two MFMAs");
    run<8>(din, dbad, s.threads, s.blocks, iters, "source pair written by a global load, s_nop behind every instruction");
    run<9>(din, dbad, s.threads, s.blocks, iters, "... behind two MFMAs");
    run<10>(din, dbad, s.threads, s.blocks, iters, "... with a global load and an LDS read in flight per link");
    run<11>(din, dbad, s.threads, s.blocks, iters, "... in-place form behind two MFMAs");
    run<12>(din, dbad, s.threads, s.blocks, iters, "load-written source pair, an MFMA into AGPRs inside every link");
    run<13>(din, dbad, s.threads, s.blocks, iters, "... an MFMA into VGPRs inside every link");
    run<14>(din, dbad, s.threads, s.blocks, iters, "... AGPR form without the s_nops");
    run<15>(din, dbad, s.threads, s.blocks, iters, "as 13, the source pair written by v_mov instead of a load");
    run<16>(din, dbad, s.threads, s.blocks, iters, "as 13, control: op_sel_hi:[1,0] form");
    run<17>(din, dbad, s.threads, s.blocks, iters, "as 13, two v_mul_f32 instead of the packed multiply");
    run<18>(din, dbad, s.threads, s.blocks, iters, "as 13, one wait state only, right behind the MFMA");
    run<19>(din, dbad, s.threads, s.blocks, iters, "as 13, wait states only around the packed multiply");
  }
  hipFree(din); hipFree(dbad);
  return 0;
}
