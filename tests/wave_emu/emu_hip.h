// TEST INFRASTRUCTURE ONLY -- a lane-level emulator of the gfx950 execution model for the kernels of librsp_hip.so.
//
// The product library is HIP for gfx950 and nothing else (rsprompter_amd/_lib.py fails loudly without it); this header is
// never seen by the product build.  tests/wave_emu/build.py compiles the UNCHANGED kernel sources for the host against it,
// so that `-m "not gpu"` tests can execute the very code paths the GPU runs -- MFMA fragment layouts, LDS images, the DMA
// (`buffer_load ... lds`) address rule, the transposing LDS read, buffer-resource bounds, wave shuffles, block barriers,
// tile tickets -- on tiny shapes, lane by lane, before a GPU minute is spent.  What it does NOT model: timing, bank
// conflicts, register pressure, the asynchrony of register loads (they complete at issue) and the exact rounding of MFMA accumulation or of v_exp / v_rcp (results agree with the GPU to fp32 round-off, not
// bit for bit).  Asynchrony of the DMA engine is modelled at its two extremes (State::lazy_dma): LDS is written at issue, or
// only when an s_waitcnt vmcnt of the wave forces it.
//
// Execution model: one fiber per work-item (own stack, hand-written x86-64 context switch), blocks run one after the
// other, a wave is 64 consecutive fibers.  Cross-lane operations (MFMA, shuffles, ballot, readfirstlane, ds_read_tr,
// DMA-to-LDS) and barriers are rendezvous points: a lane deposits its operands and yields; the last lane to arrive
// computes the whole wave's result.  Divergence around a cross-lane operation (a partly masked DMA instruction, a shuffle
// inside a branch) is resolved like the hardware does: the pending call site with the lowest address runs first with its
// partial EXEC mask (wave_try_fire).
//
// Semantics taken from: /opt/skills/guides (MFMA 32x32x16 layouts, ds_read_b64_tr_b16), the probes recorded in
// rsprompter_amd/csrc/attn_stream.hip's header (transposing read: inside a 16-lane group lane i supplies the address of
// four halves D_i[0..3], lane l receives D_{4j + l/4}[l % 4]), and the kernels this repository has verified on MI355X.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <type_traits>
#include <vector>
#include <sys/mman.h>

// ------------------------------------------------------------------------------------------------ language keywords
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
// LDS arrays are function-local statics gathered in ONE linker section, so that the poison mode (emu_set_poison_lds) can
// overwrite all of LDS with 0xFF bytes (NaN as fp32 / fp16) at the start of every block: on the device a block inherits whatever
// the previous block on its CU left there, and a kernel that reads LDS it has not written is box- and timing-dependent
#define __shared__ static __attribute__((section("emu_lds")))

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu { unsigned x, y, z; };

typedef void* hipStream_t;
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
constexpr int hipFuncAttributeMaxDynamicSharedMemorySize = 8;
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
#define HIP_SYMBOL(x) (&(x))
template <class T>
inline hipError_t hipGetSymbolAddress(void** out, T* sym) { *out = (void*)sym; return hipSuccess; }

struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) float2 { float x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(8) int2 { int x, y; };
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
inline int2 make_int2(int x, int y) { return int2{x, y}; }

namespace emu {

constexpr int WAVE = 64;
constexpr size_t STACK_BYTES = 256 * 1024;
constexpr int ARG_BYTES = 192, RES_BYTES = 64;

[[noreturn]] inline void die(const char* msg) {
  fprintf(stderr, "wave_emu: %s\n", msg);
  abort();
}

// ---- context switch (callee-saved registers + stack pointer) ----
extern "C" void emu_switch_ctx(void** from_sp, void* to_sp);
#ifdef EMU_IMPLEMENTATION
__asm__(
    ".text\n.globl emu_switch_ctx\n.type emu_switch_ctx,@function\nemu_switch_ctx:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n"
    ".size emu_switch_ctx, .-emu_switch_ctx\n");
#endif

struct Wave;
struct Block;

struct Fiber {
  void* sp = nullptr;
  char* stack = nullptr;
  bool done = false;
  uint3_emu tid{0, 0, 0};
  int lane = 0;
  Wave* wave = nullptr;
  Block* block = nullptr;
  // blocked while *wait_gen == wait_val
  const uint64_t* wait_gen = nullptr;
  uint64_t wait_val = 0;
};

typedef void (*FireFn)(Wave&);

// a rendezvous in progress: the lanes of one wave that wait at one call site
struct Group {
  const void* site;
  int op;
  FireFn fire;
  uint64_t mask;
};

// a DMA-to-LDS instruction whose data has not been written to LDS yet (lazy mode, see dma_commit)
struct DmaOp {
  uint64_t mask;
  unsigned char* base;
  int size;
  unsigned char data[WAVE][16];
};

struct Wave {
  std::deque<DmaOp> dmaq;
  int nlanes = 0;                 // lanes that exist (the last wave of a block may be partial)
  int live = 0;                   // lanes that have not returned
  uint64_t live_mask = 0;
  uint64_t barrier_mask = 0;      // lanes waiting at the block barrier
  uint64_t arrived_mask = 0;      // the participants of the rendezvous being computed (EXEC)
  int arrived = 0;
  std::vector<Group> groups;      // usually one; more when the wave has diverged around a cross-lane operation
  uint64_t release[WAVE];         // bumped when the lane's rendezvous has been computed
  alignas(64) unsigned char args[WAVE][ARG_BYTES];
  alignas(64) unsigned char res[WAVE][RES_BYTES];
};

struct Block {
  std::vector<Fiber> fibers;
  std::vector<Wave> waves;
  int live = 0;
  int bar_arrived = 0;
  uint64_t bar_gen = 0;
};

struct Launch {
  dim3 grid, block;
  uint3_emu block_idx{0, 0, 0};
  std::function<void()> body;
};

struct State {
  Launch L;
  Block* blk = nullptr;
  Fiber* cur = nullptr;
  void* sched_sp = nullptr;
  std::vector<char*> stack_pool;
  uint64_t n_switch = 0, n_collective = 0;
  // 0: a DMA-to-LDS instruction writes LDS when it is issued (the EARLIEST the hardware may do it: exposes a buffer that is
  // refilled while another wave still reads it); 1: it writes LDS only when an `s_waitcnt vmcnt(n)` of the wave demands it
  // (the LATEST the hardware may do it: exposes a missing wait in front of the barrier).  Tests run DMA kernels both ways.
  int lazy_dma = 0;
  int poison_lds = 0;      // 1: every LAUNCH, 2: every BLOCK starts with all of LDS = 0xFF bytes (2 costs an 8 MB memset per block)
};
extern State g;
#ifdef EMU_IMPLEMENTATION
State g;
#endif

inline void dma_apply(const DmaOp& op) {
  for (int l = 0; l < WAVE; ++l)
    if ((op.mask >> l) & 1ull) memcpy(op.base + (size_t)l * op.size, op.data[l], op.size);
}
inline void dma_flush(Wave& w, size_t keep) {
  while (w.dmaq.size() > keep) { dma_apply(w.dmaq.front()); w.dmaq.pop_front(); }
}

inline void yield_to_scheduler() {
  Fiber* f = g.cur;
  ++g.n_switch;
  emu_switch_ctx(&f->sp, g.sched_sp);
}

inline void block_on(const uint64_t* gen, uint64_t val) {
  Fiber* f = g.cur;
  f->wait_gen = gen;
  f->wait_val = val;
  do { yield_to_scheduler(); } while (*gen == val);
  f->wait_gen = nullptr;
}

// ---- wave rendezvous: every participating lane deposits ARG_BYTES and waits; the rendezvous is computed for all of
// them at once when (a) every live lane of the wave waits at this call site, or (b) the wave has DIVERGED around it --
// every live lane is blocked somewhere (other call sites, the block barrier) -- in which case the pending call site with
// the lowest code address runs first with its partial lane mask, which is how the hardware serialises the sides of a
// branch (EXEC mask; e.g. the partly masked last DMA instruction of a tile).
inline void fire_group(Wave& w, size_t gi) {
  const Group grp = w.groups[gi];
  w.groups.erase(w.groups.begin() + gi);
  ++g.n_collective;
  w.arrived_mask = grp.mask;
  w.arrived = __builtin_popcountll(grp.mask);
  grp.fire(w);
  for (int l = 0; l < WAVE; ++l) if ((grp.mask >> l) & 1ull) ++w.release[l];
}

inline void wave_try_fire(Wave& w) {
  for (;;) {
    if (w.groups.empty()) return;
    size_t pick = w.groups.size();
    uint64_t waiting = w.barrier_mask;
    for (size_t i = 0; i < w.groups.size(); ++i) {
      if (w.groups[i].mask == w.live_mask) { pick = i; break; }
      waiting |= w.groups[i].mask;
    }
    if (pick == w.groups.size()) {
      if (waiting != w.live_mask) return;                       // somebody is still running: it may join a group
      pick = 0;
      for (size_t i = 1; i < w.groups.size(); ++i)
        if ((uintptr_t)w.groups[i].site < (uintptr_t)w.groups[pick].site) pick = i;
    }
    fire_group(w, pick);
  }
}

inline void wave_rendezvous(FireFn fire, int op, const void* site) {
  Fiber* f = g.cur;
  Wave& w = *f->wave;
  size_t gi = 0;
  for (; gi < w.groups.size(); ++gi) if (w.groups[gi].site == site && w.groups[gi].op == op) break;
  if (gi == w.groups.size()) w.groups.push_back(Group{site, op, fire, 0});
  w.groups[gi].mask |= 1ull << f->lane;
  const uint64_t rel = w.release[f->lane];
  wave_try_fire(w);
  if (w.release[f->lane] == rel) block_on(&w.release[f->lane], rel);
}

inline void block_barrier() {
  Fiber* f = g.cur;
  Block& b = *f->block;
  Wave& w = *f->wave;
  const uint64_t gen = b.bar_gen;
  ++b.bar_arrived;
  w.barrier_mask |= 1ull << f->lane;
  if (b.bar_arrived == b.live) {
    b.bar_arrived = 0; ++b.bar_gen;
    for (Wave& x : b.waves) x.barrier_mask = 0;
  } else {
    wave_try_fire(w);                                           // lanes of this wave may wait for a diverged rendezvous
    if (b.bar_gen == gen) block_on(&b.bar_gen, gen);
  }
}

// a lane returned from the kernel: rendezvous points it will never reach must not wait for it
inline void lane_exit() {
  Fiber* f = g.cur;
  Wave& w = *f->wave;
  Block& b = *f->block;
  f->done = true;
  --w.live; --b.live;
  w.live_mask &= ~(1ull << f->lane);
  if (w.live == 0) dma_flush(w, 0);                             // the wave ends: its memory operations complete
  if (w.live > 0) wave_try_fire(w);
  if (b.live > 0 && b.bar_arrived == b.live) {
    b.bar_arrived = 0; ++b.bar_gen;
    for (Wave& x : b.waves) x.barrier_mask = 0;
  }
}

void fiber_main();
void run_launch();
void poison_lds_now();
#ifdef EMU_IMPLEMENTATION
void fiber_main() {
  g.L.body();
  lane_exit();
  for (;;) yield_to_scheduler();
}

static char* get_stack() {
  if (!g.stack_pool.empty()) { char* s = g.stack_pool.back(); g.stack_pool.pop_back(); return s; }
  void* p = mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
  if (p == MAP_FAILED) die("mmap of a fiber stack failed");
  return (char*)p;
}

void run_launch() {
  const Launch& L = g.L;
  const int T = (int)(L.block.x * L.block.y * L.block.z);
  if (T <= 0 || T > 1024) die("block size");
  const int nw = (T + WAVE - 1) / WAVE;
  Block blk;
  blk.fibers.resize(T);
  blk.waves.resize(nw);
  for (int i = 0; i < T; ++i) blk.fibers[i].stack = get_stack();
  for (unsigned bz = 0; bz < L.grid.z; ++bz)
    for (unsigned by = 0; by < L.grid.y; ++by)
      for (unsigned bx = 0; bx < L.grid.x; ++bx) {
        g.L.block_idx = uint3_emu{bx, by, bz};
        if (g.poison_lds == 2 || (g.poison_lds == 1 && bx == 0 && by == 0 && bz == 0)) poison_lds_now();
        g.blk = &blk;
        blk.live = T; blk.bar_arrived = 0;
        for (int w = 0; w < nw; ++w) {
          Wave& W = blk.waves[w];
          W.nlanes = std::min(WAVE, T - w * WAVE);
          W.live = W.nlanes; W.arrived = 0; W.arrived_mask = 0; W.barrier_mask = 0; W.groups.clear(); W.dmaq.clear();
          W.live_mask = W.nlanes == 64 ? ~0ull : ((1ull << W.nlanes) - 1);
        }
        for (int i = 0; i < T; ++i) {
          Fiber& f = blk.fibers[i];
          f.done = false; f.wait_gen = nullptr;
          f.tid = uint3_emu{(unsigned)(i % L.block.x), (unsigned)((i / L.block.x) % L.block.y), (unsigned)(i / (L.block.x * L.block.y))};
          f.lane = i % WAVE; f.wave = &blk.waves[i / WAVE]; f.block = &blk;
          // initial frame: six callee-saved registers, then the "return address" = fiber_main, at a 16-byte boundary
          uintptr_t top = ((uintptr_t)f.stack + STACK_BYTES) & ~(uintptr_t)15;
          void** s = (void**)(top - 16);
          s[0] = (void*)&fiber_main;
          s -= 6;
          for (int r = 0; r < 6; ++r) s[r] = nullptr;
          f.sp = (void*)s;
        }
        int remaining = T;
        while (remaining > 0) {
          bool progressed = false;
          for (int i = 0; i < T; ++i) {
            Fiber& f = blk.fibers[i];
            if (f.done) continue;
            if (f.wait_gen && *f.wait_gen == f.wait_val) continue;
            g.cur = &f;
            emu_switch_ctx(&g.sched_sp, f.sp);
            progressed = true;
            if (f.done) --remaining;
          }
          if (!progressed) {
            fprintf(stderr, "wave_emu: deadlock in block (%u,%u,%u): %d work-items blocked; ", bx, by, bz, remaining);
            for (int w = 0; w < nw; ++w) {
              fprintf(stderr, "[wave %d: live %016llx at barrier %016llx", w, (unsigned long long)blk.waves[w].live_mask,
                      (unsigned long long)blk.waves[w].barrier_mask);
              for (const Group& gr : blk.waves[w].groups)
                fprintf(stderr, " | op %d site %p lanes %016llx", gr.op, gr.site, (unsigned long long)gr.mask);
              fprintf(stderr, "] ");
            }
            fprintf(stderr, "barrier arrived %d of %d\n", blk.bar_arrived, blk.live);
            abort();
          }
        }
      }
  for (int i = 0; i < T; ++i) g.stack_pool.push_back(blk.fibers[i].stack);
  g.blk = nullptr; g.cur = nullptr;
}
#endif

alignas(64) extern unsigned char dyn_smem_buf[160 * 1024];
#ifdef EMU_IMPLEMENTATION
alignas(64) unsigned char dyn_smem_buf[160 * 1024];
extern "C" void emu_set_lazy_dma(int on) { g.lazy_dma = on; }
extern "C" void emu_set_poison_lds(int on) { g.poison_lds = on; }
extern "C" unsigned char __start_emu_lds[], __stop_emu_lds[];
void poison_lds_now() {
  memset(__start_emu_lds, 0xFF, (size_t)(__stop_emu_lds - __start_emu_lds));
  memset(dyn_smem_buf, 0xFF, sizeof(dyn_smem_buf));
}
#endif
inline void* dyn_smem() { return dyn_smem_buf; }

template <class F>
inline void launch(dim3 grid, dim3 block, size_t smem, F&& body) {
  if (g.cur) die("nested launch");
  if (smem > sizeof(dyn_smem_buf)) die("dynamic shared memory request above 160 KB");
  g.L.grid = grid; g.L.block = block; g.L.body = std::function<void()>(body);
  run_launch();
}

inline Fiber& cur() { return *g.cur; }

// typed access to this lane's rendezvous slots
template <class T> inline T& arg_at(Wave& w, int lane, int off = 0) { return *reinterpret_cast<T*>(w.args[lane] + off); }
template <class T> inline T& res_at(Wave& w, int lane, int off = 0) { return *reinterpret_cast<T*>(w.res[lane] + off); }
inline bool lane_in(const Wave& w, int l) { return (w.arrived_mask >> l) & 1ull; }

}  // namespace emu

#define threadIdx (emu::cur().tid)
#define blockIdx (emu::g.L.block_idx)
#define blockDim (emu::g.L.block)
#define gridDim (emu::g.L.grid)

#define hipLaunchKernelGGL(kernel, grid, block, smem, stream, ...) \
  emu::launch((grid), (block), (smem), [=]() { kernel(__VA_ARGS__); })

// ------------------------------------------------------------------------------------------------ per-lane helpers
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * b) >> 32); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
using std::isinf;
using std::isnan;
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline float __expf(float x) { return expf(x); }
inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
template <class T> inline T min(T a, T b) { return b < a ? b : a; }
template <class T> inline T max(T a, T b) { return a < b ? b : a; }
inline int min(int a, unsigned b) { return (int)b < a ? (int)b : a; }
inline int64_t min(int64_t a, int b) { return b < a ? b : a; }
inline int64_t min(int a, int64_t b) { return b < a ? b : a; }
inline int64_t max(int64_t a, int b) { return a < b ? b : a; }
inline int64_t max(int a, int64_t b) { return a < b ? b : a; }

inline void __syncthreads() { emu::block_barrier(); }
inline void emu_amdgcn_s_barrier() { emu::block_barrier(); }
inline void emu_amdgcn_sched_barrier(int) {}
// s_waitcnt vmcnt(n): at most n of the wave's vector-memory operations stay outstanding.  Only DMA-to-LDS operations are
// tracked (register loads complete at issue here), i.e. the count is the WEAKEST the hardware could apply: interleaved
// loads / stores only make the real wait stronger (vmcnt retires in order).
// (lazy_dma == 2 ignores the waits -- the emulator's self-test that a kernel WITHOUT its waits is caught in mode 1)
inline void emu_vmcnt_wait(int n) { if (emu::g.lazy_dma != 2) emu::dma_flush(*emu::cur().wave, (size_t)(n < 0 ? 0 : n)); }
// A returning global atomic is a vector-memory operation of the wave: it takes a place in the vmcnt order (the persistent
// GEMM counts its tile-ticket atomic into its waits: "vmcnt(NDMA + 1)").  build.py routes __hip_atomic_fetch_add here.
template <class T, class U>
inline T emu_hip_atomic_fetch_add(T* p, U v, int order, int scope) {
  const T old = __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, 4 /* agent scope */);
  if (emu::g.lazy_dma) {
    emu::DmaOp op;
    op.mask = 0; op.base = nullptr; op.size = 0;
    emu::cur().wave->dmaq.push_back(op);
  }
  return old;
}
// gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14]
inline void emu_amdgcn_s_waitcnt(int imm) { emu_vmcnt_wait((imm & 15) | (((imm >> 14) & 3) << 4)); }
inline void emu_amdgcn_s_setprio(int) {}
#define emu_amdgcn_fence(...) ((void)0)
inline uint64_t emu_amdgcn_s_memtime() { return emu::g.n_switch; }
inline unsigned emu_amdgcn_s_getreg(int) { return 0; }
inline float emu_amdgcn_exp2f(float x) { return exp2f(x); }
inline float emu_amdgcn_rcpf(float x) { return 1.0f / x; }

// atomics (blocks and lanes run one at a time: plain read-modify-write)
template <class T, class U> inline T atomicAdd(T* p, U v) { T o = *p; *p = (T)(o + (T)v); return o; }
template <class T, class U> inline T atomicMax(T* p, U v) { T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <class T, class U> inline T atomicMin(T* p, U v) { T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <class T, class U> inline T atomicOr(T* p, U v) { T o = *p; *p = (T)(o | (T)v); return o; }
// __hip_atomic_fetch_add / _store / _load are clang builtins on the host as well (scoped atomics on host memory)
#ifndef __HIP_MEMORY_SCOPE_AGENT
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#endif

// ------------------------------------------------------------------------------------------------ conversions
typedef _Float16 emu_half2 __attribute__((ext_vector_type(2)));
inline _Float16 emu_f2h_rtz(float x) {
  _Float16 h = (_Float16)x;                                   // round to nearest even
  if (x != x) return h;
  uint16_t b; memcpy(&b, &h, 2);
  const float back = (float)h;
  if (std::isinf(back) && !std::isinf(x)) { b = (uint16_t)((b & 0x8000u) | 0x7bffu); memcpy(&h, &b, 2); return h; }
  if (fabsf(back) > fabsf(x)) { b = (uint16_t)(b - 1); memcpy(&h, &b, 2); }   // one ulp towards zero (sign-magnitude encoding)
  return h;
}
inline emu_half2 emu_amdgcn_cvt_pkrtz(float a, float b) { emu_half2 r; r[0] = emu_f2h_rtz(a); r[1] = emu_f2h_rtz(b); return r; }

// OCP e4m3 (fn: no inf, max 448), round to nearest even, saturating
inline unsigned emu_f2e4m3(float x) {
  if (x != x) return 0x7f;
  const unsigned sign = std::signbit(x) ? 0x80u : 0u;
  float a = fabsf(x);
  if (a >= 448.0f) return sign | 0x7e;
  if (a < 0.0009765625f) return sign;                          // below half the smallest subnormal (2^-10)
  int e; float m = frexpf(a, &e);                               // a = m 2^e, m in [0.5, 1)
  int E = e - 1;                                                // a = (2m) 2^E
  if (E < -6) {                                                 // subnormal: multiples of 2^-9
    const float q = nearbyintf(a * 512.0f);
    return sign | (unsigned)q;                                  // q == 8 carries into the first normal
  }
  float q = nearbyintf((2.0f * m - 1.0f) * 8.0f);
  if (q == 8.0f) { q = 0.0f; ++E; }
  if (E > 8) return sign | 0x7e;
  unsigned code = ((unsigned)(E + 7) << 3) | (unsigned)q;
  if (code > 0x7e) code = 0x7e;
  return sign | code;
}
inline int emu_amdgcn_cvt_pk_fp8_f32(float a, float b, int old, bool hi_word) {
  const unsigned pk = emu_f2e4m3(a) | (emu_f2e4m3(b) << 8);
  return hi_word ? (int)(((unsigned)old & 0x0000ffffu) | (pk << 16)) : (int)(((unsigned)old & 0xffff0000u) | pk);
}

// ------------------------------------------------------------------------------------------------ wave collectives
namespace emu {
enum { OP_SHFL = 1, OP_BALLOT, OP_RFL, OP_MFMA_F16, OP_TR16, OP_DMA, OP_LOCKSTEP };

template <class T>
__attribute__((noinline)) T shfl_generic(T v, int src_lane_rel, int width, int mode) {
  static_assert(sizeof(T) <= 8, "shuffle of at most 8 bytes");
  Fiber& f = cur();
  Wave& w = *f.wave;
  struct A { uint64_t v; int sel, width, mode, size; };
  A a{0, src_lane_rel, width, mode, (int)sizeof(T)};
  memcpy(&a.v, &v, sizeof(T));
  arg_at<A>(w, f.lane) = a;
  wave_rendezvous([](Wave& W) {
    for (int l = 0; l < WAVE; ++l) {
      if (!lane_in(W, l)) continue;
      const A& a = arg_at<A>(W, l);
      const int wd = a.width, base = l / wd * wd, rel = l - base;
      int src;
      if (a.mode == 0) src = base + ((a.sel % wd + wd) % wd);                 // __shfl
      else if (a.mode == 1) src = base + ((rel ^ a.sel) < wd ? (rel ^ a.sel) : rel);   // __shfl_xor
      else if (a.mode == 2) src = rel - a.sel >= 0 ? l - a.sel : l;            // __shfl_up
      else src = rel + a.sel < wd ? l + a.sel : l;                             // __shfl_down
      // an inactive source lane returns the caller's own value (the hardware result is undefined there)
      res_at<uint64_t>(W, l) = lane_in(W, src) ? arg_at<A>(W, src).v : a.v;
    }
  }, OP_SHFL, __builtin_return_address(0));
  T out;
  memcpy(&out, &res_at<uint64_t>(w, f.lane), sizeof(T));
  return out;
}
}  // namespace emu
template <class T> __forceinline__ T __shfl(T v, int src, int width = 64) { return emu::shfl_generic(v, src, width, 0); }
template <class T> __forceinline__ T __shfl_xor(T v, int m, int width = 64) { return emu::shfl_generic(v, m, width, 1); }
template <class T> __forceinline__ T __shfl_up(T v, unsigned d, int width = 64) { return emu::shfl_generic(v, (int)d, width, 2); }
template <class T> __forceinline__ T __shfl_down(T v, unsigned d, int width = 64) { return emu::shfl_generic(v, (int)d, width, 3); }

// RSP_WAVE_LOCKSTEP() of rsp_common.h: lanes run one at a time here, so intra-wave exchanges through LDS need the point
__attribute__((noinline)) inline void emu_wave_lockstep() {
  emu::wave_rendezvous([](emu::Wave&) {}, emu::OP_LOCKSTEP, __builtin_return_address(0));
}
#define RSP_WAVE_LOCKSTEP() emu_wave_lockstep()
// s_wave_barrier: the explicit form of the same point (csrc/samattn.hip uses it with wavefront-scope fences)
__forceinline__ void emu_amdgcn_wave_barrier() { emu_wave_lockstep(); }

__attribute__((noinline)) inline unsigned long long __ballot(int pred) {
  emu::Fiber& f = emu::cur();
  emu::Wave& w = *f.wave;
  emu::arg_at<int>(w, f.lane) = pred ? 1 : 0;
  emu::wave_rendezvous([](emu::Wave& W) {
    uint64_t m = 0;
    for (int l = 0; l < emu::WAVE; ++l) if (emu::lane_in(W, l) && emu::arg_at<int>(W, l)) m |= 1ull << l;
    // (per-lane result slots: another diverged group of the wave may be computed before these lanes run again)
    for (int l = 0; l < emu::WAVE; ++l) if (emu::lane_in(W, l)) emu::res_at<uint64_t>(W, l) = m;
  }, emu::OP_BALLOT, __builtin_return_address(0));
  return emu::res_at<uint64_t>(w, f.lane);
}
__forceinline__ unsigned long long emu_amdgcn_ballot_w64(bool p) { return __ballot(p ? 1 : 0); }

template <class T>
__attribute__((noinline)) T emu_amdgcn_readfirstlane(T v) {
  static_assert(sizeof(T) <= 8, "");
  emu::Fiber& f = emu::cur();
  emu::Wave& w = *f.wave;
  uint64_t raw = 0; memcpy(&raw, &v, sizeof(T));
  emu::arg_at<uint64_t>(w, f.lane) = raw;
  emu::wave_rendezvous([](emu::Wave& W) {
    const int first = __builtin_ctzll(W.arrived_mask);
    const uint64_t v0 = emu::arg_at<uint64_t>(W, first);
    for (int l = 0; l < emu::WAVE; ++l) if (emu::lane_in(W, l)) emu::res_at<uint64_t>(W, l) = v0;
  }, emu::OP_RFL, __builtin_return_address(0));
  T out; memcpy(&out, &emu::res_at<uint64_t>(w, f.lane), sizeof(T));
  return out;
}

// ---- MFMA v_mfma_f32_32x32x16_f16: D[m][n] = C[m][n] + sum_k A[m][k] B[k][n]
// lane l holds A[m = l % 32][k = 8 (l / 32) + j], B[k = 8 (l / 32) + j][n = l % 32], j = 0..7, and
// C / D[m = 8 (r / 4) + 4 (l / 32) + r % 4][n = l % 32] in accumulator register r = 0..15.
typedef _Float16 emu_half8 __attribute__((ext_vector_type(8)));
typedef float emu_f32x16 __attribute__((ext_vector_type(16)));
__attribute__((noinline)) inline emu_f32x16 emu_amdgcn_mfma_f32_32x32x16_f16(emu_half8 a, emu_half8 b, emu_f32x16 c, int, int, int) {
  emu::Fiber& f = emu::cur();
  emu::Wave& w = *f.wave;
  memcpy(w.args[f.lane], &a, 16);
  memcpy(w.args[f.lane] + 16, &b, 16);
  memcpy(w.args[f.lane] + 32, &c, 64);
  emu::wave_rendezvous([](emu::Wave& W) {
    if (W.arrived != emu::WAVE) emu::die("MFMA with inactive lanes");
    static thread_local float A[32][16], B[16][32];
    for (int l = 0; l < 64; ++l) {
      const _Float16* pa = reinterpret_cast<const _Float16*>(W.args[l]);
      const _Float16* pb = reinterpret_cast<const _Float16*>(W.args[l] + 16);
      for (int j = 0; j < 8; ++j) {
        A[l & 31][8 * (l >> 5) + j] = (float)pa[j];
        B[8 * (l >> 5) + j][l & 31] = (float)pb[j];
      }
    }
    for (int l = 0; l < 64; ++l) {
      const float* pc = reinterpret_cast<const float*>(W.args[l] + 32);
      float* pd = reinterpret_cast<float*>(W.res[l]);
      const int n = l & 31;
      for (int r = 0; r < 16; ++r) {
        const int m = 8 * (r >> 2) + 4 * (l >> 5) + (r & 3);
        double s = 0.0;
        for (int k = 0; k < 16; ++k) s += (double)A[m][k] * (double)B[k][n];   // products of halves are exact
        pd[r] = (float)((double)pc[r] + s);
      }
    }
  }, emu::OP_MFMA_F16, __builtin_return_address(0));
  emu_f32x16 d;
  memcpy(&d, w.res[f.lane], 64);
  return d;
}
// ---- v_mfma_f32_16x16x32_f16 (no kernel of the library uses it yet).  The C / D map is the one the CDNA4 guide states
// (col = lane & 15, row = 4 (lane >> 4) + register); the A / B map -- 8 consecutive k per lane, k block = lane >> 4 -- is the
// 32x32x16 rule carried over and NOT yet compared with the device (tools/probes/mfma16_probe.hip, round 5's first GPU job).
// A kernel only depends on it through the pairing of A's and B's k sets per lane group: any k permutation common to both
// operands gives the same sums.
// A[m = l % 16][k = 8 (l / 16) + j], B[k = 8 (l / 16) + j][n = l % 16], C / D[m = 4 (l / 16) + r][n = l % 16], r = 0..3
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));
__attribute__((noinline)) inline emu_f32x4 emu_amdgcn_mfma_f32_16x16x32_f16(emu_half8 a, emu_half8 b, emu_f32x4 c, int, int, int) {
  emu::Fiber& f = emu::cur();
  emu::Wave& w = *f.wave;
  memcpy(w.args[f.lane], &a, 16);
  memcpy(w.args[f.lane] + 16, &b, 16);
  memcpy(w.args[f.lane] + 32, &c, 16);
  emu::wave_rendezvous([](emu::Wave& W) {
    if (W.arrived != emu::WAVE) emu::die("MFMA with inactive lanes");
    static thread_local float A[16][32], B[32][16];
    for (int l = 0; l < 64; ++l) {
      const _Float16* pa = reinterpret_cast<const _Float16*>(W.args[l]);
      const _Float16* pb = reinterpret_cast<const _Float16*>(W.args[l] + 16);
      for (int j = 0; j < 8; ++j) {
        A[l & 15][8 * (l >> 4) + j] = (float)pa[j];
        B[8 * (l >> 4) + j][l & 15] = (float)pb[j];
      }
    }
    for (int l = 0; l < 64; ++l) {
      const float* pc = reinterpret_cast<const float*>(W.args[l] + 32);
      float* pd = reinterpret_cast<float*>(W.res[l]);
      for (int r = 0; r < 4; ++r) {
        const int m = 4 * (l >> 4) + r, n = l & 15;
        double s = 0.0;
        for (int k = 0; k < 32; ++k) s += (double)A[m][k] * (double)B[k][n];
        pd[r] = (float)((double)pc[r] + s);
      }
    }
  }, emu::OP_MFMA_F16, __builtin_return_address(0));
  emu_f32x4 d;
  memcpy(&d, w.res[f.lane], 16);
  return d;
}
// ---- v_mfma_scale_f32_32x32x64_f8f6f4 with both operands OCP e4m3 (cbsz = blgp = 0; the other formats are refused):
// lane l holds A[m = l % 32][k = 32 (l / 32) + j] and B[k = 32 (l / 32) + j][n = l % 32], j = 0..31 (one byte each, 8
// registers), C / D as the 32x32 forms above.  Each lane's 32-element K block carries an E8M0 scale 2^(e - 127): byte
// opsel_a of its scale_a register for A, byte opsel_b of scale_b for B (the use csrc/gemm_dma.hip documents; the order of
// the bytes INSIDE a block cannot matter to a kernel that feeds A and B the same way).  Products are exact in double.
typedef int emu_i32x8 __attribute__((ext_vector_type(8)));
inline float emu_e4m3_to_f(unsigned v) {
  const int e = (v >> 3) & 15, m = v & 7;
  float x;
  if (e == 15 && m == 7) x = __builtin_nanf("");
  else if (e == 0) x = ldexpf((float)m, -9);                  // subnormal: m / 8 * 2^-6
  else x = ldexpf(1.0f + (float)m * 0.125f, e - 7);
  return (v & 0x80) ? -x : x;
}
__attribute__((noinline)) inline emu_f32x16 emu_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(emu_i32x8 a, emu_i32x8 b, emu_f32x16 c, int cbsz,
                                                                                    int blgp, int opsel_a, int scale_a,
                                                                                    int opsel_b, int scale_b) {
  if (cbsz != 0 || blgp != 0) emu::die("mfma_scale f8f6f4: only e4m3 x e4m3 (cbsz = blgp = 0) is emulated");
  emu::Fiber& f = emu::cur();
  emu::Wave& w = *f.wave;
  memcpy(w.args[f.lane], &a, 32);
  memcpy(w.args[f.lane] + 32, &b, 32);
  memcpy(w.args[f.lane] + 64, &c, 64);
  const int ea = (scale_a >> (8 * (opsel_a & 3))) & 255, eb = (scale_b >> (8 * (opsel_b & 3))) & 255;
  memcpy(w.args[f.lane] + 128, &ea, 4);
  memcpy(w.args[f.lane] + 132, &eb, 4);
  emu::wave_rendezvous([](emu::Wave& W) {
    if (W.arrived != emu::WAVE) emu::die("MFMA with inactive lanes");
    static thread_local double A[32][64], B[64][32];
    for (int l = 0; l < 64; ++l) {
      const unsigned char* pa = W.args[l];
      const unsigned char* pb = W.args[l] + 32;
      int ea, eb;
      memcpy(&ea, W.args[l] + 128, 4);
      memcpy(&eb, W.args[l] + 132, 4);
      const double sa = ldexp(1.0, ea - 127), sb = ldexp(1.0, eb - 127);
      for (int j = 0; j < 32; ++j) {
        A[l & 31][32 * (l >> 5) + j] = (double)emu_e4m3_to_f(pa[j]) * sa;
        B[32 * (l >> 5) + j][l & 31] = (double)emu_e4m3_to_f(pb[j]) * sb;
      }
    }
    for (int l = 0; l < 64; ++l) {
      const float* pc = reinterpret_cast<const float*>(W.args[l] + 64);
      float* pd = reinterpret_cast<float*>(W.res[l]);
      const int n = l & 31;
      for (int r = 0; r < 16; ++r) {
        const int m = 8 * (r >> 2) + 4 * (l >> 5) + (r & 3);
        double s = 0.0;
        for (int k = 0; k < 64; ++k) s += A[m][k] * B[k][n];
        pd[r] = (float)((double)pc[r] + s);
      }
    }
  }, emu::OP_MFMA_F16, __builtin_return_address(0));
  emu_f32x16 d;
  memcpy(&d, w.res[f.lane], 64);
  return d;
}

// ---- ds_read_b64_tr_b16: inside a 16-lane group lane i supplies the address of four consecutive halves D_i[0..3];
// lane l of the group receives D_{4 j + l / 4}[l % 4], j = 0..3
typedef short emu_v4s __attribute__((ext_vector_type(4)));
template <class P>
__attribute__((noinline)) emu_v4s emu_amdgcn_ds_read_tr16_b64_v4i16(P ptr) {
  emu::Fiber& f = emu::cur();
  emu::Wave& w = *f.wave;
  const unsigned char* p = (const unsigned char*)ptr;
  if (((uintptr_t)p) & 7) emu::die("ds_read_b64_tr_b16: address not 8-byte aligned");
  memcpy(w.args[f.lane], p, 8);
  emu::wave_rendezvous([](emu::Wave& W) {
    if (W.arrived != emu::WAVE) emu::die("ds_read_tr with inactive lanes");
    for (int l = 0; l < 64; ++l) {
      const int g0 = l & ~15, li = l & 15;
      short* out = reinterpret_cast<short*>(W.res[l]);
      for (int j = 0; j < 4; ++j) out[j] = reinterpret_cast<const short*>(W.args[g0 + 4 * j + li / 4])[li % 4];
    }
  }, emu::OP_TR16, __builtin_return_address(0));
  emu_v4s r;
  memcpy(&r, w.res[f.lane], 8);
  return r;
}

// ---- buffer resources (raw buffers: byte offsets, reads beyond num_records return 0, writes are dropped) ----
struct emu_rsrc { unsigned char* base; unsigned stride; unsigned num; unsigned flags; };
typedef emu_rsrc __amdgpu_buffer_rsrc_t;
template <class T>
inline emu_rsrc emu_amdgcn_make_buffer_rsrc(T* p, short stride, int num, int flags) {
  return emu_rsrc{(unsigned char*)const_cast<typename std::remove_const<T>::type*>(p), (unsigned)stride, (unsigned)num, (unsigned)flags};
}
inline emu_rsrc emu_amdgcn_make_buffer_rsrc(std::nullptr_t, short stride, int num, int flags) {
  return emu_rsrc{nullptr, (unsigned)stride, (unsigned)num, (unsigned)flags};
}
inline void emu_buf_read(const emu_rsrc& r, int64_t off, void* dst, int bytes) {
  for (int d = 0; d < bytes; d += 4) {                           // dword granular bounds check
    unsigned v = 0;
    if (r.base && off + d >= 0 && (uint64_t)(off + d + 4) <= (uint64_t)r.num) memcpy(&v, r.base + off + d, 4);
    memcpy((char*)dst + d, &v, 4);
  }
}
inline void emu_buf_write(const emu_rsrc& r, int64_t off, const void* src, int bytes) {
  for (int d = 0; d < bytes; d += 4)
    if (r.base && off + d >= 0 && (uint64_t)(off + d + 4) <= (uint64_t)r.num) memcpy(r.base + off + d, (const char*)src + d, 4);
}
typedef unsigned emu_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned emu_u32x2 __attribute__((ext_vector_type(2)));
inline emu_u32x4 emu_amdgcn_raw_buffer_load_b128(emu_rsrc r, int voff, int soff, int) {
  emu_u32x4 v; emu_buf_read(r, (int64_t)(unsigned)voff + (unsigned)soff, &v, 16); return v;
}
inline unsigned emu_amdgcn_raw_buffer_load_b32(emu_rsrc r, int voff, int soff, int) {
  unsigned v; emu_buf_read(r, (int64_t)(unsigned)voff + (unsigned)soff, &v, 4); return v;
}
// A buffer STORE takes a place in the wave's vmcnt order (vmcnt retires in order and counts stores): gemm_pp2.hip issues
// epilogue stores between its DMA instructions and counts them in its s_waitcnt immediates.  One marker per wave
// instruction (pushed by the wave's first live lane; lanes run one after the other up to the next rendezvous, so the marker
// sits in front of every DMA instruction that follows it in program order).
namespace emu {
inline void vmem_store_marker() {
  if (!g.lazy_dma) return;
  Fiber& f = cur();
  Wave& w = *f.wave;
  if (f.lane != __builtin_ctzll(w.live_mask)) return;
  DmaOp op;
  op.mask = 0; op.base = nullptr; op.size = 0;
  w.dmaq.push_back(op);
}
}  // namespace emu
// loads that a kernel issues as inline assembly and counts in its own `s_waitcnt vmcnt(n)` (csrc/rsp_common.h
// RSP_GLOBAL_LOAD_B128): here a plain load (complete at issue, like every register load of the emulator) + a place in the
// vmcnt order of the lazy-DMA mode, so that the counted waits of such a kernel mean what they mean on the device
#define RSP_GLOBAL_LOAD_B128(dst, ptr) \
  do { memcpy(&(dst), (const void*)(ptr), 16); emu::vmem_store_marker(); } while (0)
// ... and the LDS-DMA twin: the emulated builtin (declared below)
// (the LDS "address" stays a byte pointer here: rsp_lds_addr_t / rsp_lds_addr)
#define RSP_HAVE_LDS_ADDR 1
typedef unsigned char* rsp_lds_addr_t;
template <class P> inline rsp_lds_addr_t rsp_lds_addr(P p) { return (unsigned char*)(uintptr_t)p; }
#define RSP_GLOBAL_LOAD_LDS_B128(gptr, lds) emu_amdgcn_global_load_lds((gptr), (lds), 16, 0, 0)
#define RSP_BUFFER_LOAD_LDS_B128(rsrc, lds, voff, soff) emu_amdgcn_raw_ptr_buffer_load_lds((rsrc), (lds), 16, (int)(voff), (int)(soff), 0, 0)
inline void emu_amdgcn_raw_buffer_store_b128(emu_u32x4 v, emu_rsrc r, int voff, int soff, int) {
  emu_buf_write(r, (int64_t)(unsigned)voff + (unsigned)soff, &v, 16);
  emu::vmem_store_marker();
}
inline void emu_amdgcn_raw_buffer_store_b64(emu_u32x2 v, emu_rsrc r, int voff, int soff, int) {
  emu_buf_write(r, (int64_t)(unsigned)voff + (unsigned)soff, &v, 8);
  emu::vmem_store_marker();
}

// ---- DMA to LDS: every lane fetches `size` bytes from its own global address; the LDS address is wave-uniform
// (M0 = the first active lane's pointer) + imm + lane * size.  Completes at issue (see the header).
namespace emu {
struct DmaArg { unsigned char data[16]; unsigned char* lds; int size; };
inline void dma_commit(const void* src16, void* lds, int size, const void* site) {
  Fiber& f = cur();
  Wave& w = *f.wave;
  DmaArg a;
  memcpy(a.data, src16, 16);
  a.lds = (unsigned char*)lds; a.size = size;
  arg_at<DmaArg>(w, f.lane) = a;
  wave_rendezvous([](Wave& W) {
    const int first = __builtin_ctzll(W.arrived_mask);
    unsigned char* base = arg_at<DmaArg>(W, first).lds;
    DmaOp op;
    op.mask = W.arrived_mask; op.base = base; op.size = arg_at<DmaArg>(W, first).size;
    for (int l = 0; l < WAVE; ++l) {
      if (!lane_in(W, l)) continue;
      const DmaArg& a = arg_at<DmaArg>(W, l);
      if (a.lds != base) die("DMA to LDS with a lane-varying LDS base (M0 is wave-uniform)");
      memcpy(op.data[l], a.data, 16);
    }
    if (g.lazy_dma) W.dmaq.push_back(op);
    else dma_apply(op);
  }, OP_DMA, site);
}
}  // namespace emu
template <class G, class L>
__attribute__((noinline)) void emu_amdgcn_global_load_lds(G gptr, L lptr, int size, int imm, int) {
  if (size != 4 && size != 12 && size != 16) emu::die("global_load_lds size");
  unsigned char buf[16] = {0};
  memcpy(buf, (const unsigned char*)gptr + imm, size);
  emu::dma_commit(buf, (unsigned char*)lptr + imm, size, __builtin_return_address(0));
}
template <class L>
__attribute__((noinline)) void emu_amdgcn_raw_ptr_buffer_load_lds(emu_rsrc r, L lptr, int size, int voff, int soff, int imm, int) {
  if (size != 4 && size != 12 && size != 16) emu::die("buffer_load_lds size");
  unsigned char buf[16] = {0};
  emu_buf_read(r, (int64_t)(unsigned)voff + (unsigned)soff + imm, buf, size);
  emu::dma_commit(buf, (unsigned char*)lptr + imm, size, __builtin_return_address(0));
}
