// tests/emu/zh_platform.h — TEST INFRASTRUCTURE ONLY.
//
// Lock-step CPU emulator of the small slice of HIP that zultra_amd/csrc uses, so that the *product's own
// kernel sources* can be executed (slowly) on a machine without a GPU and diffed against the oracle in the
// `-m "not gpu"` test-suite. It shadows csrc/zh_platform.h purely through include-path order
// (tests/emu/build_emu.py); the product library is never built against it and no product code path can
// reach it.
//
// Model: one OS thread; every GPU thread of the running workgroup is a ucontext fiber; workgroups run one
// after another. A fiber runs until it reaches a collective (wave primitive or __syncthreads), where it
// parks until all live fibers of its wave / workgroup have arrived — i.e. collectives must be reached by
// all lanes of the group (the same discipline the real kernels keep on the GPU).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <algorithm>
#include <functional>
#include <vector>

#define ZH_EMU 1
#define ZH_WAVE 64

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

using std::max;
using std::min;

struct dim3 {
   unsigned x, y, z;
   dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};

namespace zh_emu {
struct Fiber {
   ucontext_t ctx;
   char *stack = nullptr;
   bool done = false;
   int waiting = 0;   // 0 = runnable, 1 = wave collective, 2 = workgroup barrier
};
inline std::vector<Fiber> g_fibers;
inline ucontext_t g_sched;
inline int g_cur = -1;
inline std::function<void()> g_body;
inline uint64_t g_slot[1024];
inline const size_t kStack = 256 * 1024;
}   // namespace zh_emu

inline dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace zh_emu {

inline void park(int kind) {
   Fiber &f = g_fibers[g_cur];
   f.waiting = kind;
   swapcontext(&f.ctx, &g_sched);
}

inline void trampoline() {
   g_body();
   g_fibers[g_cur].done = true;
   swapcontext(&g_fibers[g_cur].ctx, &g_sched);
}

// release a group if every live member is parked on the same kind
inline bool try_release(int lo, int hi, int kind) {
   bool any = false;
   for (int i = lo; i < hi; i++) {
      if (g_fibers[i].done) continue;
      if (g_fibers[i].waiting != kind) return false;
      any = true;
   }
   if (!any) return false;
   for (int i = lo; i < hi; i++) g_fibers[i].waiting = 0;
   return true;
}

inline void run_block(unsigned nthreads) {
   if (g_fibers.size() < nthreads) g_fibers.resize(nthreads);
   for (unsigned t = 0; t < nthreads; t++) {
      Fiber &f = g_fibers[t];
      if (!f.stack) f.stack = (char *)malloc(kStack);
      f.done = false;
      f.waiting = 0;
      getcontext(&f.ctx);
      f.ctx.uc_stack.ss_sp = f.stack;
      f.ctx.uc_stack.ss_size = kStack;
      f.ctx.uc_link = &g_sched;
      makecontext(&f.ctx, (void (*)())trampoline, 0);
   }
   for (;;) {
      bool progressed = false, alive = false;
      for (unsigned t = 0; t < nthreads; t++) {
         Fiber &f = g_fibers[t];
         if (f.done) continue;
         alive = true;
         if (f.waiting) continue;
         g_cur = (int)t;
         threadIdx = dim3(t, 0, 0);
         swapcontext(&g_sched, &f.ctx);
         progressed = true;
      }
      if (!alive) break;
      bool released = false;
      for (unsigned w = 0; w * 64 < nthreads; w++)
         released |= try_release((int)(w * 64), (int)std::min(nthreads, (w + 1) * 64), 1);
      released |= try_release(0, (int)nthreads, 2);
      if (!progressed && !released) {
         fprintf(stderr, "zh_emu: deadlock (divergent collective?) in block %u\n", blockIdx.x);
         abort();
      }
   }
}

template <typename... KArgs, typename... Args>
inline void launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, Args... args) {
   gridDim = grid;
   blockDim = block;
   for (unsigned b = 0; b < grid.x; b++) {
      blockIdx = dim3(b, 0, 0);
      g_body = [=]() { kernel(args...); };
      run_block(block.x);
   }
}

inline int wave_base() { return g_cur & ~63; }
inline int wave_end() { return std::min((int)blockDim.x, wave_base() + 64); }

template <typename F>
inline uint64_t collect(uint64_t mine, F combine) {
   g_slot[g_cur] = mine;
   park(1);
   uint64_t r = combine();
   park(1);
   return r;
}
}   // namespace zh_emu

#define ZH_LAUNCH(kernel, grid, block, stream, ...) zh_emu::launch(kernel, dim3(grid), dim3(block), __VA_ARGS__)
#define ZH_LAUNCH_LDS(kernel, grid, block, lds, stream, ...) zh_emu::launch(kernel, dim3(grid), dim3(block), __VA_ARGS__)
#define ZH_DYN_LDS(name) static uint32_t name[160 * 1024 / 4]


inline void __syncthreads() { zh_emu::park(2); }
inline void zh_sync() { zh_emu::park(2); }
inline void zh_sync_lds() { zh_emu::park(2); }
// caller-tracked global loads (zh_platform.h of the product): here simply loads
struct uint4;
struct zh_u32x4_t {
   uint32_t x, y, z, w;
};
struct zh_async_row_t {
   zh_u32x4_t a, b;
   uint32_t byte;
};
#define ZH_ASYNC_ROW_LOADS 3
template <int N>
inline void zh_async_wait() {}
inline void zh_async_landed(zh_async_row_t &) {}
struct zh_async_tile_t {
   uint32_t b, y;
};
#define ZH_ASYNC_TILE_LOADS 2
inline void zh_async_load_tile(zh_async_tile_t &t, const uint32_t *pb, const uint8_t *py) {
   t.b = *pb;
   t.y = *py;
}
inline void zh_async_landed(zh_async_tile_t &) {}
inline void zh_wave_sync() {
   zh_emu::collect(0, [] { return (uint64_t)0; });
}

inline void zh_lockstep_sync() {
   zh_emu::collect(0, [] { return (uint64_t)0; });
}

inline uint64_t zh_ballot(bool p);
inline unsigned zh_lane();
inline uint64_t zh_peers8(uint32_t d, bool valid) {
   uint64_t peers = zh_ballot(valid);
   for (int bit = 0; bit < 8; bit++) {
      const bool one = (d >> bit) & 1u;
      const uint64_t m = zh_ballot(valid && one);
      peers &= one ? m : ~m;
   }
   return peers;
}
inline uint32_t zh_rank_below(uint64_t m) { return (uint32_t)__builtin_popcountll(m & ((1ull << zh_lane()) - 1ull)); }
inline void zh_lockstep_point() {
   zh_emu::collect(0, [] { return (uint64_t)0; });
}

inline unsigned zh_lane() { return (unsigned)zh_emu::g_cur & 63u; }
inline uint64_t zh_ballot(bool p) {
   using namespace zh_emu;
   return collect(p ? 1 : 0, [] {
      uint64_t m = 0;
      for (int i = wave_base(); i < wave_end(); i++)
         if (!g_fibers[i].done && g_slot[i]) m |= 1ull << (i & 63);
      return m;
   });
}
inline uint32_t zh_shfl(uint32_t v, int src) {
   using namespace zh_emu;
   return (uint32_t)collect(v, [src] { return g_slot[wave_base() + (src & 63)]; });
}
inline uint32_t zh_readlane(uint32_t v, int lane) { return zh_shfl(v, lane); }
inline uint32_t zh_readfirstlane(uint32_t v) {
   using namespace zh_emu;
   return (uint32_t)collect(v, [] {
      for (int i = wave_base(); i < wave_end(); i++)
         if (!g_fibers[i].done) return g_slot[i];
      return (uint64_t)0;
   });
}
inline int zh_popc64(uint64_t m) { return __builtin_popcountll(m); }
inline int zh_ctz64(uint64_t m) { return m ? __builtin_ctzll(m) : -1; }
inline int zh_clz32(uint32_t v) { return v ? __builtin_clz(v) : 32; }

template <int N>
inline uint32_t zh_row_shr(uint32_t v) {
   using namespace zh_emu;
   return (uint32_t)collect(v, [] {
      int l = g_cur & 15;
      return l >= N ? g_slot[g_cur - N] : g_slot[g_cur];
   });
}
template <int N>
inline uint32_t zh_row_shl(uint32_t v) {
   using namespace zh_emu;
   return (uint32_t)collect(v, [] {
      int l = g_cur & 15;
      return l + N < 16 ? g_slot[g_cur + N] : g_slot[g_cur];
   });
}
inline uint32_t zh_wave_shr1(uint32_t v, uint32_t feed) {
   using namespace zh_emu;
   return (uint32_t)collect(v, [feed] { return (g_cur & 63) ? g_slot[g_cur - 1] : (uint64_t)feed; });
}
inline uint32_t zh_row_min(uint32_t v) {
   using namespace zh_emu;
   return (uint32_t)collect(v, [] {
      int lo = g_cur & ~15;
      uint64_t m = ~0ull;
      for (int i = lo; i < lo + 16 && i < (int)blockDim.x; i++) m = std::min(m, g_slot[i]);
      return m;
   });
}
inline uint32_t zh_quad_min(uint32_t v) {
   using namespace zh_emu;
   return (uint32_t)collect(v, [] {
      int lo = g_cur & ~3;
      uint64_t m = ~0ull;
      for (int i = lo; i < lo + 4 && i < (int)blockDim.x; i++) m = std::min(m, g_slot[i]);
      return m;
   });
}
inline uint32_t zh_quad_shr1(uint32_t v) {
   using namespace zh_emu;
   return (uint32_t)collect(v, [] { return (g_cur & 3) ? g_slot[g_cur - 1] : g_slot[g_cur]; });
}
inline uint32_t zh_quad_lo2(uint32_t v) {
   using namespace zh_emu;
   return (uint32_t)collect(v, [] { return (g_cur & 2) ? g_slot[g_cur - 2] : g_slot[g_cur]; });
}
inline uint32_t zh_wave_min(uint32_t v) {
   using namespace zh_emu;
   return (uint32_t)collect(v, [] {
      uint64_t m = ~0ull;
      for (int i = wave_base(); i < wave_end(); i++) m = std::min(m, g_slot[i]);
      return m;
   });
}
inline uint32_t zh_wave_min_bcast(uint32_t v) { return zh_wave_min(v); }
inline void zh_wave_min3_lane63(uint32_t &a, uint32_t &b, uint32_t &c) {   // results in every lane here: a superset of "in lane 63"
   a = zh_wave_min(a);
   b = zh_wave_min(b);
   c = zh_wave_min(c);
}
inline uint32_t zh_wave_sum(uint32_t v) {
   using namespace zh_emu;
   return (uint32_t)collect(v, [] {
      uint64_t s = 0;
      for (int i = wave_base(); i < wave_end(); i++) s += g_slot[i];
      return s;
   });
}
inline uint32_t zh_wave_incl_max(uint32_t v) {
   using namespace zh_emu;
   return (uint32_t)collect(v, [] {
      uint64_t m = 0;
      for (int i = wave_base(); i <= g_cur; i++) m = std::max(m, g_slot[i]);
      return m;
   });
}
inline uint32_t zh_wave_excl_sum(uint32_t v) {
   using namespace zh_emu;
   return (uint32_t)collect(v, [] {
      uint64_t s = 0;
      for (int i = wave_base(); i < g_cur; i++) s += g_slot[i];
      return s;
   });
}

inline uint32_t atomicAdd(uint32_t *p, uint32_t v) {
   uint32_t o = *p;
   *p = o + v;
   return o;
}
inline int atomicAdd(int *p, int v) {
   int o = *p;
   *p = o + v;
   return o;
}
inline uint32_t atomicMax(uint32_t *p, uint32_t v) {
   uint32_t o = *p;
   if (v > o) *p = v;
   return o;
}
inline uint32_t atomicSub(uint32_t *p, uint32_t v) {
   uint32_t o = *p;
   *p = o - v;
   return o;
}
inline uint32_t atomicCAS(uint32_t *p, uint32_t cmp, uint32_t val) {
   uint32_t o = *p;
   if (o == cmp) *p = val;
   return o;
}
inline void zh_stores_done() {}
inline uint32_t atomicMin(uint32_t *p, uint32_t v) {
   uint32_t o = *p;
   if (v < o) *p = v;
   return o;
}
inline uint32_t atomicOr(uint32_t *p, uint32_t v) {
   uint32_t o = *p;
   *p = o | v;
   return o;
}
inline uint64_t zh_clock() { return 0; }
inline uint64_t zh_wall_clock() { return 0; }
inline uint32_t zh_load_agent_u32(const uint32_t *p) { return *p; }
inline uint32_t zh_load_agent_u16(const uint16_t *p) { return *p; }
inline uint32_t zh_funnel(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (sh & 31u)); }
typedef uint32_t __attribute__((aligned(1))) zh_u32_any_t;
struct __attribute__((packed, aligned(1))) zh_u128_any_t {
   uint32_t x, y, z, w;
};
inline uint32_t zh_load32_any(const void *p) { return *(const zh_u32_any_t *)p; }
inline zh_u128_any_t zh_load128_any(const void *p) { return *(const zh_u128_any_t *)p; }
inline void zh_set_wave_priority_high() {}
inline void zh_set_wave_priority_mid() {}
inline void zh_set_wave_priority_normal() {}
inline void __threadfence_block() {}
inline void __threadfence() {}
struct uint2 {
   uint32_t x, y;
};
struct uint4 {
   uint32_t x, y, z, w;
};
inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
inline void zh_async_load_row(zh_async_row_t &r, const uint4 *lo, const uint4 *hi, const uint8_t *byte) {
   r.a = {lo->x, lo->y, lo->z, lo->w};
   r.b = {hi->x, hi->y, hi->z, hi->w};
   r.byte = *byte;
}
inline uint32_t zh_load_relaxed(const uint32_t *p) { return *(const volatile uint32_t *)p; }
inline uint32_t zh_atomic_add_lds(uint32_t *p, uint32_t v) { return atomicAdd(p, v); }
inline uint32_t zh_atomic_add_global(uint32_t *p, uint32_t v) { return atomicAdd(p, v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __popc(uint32_t v) { return __builtin_popcount(v); }
inline int __popcll(uint64_t v) { return __builtin_popcountll(v); }

// ---- the sliver of the HIP runtime API the host layer uses ---------------------------------------------
typedef int hipError_t;
typedef void *hipStream_t;
typedef struct zh_emu_event {
   double t;
} *hipEvent_t;
#define hipSuccess 0
#define hipMemcpyHostToDevice 1
#define hipMemcpyDeviceToHost 2
#define hipMemcpyDeviceToDevice 3
#define hipMemcpyDefault 4
inline hipError_t hipGetDeviceCount(int *n) {
   *n = 1;
   return 0;
}
inline hipError_t hipSetDevice(int) { return 0; }
inline hipError_t hipGetDevice(int *d) {
   *d = 0;
   return hipSuccess;
}
enum { hipDeviceAttributeMultiprocessorCount = 1, hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void *, int, int) { return 0; }
inline hipError_t hipDeviceGetAttribute(int *v, int, int) {
   *v = 3;   // a grid smaller than most test batches: the ticket loops of the persistent kernels get exercised
   return 0;
}
inline hipError_t hipMalloc(void **p, size_t n) {
   *p = malloc(n ? n : 1);
   return *p ? 0 : 2;
}
inline hipError_t hipFree(void *p) {
   free(p);
   return 0;
}
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
inline hipError_t hipHostFree(void *p) { return hipFree(p); }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, int) {
   memmove(d, s, n);
   return 0;
}
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, hipStream_t) {
   memmove(d, s, n);
   return 0;
}
inline hipError_t hipMemset(void *d, int v, size_t n) {
   memset(d, v, n);
   return 0;
}
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) {
   memset(d, v, n);
   return 0;
}
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
inline hipError_t hipDeviceSynchronize() { return 0; }
inline hipError_t hipStreamCreate(hipStream_t *s) {
   *s = nullptr;
   return 0;
}
#define hipStreamNonBlocking 1
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) {
   *s = nullptr;
   return 0;
}
inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t *s, uint32_t, const uint32_t *) {
   *s = nullptr;
   return 0;
}
inline hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) {
   *lo = 0;
   *hi = 0;
   return 0;
}
inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) {
   *s = nullptr;
   return 0;
}
inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
inline hipError_t hipGetLastError() { return 0; }
inline const char *hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipEventCreate(hipEvent_t *e) {
   *e = new zh_emu_event{0};
   return 0;
}
inline hipError_t hipEventDestroy(hipEvent_t e) {
   delete e;
   return 0;
}
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
typedef void *hipGraph_t;       // the emulator runs the plain kernel sequence; graphs only exist as null handles
typedef void *hipGraphExec_t;
inline hipError_t hipGraphDestroy(hipGraph_t) { return 0; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return 0; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) {
   *ms = 0.f;
   return 0;
}
