// zh_platform.h — gfx950 (MI355X / CDNA4) platform layer: HIP runtime + wave64 primitives.
//
// Everything in csrc/ includes this as <zh_platform.h>. The only other provider of that header name is
// tests/emu/zh_platform.h, a lock-step CPU emulator used by the `-m "not gpu"` tests to run the kernel
// logic on a GPU-less machine; the product build never sees it (include path order in zultra_amd/build.py).
//
// Wave primitives are written for wave64 directly: DPP row operations for the intra-row steps, scalar
// readlane for the cross-row step. No warp-32 idioms, no CUDA compatibility shims.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ZH_WAVE 64

#ifdef ZH_TRACE_LAUNCH
// probe builds only (tools/build_variant.sh trace -DZH_TRACE_LAUNCH): every launch is named on stderr, waited for, and its outcome printed — the
// kernel a memory fault belongs to is the last one named
#include <stdio.h>
#define ZH_LAUNCH_LDS(kernel, grid, block, lds, stream, ...)                                                                    \
   do {                                                                                                                         \
      fprintf(stderr, "launch %s grid %u block %u\n", #kernel, (unsigned)(grid), (unsigned)(block));                            \
      (void)hipDeviceSynchronize();                                                                                             \
      hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), (lds), (hipStream_t)(stream), __VA_ARGS__);                           \
      const hipError_t e_ = hipDeviceSynchronize();                                                                             \
      fprintf(stderr, "   done %s: %s\n", #kernel, hipGetErrorString(e_));                                                      \
   } while (0)
#define ZH_LAUNCH(kernel, grid, block, stream, ...) ZH_LAUNCH_LDS(kernel, grid, block, 0, stream, __VA_ARGS__)
#else
#define ZH_LAUNCH(kernel, grid, block, stream, ...) \
   hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, (hipStream_t)(stream), __VA_ARGS__)
// launch with `lds` bytes of dynamic LDS; the kernel declares it with ZH_DYN_LDS(name) and carves its arrays out of it
#define ZH_LAUNCH_LDS(kernel, grid, block, lds, stream, ...) \
   hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), (lds), (hipStream_t)(stream), __VA_ARGS__)
#endif
#define ZH_DYN_LDS(name) extern __shared__ uint32_t name[]

__device__ __forceinline__ unsigned zh_lane() { return __lane_id(); }
__device__ __forceinline__ uint64_t zh_ballot(bool p) { return __ballot(p); }
__device__ __forceinline__ uint32_t zh_shfl(uint32_t v, int src_lane) { return (uint32_t)__shfl((int)v, src_lane, 64); }
// lane must be wave-uniform
__device__ __forceinline__ uint32_t zh_readlane(uint32_t v, int lane) {
   return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}
// a load that sees other workgroups' device-scope atomics (not a stale line of this XCD's L2)
__device__ __forceinline__ uint32_t zh_load_relaxed(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t zh_readfirstlane(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ int zh_popc64(uint64_t m) { return __popcll(m); }
__device__ __forceinline__ int zh_ctz64(uint64_t m) { return __ffsll((long long)m) - 1; }
__device__ __forceinline__ int zh_clz32(uint32_t v) { return __clz((int)v); }

// DPP controls (CDNA ISA): quad_perm, row_half_mirror, row_mirror
#define ZH_DPP_QUAD_XOR1 0xB1
#define ZH_DPP_QUAD_XOR2 0x4E
#define ZH_DPP_ROW_HALF_MIRROR 0x141
#define ZH_DPP_ROW_MIRROR 0x140

template <int CTRL>
__device__ __forceinline__ uint32_t zh_dpp(uint32_t v) {
   return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}

// value of lane (l - N) / (l + N) of the same 16-lane DPP row; lanes without such a neighbour keep their own value
template <int N>
__device__ __forceinline__ uint32_t zh_row_shr(uint32_t v) {
   return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x110 + N, 0xF, 0xF, false);
}
template <int N>
__device__ __forceinline__ uint32_t zh_row_shl(uint32_t v) {
   return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x100 + N, 0xF, 0xF, false);
}

// whole-wave shift by one lane: lane l receives lane l-1's value, lane 0 receives `feed` (DPP wave_shr:1)
__device__ __forceinline__ uint32_t zh_wave_shr1(uint32_t v, uint32_t feed) {
   return (uint32_t)__builtin_amdgcn_update_dpp((int)feed, (int)v, 0x138, 0xF, 0xF, false);
}

// minimum over each 16-lane DPP row, result in every lane of the row (4 DPP steps, no LDS traffic)
// `old` = the identity of min: lets the compiler fold each step into one v_min_u32_dpp
template <int CTRL>
__device__ __forceinline__ uint32_t zh_dpp_min(uint32_t v) {
   return min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ uint32_t zh_row_min(uint32_t v) {
   v = zh_dpp_min<ZH_DPP_QUAD_XOR1>(v);
   v = zh_dpp_min<ZH_DPP_QUAD_XOR2>(v);
   v = zh_dpp_min<ZH_DPP_ROW_HALF_MIRROR>(v);
   v = zh_dpp_min<ZH_DPP_ROW_MIRROR>(v);
   return v;
}
__device__ __forceinline__ uint32_t zh_row_sum(uint32_t v) {
   v += zh_dpp<ZH_DPP_QUAD_XOR1>(v);
   v += zh_dpp<ZH_DPP_QUAD_XOR2>(v);
   v += zh_dpp<ZH_DPP_ROW_HALF_MIRROR>(v);
   v += zh_dpp<ZH_DPP_ROW_MIRROR>(v);
   return v;
}
// quads (four neighbouring lanes, DPP quad_perm): minimum over the quad in every lane; lane q <- lane q-1 (lane 0 keeps its own);
// lanes 2, 3 <- lanes 0, 1 (lanes 0, 1 keep their own)
__device__ __forceinline__ uint32_t zh_quad_min(uint32_t v) {
   v = zh_dpp_min<ZH_DPP_QUAD_XOR1>(v);
   v = zh_dpp_min<ZH_DPP_QUAD_XOR2>(v);
   return v;
}
// (every lane reads a lane of its quad: no "old" value is needed, which saves the copy that keeps it)
__device__ __forceinline__ uint32_t zh_quad_shr1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x90, 0xF, 0xF, true); }   // quad_perm:[0,0,1,2]
__device__ __forceinline__ uint32_t zh_quad_lo2(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x44, 0xF, 0xF, true); }    // quad_perm:[0,1,0,1]
// whole-wave reductions: row step on the VALU, the 4 row results combined on the scalar unit
__device__ __forceinline__ uint32_t zh_wave_min(uint32_t v) {
   v = zh_row_min(v);
   uint32_t a = zh_readlane(v, 0), b = zh_readlane(v, 16), c = zh_readlane(v, 32), d = zh_readlane(v, 48);
   return min(min(a, b), min(c, d));
}
// whole-wave minimum by DPP alone: the four row minima are combined with row_bcast:15 / row_bcast:31 and arrive in lane 63
// (one readlane instead of four and no scalar mins: the shortest instruction sequence for a wave that runs alone)
__device__ __forceinline__ uint32_t zh_wave_min_bcast(uint32_t v) {
   v = zh_row_min(v);
   v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0x142, 0xA, 0xF, false));   // rows 1, 3 <- lane 15 of rows 0, 2
   v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0x143, 0xC, 0xF, false));   // rows 2, 3 <- lane 31
   return zh_readlane(v, 63);
}
// the same for three values at once, the steps interleaved: a DPP instruction needs its source two cycles old, and three
// independent chains fill each other's gaps. The minima arrive in lane 63 of a, b, c.
template <int CTRL, int ROWS>
__device__ __forceinline__ uint32_t zh_dpp_min_rows(uint32_t v) {
   return min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, CTRL, ROWS, 0xF, false));
}
__device__ __forceinline__ void zh_wave_min3_lane63(uint32_t &a, uint32_t &b, uint32_t &c) {
   a = zh_dpp_min<ZH_DPP_QUAD_XOR1>(a); b = zh_dpp_min<ZH_DPP_QUAD_XOR1>(b); c = zh_dpp_min<ZH_DPP_QUAD_XOR1>(c);
   a = zh_dpp_min<ZH_DPP_QUAD_XOR2>(a); b = zh_dpp_min<ZH_DPP_QUAD_XOR2>(b); c = zh_dpp_min<ZH_DPP_QUAD_XOR2>(c);
   a = zh_dpp_min<ZH_DPP_ROW_HALF_MIRROR>(a); b = zh_dpp_min<ZH_DPP_ROW_HALF_MIRROR>(b); c = zh_dpp_min<ZH_DPP_ROW_HALF_MIRROR>(c);
   a = zh_dpp_min<ZH_DPP_ROW_MIRROR>(a); b = zh_dpp_min<ZH_DPP_ROW_MIRROR>(b); c = zh_dpp_min<ZH_DPP_ROW_MIRROR>(c);
   a = zh_dpp_min_rows<0x142, 0xA>(a); b = zh_dpp_min_rows<0x142, 0xA>(b); c = zh_dpp_min_rows<0x142, 0xA>(c);
   a = zh_dpp_min_rows<0x143, 0xC>(a); b = zh_dpp_min_rows<0x143, 0xC>(b); c = zh_dpp_min_rows<0x143, 0xC>(c);
}
__device__ __forceinline__ uint32_t zh_wave_sum(uint32_t v) {
   v = zh_row_sum(v);
   return zh_readlane(v, 0) + zh_readlane(v, 16) + zh_readlane(v, 32) + zh_readlane(v, 48);
}
// exclusive prefix sum over the 64 lanes: Hillis-Steele inside each 16-lane row (row_shr 1, 2, 4, 8, lanes without a source add 0), then the
// row totals travel down with row_bcast:15 / row_bcast:31 — ten DPP adds on the VALU. (Rounds 1-3: six __shfl_up steps, each a
// ds_bpermute round trip through the LDS crossbar behind the last.)
__device__ __forceinline__ uint32_t zh_wave_excl_sum(uint32_t v) {
   uint32_t x = v;
   x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);
   x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);
   x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);
   x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);
   x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);   // rows 1, 3 += lane 15 of rows 0, 2
   x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);   // rows 2, 3 += lane 31
   return x - v;
}

// inclusive running maximum over the 64 lanes (unsigned values; 0 is the identity): the same ten DPP steps with max
__device__ __forceinline__ uint32_t zh_wave_incl_max(uint32_t v) {
   uint32_t x = v;
   x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true));
   x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true));
   x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true));
   x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true));
   x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false));
   x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false));
   return x;
}

// issue priority of the calling wave (0..3): the SIMD's arbiter serves the highest priority first, then the oldest wave. A wave that
// carries a serial dependency chain next to throughput waves of other kernels needs it: at equal priority it gets one issue
// slot in as many as there are ready waves on its SIMD.
__device__ __forceinline__ void zh_set_wave_priority_high() { __builtin_amdgcn_s_setprio(3); }
__device__ __forceinline__ void zh_set_wave_priority_mid() { __builtin_amdgcn_s_setprio(2); }
__device__ __forceinline__ void zh_set_wave_priority_normal() { __builtin_amdgcn_s_setprio(0); }

// shader-clock stamp for the optional in-kernel phase profile (diagnostics only)
__device__ __forceinline__ uint64_t zh_clock() { return (uint64_t)clock64(); }
// constant-rate (100 MHz) device clock, comparable between workgroups
__device__ __forceinline__ uint64_t zh_wall_clock() { return (uint64_t)wall_clock64(); }

// a global word written earlier in this kernel (by this or another workgroup, after its fence): read past the CU's vector cache
__device__ __forceinline__ uint32_t zh_load_agent_u32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ uint32_t zh_load_agent_u16(const uint16_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// the 32 bits at bit offset (sh & 31) of hi:lo — one v_alignbit_b32 (the funnel shift of an unaligned 4-byte read from two aligned words)
__device__ __forceinline__ uint32_t zh_funnel(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }

// Loads at any byte address. gfx950 serves them in hardware (LDS and global: the HSA targets run in unaligned access mode, and the
// compiler emits one ds_read_b32 / ds_read_b128 for these types), so a string probe into the window copy in LDS is one
// instruction, not two aligned reads and a funnel shift. zultra_hip_selftest checks it on the device it runs on.
typedef uint32_t __attribute__((aligned(1))) zh_u32_any_t;
struct __attribute__((packed, aligned(1))) zh_u128_any_t {
   uint32_t x, y, z, w;
};
__device__ __forceinline__ uint32_t zh_load32_any(const void *p) { return *(const zh_u32_any_t *)p; }
__device__ __forceinline__ zh_u128_any_t zh_load128_any(const void *p) { return *(const zh_u128_any_t *)p; }

// LDS visibility between the lanes of a workgroup (a single wave for the 64-thread kernels)
__device__ __forceinline__ void zh_sync() { __syncthreads(); }
// Workgroup barrier that orders LDS traffic only. __syncthreads() also drains the wave's outstanding GLOBAL loads
// (s_waitcnt vmcnt(0) before s_barrier): a wave that keeps row loads in flight across several barriers — the producers of
// zh_parse_chain.h — would pay a full memory round trip at every one of them.
__device__ __forceinline__ void zh_sync_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// Global loads whose completion the CALLER tracks: issue now, use after zh_async_wait<N>() — N = how many loads issued later
// may still be in flight (vector-memory operations of a wave complete in order). The compiler's own bookkeeping gives up
// on loads kept in flight around a loop (it waits for all of them, the youngest included, at the first use of any), which
// turns a three-tile read-ahead into none. The values must not be touched before the wait.
typedef uint32_t zh_u32x4_t __attribute__((ext_vector_type(4)));
struct zh_async_row_t {
   zh_u32x4_t a, b;   // the two planes of a match row (zh_common.h)
   uint32_t byte;
};
__device__ __forceinline__ void zh_async_load_row(zh_async_row_t &r, const uint4 *lo, const uint4 *hi, const uint8_t *byte) {
   asm volatile("global_load_dwordx4 %0, %3, off\n\tglobal_load_dwordx4 %1, %4, off\n\tglobal_load_ubyte %2, %5, off"
                : "=&v"(r.a), "=&v"(r.b), "=&v"(r.byte)
                : "v"(lo), "v"(hi), "v"(byte)
                : "memory");
}
#define ZH_ASYNC_ROW_LOADS 3   // vector-memory operations per zh_async_load_row
template <int N>
__device__ __forceinline__ void zh_async_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// after the wait: the registers now hold the loaded values (keeps the compiler from having moved their use above the wait)
__device__ __forceinline__ void zh_async_landed(zh_async_row_t &r) { asm volatile("" : "+v"(r.a), "+v"(r.b), "+v"(r.byte)::"memory"); }

// ... and one tile of a forward walk over a parse (zh_parse.h, ZH_WALK_*): a parse entry and a byte per lane
struct zh_async_tile_t {
   uint32_t b, y;
};
__device__ __forceinline__ void zh_async_load_tile(zh_async_tile_t &t, const uint32_t *pb, const uint8_t *py) {
   asm volatile("global_load_dword %0, %2, off\n\tglobal_load_ubyte %1, %3, off" : "=&v"(t.b), "=&v"(t.y) : "v"(pb), "v"(py) : "memory");
}
#define ZH_ASYNC_TILE_LOADS 2   // vector-memory operations per zh_async_load_tile
__device__ __forceinline__ void zh_async_landed(zh_async_tile_t &t) { asm volatile("" : "+v"(t.b), "+v"(t.y)::"memory"); }

// LDS visibility between the lanes of ONE wave (wave-private data inside a multi-wave workgroup): LDS operations of a
// wave execute in order, so draining them and stopping the compiler from moving accesses across is all it takes.
__device__ __forceinline__ void zh_wave_sync() {
   __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
   __builtin_amdgcn_wave_barrier();
   __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Between an LDS store by one lane and a load of it by another lane of the SAME wave: the hardware runs a wave's LDS operations in
// order, so all it takes is that the compiler keeps them in order too. Unlike zh_wave_sync no counter is drained: global loads
// and LDS reads issued earlier stay in flight.
__device__ __forceinline__ void zh_lockstep_sync() {
   __builtin_amdgcn_wave_barrier();
   asm volatile("" ::: "memory");
}

// The lanes (among those with `valid`) that hold the same 8-bit digit as the calling lane, itself included: one ballot per bit, and per lane
// the OR over the bits of "lanes whose bit differs from mine" (ballot XOR own bit spread over the word).
__device__ __forceinline__ uint64_t zh_peers8(uint32_t d, bool valid) {
   // per bit: the lane's bit spread over a word (v_bfe_i32), a ballot of it, and "differs from mine" ORed in with one three-input operation per half
   // (v_bitop3_b32: a | (b ^ c)) — 32 instructions; written with shifts and plain operators (rounds 1-4) the compiler made 48 of it. zh_selftest_kernel
   // checks the result against a lane-by-lane comparison on the device.
   uint32_t dlo = 0, dhi = 0;
#pragma unroll
   for (int bit = 0; bit < 8; bit++) {
      uint32_t s = (uint32_t)__builtin_amdgcn_sbfe((int)d, (uint32_t)bit, 1u);   // all ones where this lane's bit is set
      asm volatile("" : "+v"(s));                                                // (the compare below takes s, not a shifted copy of d)
      const uint64_t m = __ballot(s != 0);                                       // (lanes without `valid` are masked out at the end)
      dlo = __builtin_amdgcn_bitop3_b32(dlo, (uint32_t)m, s, 0xf6);              // dlo | (m ^ s)
      dhi = __builtin_amdgcn_bitop3_b32(dhi, (uint32_t)(m >> 32), s, 0xf6);
   }
   const uint64_t v = __ballot(valid);
   return v & ~(((uint64_t)dhi << 32) | dlo);
}
// number of set bits of m below the calling lane (v_mbcnt_lo / v_mbcnt_hi)
__device__ __forceinline__ uint32_t zh_rank_below(uint64_t m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }

// Between two instructions whose ORDER ACROSS THE LANES matters (a returning LDS atomic per step of an unrolled loop: every lane's step u
// before any lane's step u + 1): on the GPU a wave executes in lock step and this is nothing — no compiler barrier either, the loads of the
// steps stay in flight together; the CPU emulator of tests/emu runs a lane until its next collective, and makes this one.
__device__ __forceinline__ void zh_lockstep_point() {}

// a wave's own global stores are out (before a barrier after which other waves of the workgroup, or the wave itself, read them back)
__device__ __forceinline__ void zh_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ uint32_t zh_atomic_add_lds(uint32_t *p, uint32_t v) { return atomicAdd(p, v); }
__device__ __forceinline__ uint32_t zh_atomic_add_global(uint32_t *p, uint32_t v) { return atomicAdd(p, v); }
