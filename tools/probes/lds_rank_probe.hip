// Probe (round 4): does a returning LDS atomic add, issued by the 64 lanes of a wave in ONE instruction, hand out its return values
// in lane order among the lanes that hit the same address? (The stable counting passes of zh_mf_group_lds.h would then get a lane's
// rank among the lanes of its digit from one ds_add_rtn_u32 instead of eight ballots and their mask arithmetic.)
// Also times the two ways over the same data.
//    hipcc --offload-arch=gfx950 -O3 -o lds_rank_probe lds_rank_probe.hip && ./lds_rank_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ uint32_t rnd(uint32_t x) {
   x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
   return x;
}

template <int MODE>   // 0: verify, 1: time the atomic way, 2: time the ballot way
__global__ void __launch_bounds__(1024) probe(uint32_t rounds, uint32_t *bad, uint32_t *sink) {
   __shared__ uint32_t hist[16][256];
   __shared__ uint32_t hist16[16][128];
   const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
   const uint64_t lt = (1ull << lane) - 1ull;
   uint32_t errors = 0, acc = 0;
   for (uint32_t r = 0; r < rounds; r++) {
      for (uint32_t k = lane; k < 256; k += 64) hist[wave][k] = 0;
      for (uint32_t k = lane; k < 128; k += 64) hist16[wave][k] = 0;
      __builtin_amdgcn_wave_barrier();
      const uint32_t alpha = 1u << (r % 9);   // 1 .. 256 distinct digits
      uint32_t base[4] = {0, 0, 0, 0};
      (void)base;
      for (uint32_t u = 0; u < 4; u++) {
         const uint32_t x = rnd(r * 7919u + blockIdx.x * 104729u + tid * 31u + u * 977u);
         const uint32_t d = ((x >> 8) % alpha) * (256u / alpha) + ((r & 16) ? 0u : (x & (256u / alpha - 1u)) * 0u);
         const bool valid = (x & 0x70000000u) != 0;   // 7 of 8 lanes take part
         if (MODE != 2) {
            uint32_t got = 0, got16 = 0;
            if (valid) {
               got = atomicAdd(&hist[wave][d], 1u);
               const uint32_t sh = (d & 1u) << 4;
               got16 = (atomicAdd(&hist16[wave][d >> 1], 1u << sh) >> sh) & 0xffffu;
            }
            acc += got + got16;
            if (MODE == 0) {
               uint64_t peers = __ballot(valid);
               for (int bit = 0; bit < 8; bit++) {
                  const bool one = (d >> bit) & 1u;
                  const uint64_t m = __ballot(valid && one);
                  peers &= one ? m : ~m;
               }
               // expected: what the earlier steps of this round put there + the lanes below this one with the same digit
               uint32_t before = 0;
               // (recomputed from scratch: count matching lanes of earlier steps)
               for (uint32_t v = 0; v < u; v++) {
                  const uint32_t xv = rnd(r * 7919u + blockIdx.x * 104729u + tid * 31u + v * 977u);
                  const uint32_t dv = ((xv >> 8) % alpha) * (256u / alpha);
                  const bool vv = (xv & 0x70000000u) != 0;
                  // every lane needs the count over ALL lanes of step v with digit == d: one ballot per lane value is too many; use LDS
                  (void)dv; (void)vv;
               }
               (void)before;
               const uint32_t expect_rank = (uint32_t)__popcll(peers & lt);
               // lanes of the same digit: (got - expect_rank) must be the same for all of them (= count of the earlier steps)
               const uint32_t b0 = got - expect_rank, b16 = got16 - expect_rank;
               const int leader = valid ? __ffsll((long long)peers) - 1 : (int)lane;
               const uint32_t lb0 = __shfl((int)b0, leader, 64), lb16 = __shfl((int)b16, leader, 64);
               if (valid && (b0 != lb0 || b16 != lb16 || b0 != b16)) errors++;
            }
         }
         else {
            uint64_t peers = __ballot(valid);
            for (int bit = 0; bit < 8; bit++) {
               const bool one = (d >> bit) & 1u;
               const uint64_t m = __ballot(valid && one);
               peers &= one ? m : ~m;
            }
            uint32_t before = 0;
            if (valid && (peers & lt) == 0) before = atomicAdd(&hist[wave][d], (uint32_t)__popcll(peers));
            before = __shfl((int)before, valid ? __ffsll((long long)peers) - 1 : (int)lane, 64);
            acc += before + (uint32_t)__popcll(peers & lt);
         }
      }
   }
   if (errors) atomicAdd(bad, errors);
   if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
   uint32_t *d_bad, *d_sink, bad = 0;
   hipMalloc(&d_bad, 4); hipMalloc(&d_sink, 4); hipMemset(d_bad, 0, 4);
   hipLaunchKernelGGL(probe<0>, dim3(1024), dim3(1024), 0, 0, 512u, d_bad, d_sink);
   hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost);
   printf("verify: %u mismatches over 1024 workgroups x 16 waves x 512 rounds x 4 steps\n", bad);
   hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
   for (int m = 1; m <= 2; m++) {
      float ms = 0;
      for (int it = 0; it < 2; it++) {
         hipEventRecord(a, 0);
         if (m == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(1024), 0, 0, 2048u, d_bad, d_sink);
         else hipLaunchKernelGGL(probe<2>, dim3(256), dim3(1024), 0, 0, 2048u, d_bad, d_sink);
         hipEventRecord(b, 0); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
      }
      printf("%s: %.3f ms for 2048 rounds of 4 steps per wave (%.0f cycles per step at 2.4 GHz, 4 waves per SIMD)\n", m == 1 ? "atomic rank (32-bit and packed 16-bit)" : "ballot rank", ms, ms * 1e-3 * 2.4e9 / (2048.0 * 4));
   }
   return bad ? 1 : 0;
}
