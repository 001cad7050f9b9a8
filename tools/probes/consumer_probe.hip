// micro-benchmark of the zh_parse_huge consumer step on gfx950: which part of the step costs what?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define KEYMASK 0x0801ffffu
#define BIAS (1u << 14)
__device__ __forceinline__ uint32_t row_min(uint32_t v) {
   v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0xffffffff, v, 0xB1, 0xf, 0xf, false));
   v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0xffffffff, v, 0x4E, 0xf, 0xf, false));
   v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0xffffffff, v, 0x141, 0xf, 0xf, false));
   v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0xffffffff, v, 0x140, 0xf, 0xf, false));
   return v;
}
__device__ __forceinline__ uint32_t rl(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
struct ws_t { uint16_t ring[512]; uint4 desc[16][64]; uint32_t bt[16][4]; uint32_t lit[16][4]; };
__device__ __forceinline__ uint32_t ring_at(const ws_t &ws, uint32_t d) { return *(const uint16_t *)((const uint8_t *)ws.ring + (d >> 17 & 0x3feu)); }

template <int MODE>
__global__ void probe(uint32_t *out, uint64_t *tm, int tiles) {
   __shared__ ws_t ws;
   const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, row = lane >> 4, s = lane & 15;
   for (uint32_t i = tid; i < 16 * 64 * 4; i += blockDim.x) ((uint32_t *)ws.desc)[i] = ((((i * 37u) & 511u) * 2u) << 17) | ((i & 63u) << 9) | (i & 0x1ffu);
   for (uint32_t i = tid; i < 512; i += blockDim.x) ws.ring[i] = (uint16_t)(i * 3);
   for (uint32_t i = tid; i < 64; i += blockDim.x) ((uint32_t *)ws.lit)[i] = 8 + (i & 3);
   __syncthreads();
   uint64_t c0 = clock64();
   if (wave == 0) {
      uint32_t cnext = 0, C0 = 0, C1 = 0, C2 = 0;
      const bool sel0 = s == row, sel1 = s + 1 == row, patched = s <= row && row < 3;
      uint32_t thi = 60000;
      for (int k = 0; k < tiles; k++) {
         uint4 D = ws.desc[0][lane], D1 = ws.desc[1][lane];
         uint32_t lt = ws.lit[0][row];
         uint32_t g0 = ring_at(ws, D.x), g1 = ring_at(ws, D.y), g2 = ring_at(ws, D.z), g3 = ring_at(ws, D.w);
         for (uint32_t t = 0; t < 16; t++) {
            uint4 D2 = D1; uint32_t ltn = lt, n0 = g0, n1 = g1, n2 = g2, n3 = g3;
            if (MODE != 1) {
               D2 = ws.desc[min(t + 2, 15u)][lane];
               ltn = ws.lit[min(t + 1, 15u)][row];
               n0 = ring_at(ws, D1.x); n1 = ring_at(ws, D1.y); n2 = ring_at(ws, D1.z); n3 = ring_at(ws, D1.w);
            }
            const uint32_t base = cnext - BIAS;
            const uint32_t cv = sel0 ? C0 : (sel1 ? C1 : C2);
            const uint32_t r0 = patched ? cv : g0;
            const uint32_t k0 = (((r0 - base) & 0xffffu) << 9) + (D.x & KEYMASK);
            const uint32_t k1 = (((g1 - base) & 0xffffu) << 9) + (D.y & KEYMASK);
            const uint32_t k2 = (((g2 - base) & 0xffffu) << 9) + (D.z & KEYMASK);
            const uint32_t k3 = (((g3 - base) & 0xffffu) << 9) + (D.w & KEYMASK);
            uint32_t key = min(min(k0, k1), min(k2, k3));
            const uint32_t rkey = MODE == 2 ? key : row_min(key);
            uint32_t c0_, c1_, c2_, l0, l1, l2;
            if (MODE == 3) {   // no readlane / scalar chain: lane-local fake
               l0 = lt + BIAS; c0_ = min(l0, rkey >> 9); l1 = lt + c0_; c1_ = min(l1, rkey >> 10); l2 = lt + c1_; c2_ = min(l2, rkey >> 11);
            } else {
               const uint32_t m0 = rl(rkey, 0) >> 9, m1 = rl(rkey, 16) >> 9, m2 = rl(rkey, 32) >> 9;
               l0 = rl(lt, 0) + BIAS; c0_ = min(l0, m0);
               l1 = rl(lt, 16) + c0_; c1_ = min(l1, m1);
               l2 = rl(lt, 32) + c1_; c2_ = min(l2, m2);
            }
            C0 = (base + c0_) & 0xffffu; C1 = (base + c1_) & 0xffffu; C2 = (base + c2_) & 0xffffu;
            if (MODE != 4) {
               const uint32_t lrow = row == 0 ? l0 : (row == 1 ? l1 : l2), mrow = rkey >> 9;
               if (s == 0 && row < 3) {
                  ws.ring[(thi - 1 - 3 * t - row) & 511] = (uint16_t)(base + min(lrow, mrow));
                  ws.bt[t][row] = mrow < lrow ? rkey : 0xFFFFFFFFu;
               }
            }
            __builtin_amdgcn_wave_barrier();
            cnext = C2;
            D = D1; D1 = D2; lt = ltn; g0 = n0; g1 = n1; g2 = n2; g3 = n3;
         }
         thi -= 48;
         if (MODE == 5) __syncthreads();
      }
      out[lane] = cnext + C0 + C1;
   } else if (MODE == 5 || MODE == 6) {
      // dummy producers: LDS store traffic like the staging (64 stores per thread per tile), conflict-free
      for (int k = 0; k < tiles; k++) {
         for (uint32_t i = 0; i < 16; i++) ((uint32_t *)ws.desc)[((tid - 64) * 16 + ((i + tid) & 15)) & 4095] = i + k;
         if (MODE == 5) __syncthreads();
      }
   }
   uint64_t c1 = clock64();
   if (tid == 0) tm[0] = c1 - c0;
}
template <int MODE> void run(const char *name, int threads, uint32_t *out, uint64_t *t) {
   const int tiles = 2000;
   for (int rep = 0; rep < 2; rep++) {
      probe<MODE><<<1, threads>>>(out, t, tiles);
      hipDeviceSynchronize();
   }
   uint64_t h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
   printf("%-44s %7.1f cycles/step\n", name, (double)h / (tiles * 16.0));
}
int main() {
   uint32_t *out; uint64_t *t;
   hipMalloc(&out, 4096); hipMalloc(&t, 16);
   run<0>("full step, consumer alone", 64, out, t);
   run<1>("  without the LDS prefetch reads", 64, out, t);
   run<2>("  without the DPP row minimum", 64, out, t);
   run<3>("  without readlanes / scalar chain", 64, out, t);
   run<4>("  without the LDS writes", 64, out, t);
   run<5>("full step + 3 producer waves, barrier/tile", 256, out, t);
   run<6>("full step + 3 store-only waves, no barrier", 256, out, t);
   return 0;
}
