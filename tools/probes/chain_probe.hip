// micro-benchmark: what does one wave, alone on a CU, pay per dependent instruction on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N 4096
__global__ void probe(uint32_t *out, uint64_t *t, int mode) {
   __shared__ uint32_t lds[1024];
   uint32_t v = threadIdx.x, w = out[threadIdx.x & 7];
   lds[threadIdx.x] = v;
   __syncthreads();
   uint64_t c0 = clock64(), r0 = wall_clock64();
   if (mode == 0) {
#pragma unroll 16
      for (int i = 0; i < N; i++) v = v * 3 + w;                 // dependent VALU (mad)
   } else if (mode == 1) {
#pragma unroll 16
      for (int i = 0; i < N; i++) v = v + w + i;                 // dependent add
   } else if (mode == 2) {
#pragma unroll 16
      for (int i = 0; i < N; i++) {                              // valu -> readlane -> salu -> valu
         uint32_t s = __builtin_amdgcn_readlane(v, 16);
         v = v + s * 5 + 1;
      }
   } else if (mode == 3) {
#pragma unroll 16
      for (int i = 0; i < N; i++) {                              // 4-step DPP row min + add
         v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0xffffffff, v, 0xB1, 0xf, 0xf, false));
         v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0xffffffff, v, 0x4E, 0xf, 0xf, false));
         v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0xffffffff, v, 0x141, 0xf, 0xf, false));
         v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0xffffffff, v, 0x140, 0xf, 0xf, false));
         v += w;
      }
   } else if (mode == 4) {
#pragma unroll 16
      for (int i = 0; i < N; i++) {                              // LDS write -> dependent read
         lds[(threadIdx.x + i) & 1023] = v;
         v = lds[(threadIdx.x + i + w) & 1023] + 1;
      }
   } else if (mode == 5) {
      for (int i = 0; i < N; i++) {                              // divergent-looking branch (uniformly taken) per iteration
         if ((v + i) & 0x10000000) v ^= w; else v += 7;
         if (__builtin_amdgcn_readfirstlane(v) == 0x12345) v++;
      }
   } else if (mode == 6) {
#pragma unroll 16
      for (int i = 0; i < N; i++) {                              // dependent LDS read chain (pointer chase)
         v = lds[v & 1023] + w;
      }
   } else if (mode == 7) {
      uint32_t a = v, b = v + 1, c = v + 2, d = v + 3;
#pragma unroll 16
      for (int i = 0; i < N; i++) { a = a * 3 + w; b = b * 5 + w; c = c * 7 + w; d = d * 9 + w; }   // 4 independent chains
      v = a + b + c + d;
   }
   uint64_t c1 = clock64(), r1 = wall_clock64();
   out[threadIdx.x] = v;
   if (threadIdx.x == 0) { t[0] = c1 - c0; t[1] = r1 - r0; }
}
int main() {
   uint32_t *out; uint64_t *t;
   hipMalloc(&out, 4096); hipMemset(out, 0, 4096); hipMalloc(&t, 16);
   const char *names[] = {"dependent v_mad", "dependent v_add3", "readlane->salu->valu", "4xDPP min + add", "LDS write->read", "branchy", "LDS pointer chase", "4 independent mads"};
   for (int rep = 0; rep < 2; rep++)
   for (int mode = 0; mode < 8; mode++) {
      probe<<<1, 64>>>(out, t, mode);
      hipDeviceSynchronize();
      uint64_t h[2]; hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
      printf("%-24s %8.1f shader-clk/iter  %8.2f ns/iter  (memtime/realtime ratio %.2f)\n", names[mode], (double)h[0] / N, (double)h[1] * 10.0 / N, (double)h[0] / ((double)h[1]));
   }
   return 0;
}
