// how many single-wave workgroups with L bytes of LDS are resident per CU? time of N spinning workgroups / spin time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void __launch_bounds__(64) spin(uint32_t *out, long long cycles) {
   extern __shared__ uint32_t lds[];
   lds[threadIdx.x] = threadIdx.x;
   long long t0 = clock64();
   uint32_t v = 0;
   while (clock64() - t0 < cycles) v += lds[(threadIdx.x + v) & 63];
   if (v == 0x12345) out[0] = v;
}
int main() {
   uint32_t *out; hipMalloc(&out, 4096);
   hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
   const long long cyc = 240000;   // 100 us
   const int lds[] = {0, 2052, 4096, 5600, 6736, 7168, 8272, 12368, 16464};
   for (int l : lds) {
      const int N = 256 * 64;   // 64 workgroups per CU if all fit
      spin<<<N, 64, l>>>(out, cyc); hipDeviceSynchronize();
      hipEventRecord(e0); spin<<<N, 64, l>>>(out, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("LDS %6d B: %7.3f ms for %d workgroups of 100 us -> about %.1f resident per CU\n", l, ms, N, 64.0 / (ms / 0.1));
   }
   return 0;
}
