#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void hog(uint32_t *out, long long cycles) {
   extern __shared__ uint32_t lds[];
   lds[threadIdx.x] = threadIdx.x;
   __syncthreads();
   long long t0 = clock64();
   uint32_t v = 0;
   while (clock64() - t0 < cycles) v += lds[(threadIdx.x + v) & 1023];
   if (v == 0x12345) out[0] = v;
}
__global__ void small(uint32_t *out, const uint32_t *in) {
   extern __shared__ uint32_t lds[];
   lds[threadIdx.x] = in[threadIdx.x];
   __syncthreads();
   uint32_t v = lds[(threadIdx.x * 7) & 63];
   for (int i = 0; i < 200; i++) v = v * 3 + i;
   out[blockIdx.x * 64 + threadIdx.x] = v;
}
int main() {
   uint32_t *out, *in; hipMalloc(&out, 64 << 20); hipMalloc(&in, 4096); hipMemset(in, 1, 4096);
   hipStream_t sa, sb; hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
   hipFuncSetAttribute((const void *)hog, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
   hipFuncSetAttribute((const void *)small, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
   hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
   const long long cyc = 7200000;   // ~3 ms
   const int hthreads[] = {1024, 768, 512};
   const int hlds[] = {64, 96, 112, 120, 128, 131, 144};
   const int slds[] = {1, 2, 4, 8, 12, 16, 24, 32};
   printf("rows: hog (threads, LDS KB); columns: small-kernel LDS KB; R = ran next to the hog, - = waited for it\n%-14s", "");
   for (int s : slds) printf("%4d", s);
   printf("\n");
   for (int ht : hthreads)
      for (int hl : hlds) {
         printf("%5d thr %3dK ", ht, hl);
         for (int sl : slds) {
            hog<<<256, ht, hl * 1024, sa>>>(out, cyc);
            hipStreamSynchronize(0);
            hipEventRecord(e0, sb);
            small<<<765, 64, sl * 1024, sb>>>(out, in);
            hipEventRecord(e1, sb);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipDeviceSynchronize();
            printf("%4s", ms < 1.0f ? "R" : "-");
         }
         printf("\n");
      }
   return 0;
}
