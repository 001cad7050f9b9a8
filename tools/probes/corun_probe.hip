#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define HOG_BODY                                                                  \
   lds[threadIdx.x] = threadIdx.x;                                                \
   __syncthreads();                                                               \
   long long t0 = clock64();                                                      \
   uint32_t v = 0;                                                                \
   while (clock64() - t0 < cycles) v += lds[(threadIdx.x + v) & 1023];            \
   if (v == 0x12345) out[0] = v;
__global__ void __launch_bounds__(1024) hog_static(uint32_t *out, long long cycles) { __shared__ uint32_t lds[131072 / 4]; HOG_BODY }
__global__ void __launch_bounds__(1024) hog_static_odd(uint32_t *out, long long cycles) { __shared__ uint32_t lds[131092 / 4]; HOG_BODY }
__global__ void __launch_bounds__(1024) hog_dyn(uint32_t *out, long long cycles) { extern __shared__ uint32_t lds[]; HOG_BODY }
__global__ void hog_dyn_nolb(uint32_t *out, long long cycles) { extern __shared__ uint32_t lds[]; HOG_BODY }
#define SMALL_BODY                                                                \
   lds[threadIdx.x] = in[threadIdx.x];                                            \
   __syncthreads();                                                               \
   uint32_t v = lds[(threadIdx.x * 7) & 63];                                      \
   for (int i = 0; i < 200; i++) v = v * 3 + i;                                   \
   out[blockIdx.x * 64 + threadIdx.x] = v;
__global__ void __launch_bounds__(64) small_static(uint32_t *out, const uint32_t *in) { __shared__ uint32_t lds[12048 / 4]; SMALL_BODY }
__global__ void __launch_bounds__(64) small_dyn(uint32_t *out, const uint32_t *in) { extern __shared__ uint32_t lds[]; SMALL_BODY }
__global__ void small_dyn_nolb(uint32_t *out, const uint32_t *in) { extern __shared__ uint32_t lds[]; SMALL_BODY }
int main() {
   uint32_t *out, *in; hipMalloc(&out, 64 << 20); hipMalloc(&in, 4096); hipMemset(in, 1, 4096);
   hipStream_t sa, sb; hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
   hipFuncSetAttribute((const void *)hog_dyn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
   hipFuncSetAttribute((const void *)hog_dyn_nolb, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
   hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
   const long long cyc = 7200000;
   for (int h = 0; h < 4; h++)
      for (int s = 0; s < 3; s++) {
         if (h == 0) hog_static<<<256, 1024, 0, sa>>>(out, cyc);
         if (h == 1) hog_static_odd<<<256, 1024, 0, sa>>>(out, cyc);
         if (h == 2) hog_dyn<<<256, 1024, 131072, sa>>>(out, cyc);
         if (h == 3) hog_dyn_nolb<<<256, 1024, 131072, sa>>>(out, cyc);
         hipError_t eh = hipGetLastError();
         hipStreamSynchronize(0);
         hipMemsetAsync(out, 0, 765 * 64 * 4, sb);
         hipEventRecord(e0, sb);
         if (s == 0) small_static<<<765, 64, 0, sb>>>(out, in);
         if (s == 1) small_dyn<<<765, 64, 12048, sb>>>(out, in);
         if (s == 2) small_dyn_nolb<<<765, 64, 12048, sb>>>(out, in);
         hipEventRecord(e1, sb);
         hipEventSynchronize(e1);
         float ms; hipEventElapsedTime(&ms, e0, e1);
         hipError_t es = hipGetLastError();
         hipDeviceSynchronize();
         uint32_t chk[2] = {0, 0};
         hipMemcpy(chk, out + 764 * 64 + 62, 8, hipMemcpyDeviceToHost);
         const char *hn[] = {"hog static 131072 lb1024", "hog static 131092 lb1024", "hog dynamic 131072 lb1024", "hog dynamic 131072 no launch bounds"};
         const char *sn[] = {"small static 12048 lb64", "small dynamic 12048 lb64", "small dynamic 12048 no lb"};
         printf("%-38s | %-26s : %7.3f ms   (launch errors %d %d, output written: %s)\n", hn[h], sn[s], ms, (int)eh, (int)es, chk[0] ? "yes" : "NO");
      }
   return 0;
}
