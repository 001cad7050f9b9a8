// does a CU-masked stream keep its workgroups off the masked-out CUs on MI355X (8 XCDs)?  hipcc --offload-arch=gfx950 -O3 -o cumask_probe cumask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void k_where(uint32_t *out) {
   uint32_t hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID, bits 0..31
   uint32_t xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));   // HW_REG_XCC_ID, bits 0..3
   // spin a little so that the grid spreads over every CU the stream may use
   uint64_t t0 = wall_clock64();
   while (wall_clock64() - t0 < 20000) {}
   if (threadIdx.x == 0) out[blockIdx.x] = (xcc << 16) | ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4) | (((hw >> 12) & 1) << 7);   // xcc | cu_id | se_id | sh_id
}
int main() {
   hipDeviceProp_t p;
   hipGetDeviceProperties(&p, 0);
   printf("CUs: %d\n", p.multiProcessorCount);
   uint32_t *d;
   hipMalloc(&d, 8192 * 4);
   for (int reserve : {0, 16, 32}) {
      uint32_t mask[8];
      for (int i = 0; i < 8; i++) mask[i] = 0xffffffffu;
      for (int b = 0; b < reserve; b++) mask[(255 - b) / 32] &= ~(1u << ((255 - b) % 32));
      hipStream_t s;
      hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
      if (e != hipSuccess) {
         printf("hipExtStreamCreateWithCUMask: %s\n", hipGetErrorString(e));
         return 1;
      }
      hipLaunchKernelGGL(k_where, dim3(4096), dim3(256), 0, s, d);
      hipStreamSynchronize(s);
      std::vector<uint32_t> h(4096);
      hipMemcpy(h.data(), d, 4096 * 4, hipMemcpyDeviceToHost);
      std::set<uint32_t> cus(h.begin(), h.end());
      std::set<uint32_t> xccs;
      for (uint32_t v : h) xccs.insert(v >> 16);
      printf("mask without the top %2d CUs: %zu distinct (xcc, se, sh, cu) used, %zu XCDs\n", reserve, cus.size(), xccs.size());
      hipStreamDestroy(s);
   }
   return 0;
}
