// micro-benchmark of zh_parse_chain.h on one MI355X: cycles per position of the consumer alone, of a whole chain workgroup,
// and of many chain workgroups per CU.   hipcc --offload-arch=gfx950 -O3 -I ../../zultra_amd/csrc -o chain2_probe chain2_probe.hip
#define ZH_CHAIN_PROFILE 1
#include <zh_platform.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "zh_common.h"
#include "zh_huffman.h"
#include "zh_split.h"
#include "zh_parse.h"
#include "zh_parse_chain.h"

__global__ void __launch_bounds__(256) k_consumer_only(uint64_t *out, int ntiles) {
   __shared__ zh_chain_ws_t ws;
   const uint32_t tid = threadIdx.x;
   for (uint32_t k = tid; k < 2 * ZH_CHAIN_TILE * 65; k += 256) ((uint32_t *)ws.p.desc)[k] = ((k * 2654435761u) >> 26 << 9) | ((k & 7) << 6) | 5;   // prices 0..63
   for (uint32_t k = tid; k < ZH_CHAIN_RING; k += 256) ws.p.ring[k] = 0u - (k << 23);
   for (uint32_t k = tid; k < 2 * ZH_CHAIN_TILE; k += 256) ((uint32_t *)ws.p.lit)[k] = 8u << 9;
   __syncthreads();
   if (tid >= 64) return;
   zh_chain_state_t st;
   st.cv = 0; st.c1 = 0; st.c2 = 0;
   const uint64_t c0 = clock64();
   for (int k = 0; k < ntiles; k++) {
      if (st.c1 >= ZH_CHAIN_REBASE) zh_chain_rebase(ws, st);
      zh_chain_consume(ws, k & 1, 1000000u - 32u * k, st);
   }
   const uint64_t c1 = clock64();
   if (tid == 0) {
      out[0] = c1 - c0;
      out[1] = st.c1;
   }
}

__global__ void __launch_bounds__(256) k_chain(const uint4 *rows, const uint8_t *win, uint32_t n, uint32_t *best, uint64_t *out) {
   __shared__ zh_chain_ws_t ws;
   const uint32_t tid = threadIdx.x;
   for (uint32_t k = tid; k < ZH_NLIT; k += 256) ws.litprice[k] = 8;
   for (uint32_t k = tid; k < 256; k += 256) ws.lencost[k] = 7 + (k >> 5);
   if (tid < ZH_NDIST) ws.distcost[tid] = 5 + tid / 2;
   __syncthreads();
   const uint64_t off = (uint64_t)blockIdx.x * n;
   const uint64_t c0 = clock64(), r0 = wall_clock64();
   zh_chain_parse(ws, rows + off, rows + off + (uint64_t)gridDim.x * n, win + off, 0, 0, n, n, best + off);
   const uint64_t c1 = clock64(), r1 = wall_clock64();
   if (tid == 0) {
      out[2 * blockIdx.x] = c1 - c0;
      out[2 * blockIdx.x + 1] = r1 - r0;
   }
}

// background load: single-wave workgroups doing what zh_parse_tasks does most (LDS reads, VALU, DPP), 5.7 KB of LDS each
__global__ void __launch_bounds__(64) k_noise(uint32_t *out, int iters) {
   __shared__ uint32_t lds[1428];
   uint32_t v = threadIdx.x + blockIdx.x;
   for (uint32_t k = threadIdx.x; k < 1428; k += 64) lds[k] = k * 2654435761u;
   __syncthreads();
   for (int i = 0; i < iters; i++) {
      v += lds[(v + i) % 1428u];
      v = min(v * 3u + 1u, (uint32_t)__builtin_amdgcn_update_dpp(0xffffffff, v, 0xB1, 0xf, 0xf, false));
      v ^= lds[(v >> 7) % 1428u];
   }
   out[blockIdx.x * 64 + threadIdx.x] = v;
}

int main() {
   uint64_t *d_out;
   hipMalloc(&d_out, 8192 * 16);
   uint64_t h[4];
   for (int rep = 0; rep < 2; rep++) {
      hipLaunchKernelGGL(k_consumer_only, dim3(1), dim3(256), 0, 0, d_out, 2000);
      hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost);
   }
   printf("consumer alone: %.1f cycles per position\n", (double)h[0] / (2000.0 * 32));
   const uint32_t n = 65536;
   uint32_t *d_noise;
   hipMalloc(&d_noise, 60000 * 64 * 4);
   hipStream_t s_noise, s_chain;
   hipStreamCreateWithFlags(&s_noise, hipStreamNonBlocking);
   int lo_p = 0, hi_p = 0;
   hipDeviceGetStreamPriorityRange(&lo_p, &hi_p);
   hipStreamCreateWithPriority(&s_chain, hipStreamNonBlocking, hi_p);
   for (int wgs : {1, -1, 1024}) {
      const bool noisy = wgs < 0;
      if (noisy) wgs = 1;
      const size_t tot = (size_t)wgs * n;
      std::vector<uint4> rows(2 * tot);
      std::vector<uint8_t> win(tot);
      uint32_t s = 12345;
      auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
      for (size_t i = 0; i < tot; i++) {
         win[i] = (uint8_t)rnd();
         const uint32_t p = (uint32_t)(i % n);
         uint32_t l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
         uint32_t cnt = 1 + rnd() % 5, len = 3 + rnd() % 60;
         if (rnd() % 8 == 0) len = 40 + rnd() % 200;
         for (uint32_t m = 0; m < cnt && len >= 3; m++) {
            uint32_t ll = len > n - p ? n - p : len;
            if (ll >= 3) l[m] = ll | ((1 + rnd() % 30000) << 16);
            len = len > 4 ? len - 1 - rnd() % 4 : 0;
         }
         rows[i] = {l[0], l[1], l[2], l[3]};
         rows[tot + i] = {l[4], l[5], l[6], l[7]};
      }
      uint4 *d_rows; uint8_t *d_win; uint32_t *d_best;
      hipMalloc(&d_rows, rows.size() * 16);
      hipMalloc(&d_win, tot);
      hipMalloc(&d_best, tot * 4);
      hipMemcpy(d_rows, rows.data(), rows.size() * 16, hipMemcpyHostToDevice);
      hipMemcpy(d_win, win.data(), tot, hipMemcpyHostToDevice);
      std::vector<uint64_t> o(2 * wgs);
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      for (int rep = 0; rep < 2; rep++) {
         hipDeviceSynchronize();
         if (noisy) hipLaunchKernelGGL(k_noise, dim3(48000), dim3(64), 0, s_noise, d_noise, 20000);
         hipEventRecord(e0, s_chain);
         hipLaunchKernelGGL(k_chain, dim3(wgs), dim3(256), 0, s_chain, (const uint4 *)d_rows, (const uint8_t *)d_win, n, d_best, d_out);
         hipEventRecord(e1, s_chain);
         hipEventSynchronize(e1);
         hipDeviceSynchronize();
      }
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(o.data(), d_out, o.size() * 8, hipMemcpyDeviceToHost);
      double sum = 0, mx = 0, rt = 0;
      for (int w = 0; w < wgs; w++) {
         sum += (double)o[2 * w];
         rt += (double)o[2 * w + 1];
         mx = mx > (double)o[2 * w] ? mx : (double)o[2 * w];
      }
      printf("   shader clock while the chains ran: %.2f GHz; a workgroup was inside its chain for %.3f ms on average\n", sum / rt * 0.1, rt / wgs * 1e-5);
      if (wgs == 1) {
         uint64_t prof[4];
         hipMemcpyFromSymbol(prof, HIP_SYMBOL(zh_chain_profile), sizeof(prof));
         printf("   busy cycles per position: consumer %.1f, stagers %.1f / %.1f, flusher %.1f\n", (double)prof[0] / n, (double)prof[1] / n, (double)prof[2] / n, (double)prof[3] / n);
      }
      if (noisy) printf("   (next line: one chain next to 48000 single-wave workgroups of LDS + VALU work on another stream)\n");
      printf("%4d chain workgroups x %u positions: kernel %.3f ms, %.1f cycles per position (mean), %.1f (slowest), %.3f us per position wall\n", wgs, n, ms,
             sum / wgs / n, mx / n, ms * 1e3 / n);
      hipFree(d_rows);
      hipFree(d_win);
      hipFree(d_best);
   }
   return 0;
}
