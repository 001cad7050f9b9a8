// Diagnostics: which hardware queue (rocprofv3 Queue_Id) does the runtime give a stream? Streams are created, used and destroyed the way the library's contexts do it — one
// default stream, per run a non-blocking stream and a high-priority one — in several rounds; every launch carries (round, stream index) in its grid size.
// build: hipcc --offload-arch=gfx950 -O2 -o build/queue_probe tools/probes/queue_probe.hip
// run:   rocprofv3 --kernel-trace -d out -o kt --output-format csv -- build/queue_probe ; then list Queue_Id by Grid_Size (tools/r05_queue_probe.sh)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void tag_kernel(int *p) { if (p && threadIdx.x == 9999) *p = 1; }
struct set_t { hipStream_t s0, lane[4], side[4]; };
static set_t make_set() {
   set_t s;
   int lo = 0, hi = 0;
   hipDeviceGetStreamPriorityRange(&lo, &hi);
   hipStreamCreate(&s.s0);
   for (int k = 0; k < 4; k++) {
      hipStreamCreateWithFlags(&s.lane[k], hipStreamNonBlocking);
      hipStreamCreateWithPriority(&s.side[k], hipStreamNonBlocking, hi);
   }
   return s;
}
static void use_set(const set_t &s, int round) {   // grid = 64 * (round * 16 + index + 1): index 0 = s0, 1..4 lanes, 5..8 sides
   hipLaunchKernelGGL(tag_kernel, dim3(round * 16 + 1), dim3(64), 0, s.s0, (int *)nullptr);
   for (int k = 0; k < 3; k++) {   // three runs used, as the library does by default
      hipLaunchKernelGGL(tag_kernel, dim3(round * 16 + 2 + k), dim3(64), 0, s.lane[k], (int *)nullptr);
      hipLaunchKernelGGL(tag_kernel, dim3(round * 16 + 6 + k), dim3(64), 0, s.side[k], (int *)nullptr);
   }
   hipDeviceSynchronize();
}
static void kill_set(set_t &s) {
   hipStreamDestroy(s.s0);
   for (int k = 0; k < 4; k++) { hipStreamDestroy(s.lane[k]); hipStreamDestroy(s.side[k]); }
}
int main() {
   int round = 0;
   set_t a = make_set(); use_set(a, round++); use_set(a, round++); kill_set(a);          // rounds 0, 1: the first context, twice
   set_t b = make_set(); use_set(b, round++); kill_set(b);                                // round 2: a context behind it
   set_t c1 = make_set(), c2 = make_set(), c3 = make_set();                               // rounds 3, 4, 5: three at once
   use_set(c1, round++); use_set(c2, round++); use_set(c3, round++);
   kill_set(c1); kill_set(c2); kill_set(c3);
   set_t d = make_set(); use_set(d, round++); kill_set(d);                                // round 6: a context behind those
   set_t e = make_set(); use_set(e, round++);                                             // round 7: and another, kept
   set_t f = make_set(); use_set(f, round++); use_set(e, round++);                         // rounds 8, 9: two alive
   printf("rounds %d\n", round);
   return 0;
}
