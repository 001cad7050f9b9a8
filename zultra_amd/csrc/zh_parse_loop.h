// zh_parse_loop.h — sub-blocks that go through their parse passes on their own (round 4).
//
// The reference runs the whole loop per sub-block (src/blockdeflate.c:860-921: optimal parse, histogram of the parse, rebuilt codes, four
// times). Rounds 1-3 ran it pass by pass over a whole run of max-blocks: zh_parse_lanes next to zh_parse_chain, then zh_sb_build, each a
// launch that waits for the one before — and every pass of a run lasted as long as its longest chain and its most loaded CU. Measured on
// the 100 MB step (tools/knob_sweep.py with a build that skips kernels): 24.5 ms without the parse kernels, 38.1 with them, although the
// quad kernel's work is 11 ms of the chip and the longest chain 1.3 ms per pass.
//
// Only sub-blocks with a chain task need that lock step (their chains are parsed by zh_parse_chain / zh_parse_segments, other kernels).
// All the others — nine in ten on source code — are taken through all four passes by ONE WORKGROUP of ZH_OWN_WAVES waves each, in one
// launch: every wave parses its share of the sub-block's tasks, eight at a time (zh_lp_group, exactly what a wave of zh_parse_lanes does),
// a workgroup barrier, wave 0 rebuilds the codes (zh_sb_build_one, what a wave of zh_sb_build does), a barrier, and the waves start over
// with the new prices; after the fourth pass the header is written. What the waves hand each other stays on their CU — a barrier, the
// CU's own L1 — so there is no queue, no counter in HBM and no fence.
// RESULT (MI355X, 100 MB of source code, tools/knob_sweep.py): 40.0-40.9 ms per step with this kernel against 37.5-38.3 without — whenever it
// starts (ZULTRA_HIP_OWN_AFTER) and at any stream priority. The step is not waiting for its dependencies, it is short of wave slots: a
// workgroup here holds four slots and 40 KB of LDS for ~4.5 ms, a third of it waiting at its own barriers, and a matchfinder workgroup of
// another run — which needs a CU to itself — waits for every one of them. OFF by default (ZULTRA_HIP_PARSE_LOOP=1 turns it on; the parity
// suite runs it once); kept as the measured answer to "let the sub-blocks advance on their own".
// (Two earlier builds of this file, measured on the same step: a work queue in HBM that shared a sub-block's tasks between
// any waves of the launch — correct, 78 ms: ~3 us of memory-side atomics per item on the queue's head with thousands of waves at it,
// and an agent-scope release per hand-off writes the XCD's dirty L2 lines back; one wave per sub-block — no hand-offs at all, 44 ms: a wave
// advances sixteen positions per ~1600 cycles whatever else the chip does, a 64 KiB sub-block took it 13 ms.)
#pragma once
#include <zh_platform.h>
#include "zh_common.h"
#include "zh_encode.h"
#include "zh_parse_lanes.h"

#ifndef ZH_OWN_WAVES
#define ZH_OWN_WAVES 4u
#endif
#ifndef ZH_OWN_OCC
#define ZH_OWN_OCC 3
#endif

// grid = the run's sub-blocks; WAVES = waves per workgroup (4 for max-blocks; 1 in files mode, where a sub-block is a 4 KiB input of two tasks)
// (waves per SIMD the compiler allocates registers for: at four — the quad kernel's own occupancy — it spills 24 registers; 40 KB of LDS per
// workgroup would let four workgroups onto a CU)
template <uint32_t WAVES>
__global__ void __launch_bounds__(64 * WAVES, ZH_OWN_OCC)
zh_parse_own(const uint8_t *__restrict__ data, const zh_block_t *__restrict__ blocks, const zh_match_t *__restrict__ match, uint64_t match_stride,
             const uint64_t *__restrict__ bars, uint64_t bar_stride, const zh_work_t *__restrict__ work, const uint2 *__restrict__ taskmap,
             zh_sbstate_t *states, uint32_t *best_all, uint64_t best_stride, uint16_t *cost_all, uint32_t *hist_part, const uint2 *__restrict__ taskinfo,
             uint8_t *payload, const uint32_t *__restrict__ sbflags, uint32_t nsubs, uint32_t *cnt /* the run's counters (ZH_CNT_*) */) {
   __shared__ union {
      zh_lp_ws_t lp[WAVES];   // a workspace per wave while they parse ...
      zh_sb_ws_t sb;                 // ... the code builder's between the passes
   } ws;
   __shared__ uint32_t stop;
   const uint32_t sb = blockIdx.x;
   if (sb >= nsubs || (sbflags[sb] & 1u)) return;   // (sub-blocks with a chain task: pass by pass, with the chain kernels)
   const zh_work_t wk = work[sb];
   zh_sbstate_t *st = states + sb;
   if (st->failed || !wk.ntasks) return;
   const bool dynamic = st->is_dynamic != 0;
   const uint32_t wave = threadIdx.x >> 6;
   // the wave's share of the tasks: contiguous, so that its groups of eight are the same groups a wave of zh_parse_lanes would form
   const uint32_t t_lo = wk.task_base + (uint32_t)((uint64_t)wk.ntasks * wave / WAVES), t_hi = wk.task_base + (uint32_t)((uint64_t)wk.ntasks * (wave + 1u) / WAVES);
   for (int pass = 0; pass <= 3; pass++) {
      for (uint32_t g = t_lo; g < t_hi; g += ZH_LP_TASKS)
         zh_lp_group(ws.lp[wave], g, min(t_hi, g + ZH_LP_TASKS), data, blocks, match, match_stride, bars, bar_stride, work, taskmap, states, best_all, best_stride, cost_all,
                     hist_part, pass, taskinfo);
      if (!dynamic) return;   // static sub-blocks are parsed once (blockdeflate.c:836-858)
      zh_stores_done();       // this wave's histograms (and its parse) are out ...
      __syncthreads();        // ... and so are the other waves'
      if (wave == 0) {
         const uint32_t failed = zh_sb_build_one(ws.sb, wk, st, hist_part, payload, pass, cnt);
         zh_stores_done();
         if (threadIdx.x == 0) stop = failed;
      }
      __syncthreads();
      if (stop) return;
   }
}
