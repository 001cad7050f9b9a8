// zh_encode.h — stage 3 of the hot path around the parse kernel (zh_parse.h): everything that turns a sub-block into
// its deflate bit string.
//
// Replaces, per sub-block, the calls at reference src/libzultra.c:317-324 (static vs dynamic decision:
// zultra_block_prepare_cost_evaluation, zultra_block_evaluate_static_cost, 2x estimate_dynamic_codelens,
// zultra_block_evaluate_dynamic_cost) and :343 zultra_block_deflate (src/blockdeflate.c:827-997): tentative Huffman
// codes from the greedy parse, four optimal-parse passes each followed by a histogram of the chosen parse and a rebuild
// of the codes, literalisation of cheap matches, the RLE-friendly alternative tables, the 20-way search over
// code-length RLE masks, header and token emission.
//
// The reference runs these steps one sub-block at a time. Here each step is one kernel over the whole batch, and the
// per-sub-block coder state (zh_sbstate_t: code lengths, codes, flags) lives in HBM between them:
//
//   zh_sb_init      wave per sub-block   greedy histogram -> static / dynamic price -> type; tentative (or fixed) codes
//   4 x { zh_parse_lanes   wave per group of tasks, zh_parse_chain next to it: optimal parse + histogram of the parse   (zh_parse_lanes.h, zh_parse_chain.h)
//         zh_sb_build      wave per sub-block   sum the task histograms, rebuild both codes; after the 4th pass also the
//                                               RLE-friendly alternative, the 20 code-length masks and the block header }
//   zh_post_tasks   wave per task        literalisation (blockdeflate.c:410-458) and the bit count of the task's tokens
//   zh_emit_tasks   wave per task        tokens -> bits at the task's bit offset inside the sub-block's slot
//
// Kernels "per task" walk the parse forward from a barrier (zh_parse.h): the token chain of a 64-position tile is
// followed on the scalar unit (v_readlane over one VGPR), the token lanes then work in parallel; emission gets its bit
// offsets from a wave prefix sum and ORs 48-bit token codes into an LDS staging window that is flushed to HBM in whole
// dwords (the dwords shared with the neighbouring task by atomic OR into the zero-filled slot).
// Kernels "per sub-block" are short and latency-bound: lane-parallel rank sort, serial Moffat-Katajainen merge on one
// lane, 20 lanes pricing the 20 RLE masks at once with private 19-symbol encoders.
#pragma once
#include <zh_platform.h>
#include "zh_common.h"
#include "zh_huffman.h"
#include "zh_split.h"
#include "zh_parse.h"

#define ZH_OBUF_WORDS 128   // LDS staging window for token bits: 64 tokens x 48 bits = 96 dwords + carry

// ---- work-list planning: the splitter's token boundaries become sub-block work items --------------------------------------------------
// (libzultra.c:303-314: nBlockSize = nSplitOffset[k] - (nInStart + prev)). Sub-block k of block b gets the payload slot starting at
// (block's slot base) + (offset of the sub-block in the block) + 64*k, so slots never overlap and every slot can hold size+8 bytes.
// ONE workgroup per run, a thread per max-block: the sub-block indices (stream order) and the task ranges are exclusive prefix sums over the
// run's max-blocks, taken here — the reference simply loops (libzultra.c:303-324); rounds 1-4 read the splitter's counts back to the host,
// summed them there and uploaded the bases: two host round trips in the middle of every run. The run's totals go to its counters
// (cnt[ZH_CNT_NSUBS], cnt[ZH_CNT_TASKS]): every later kernel takes its bounds from there, the host sizes grids from the input bytes alone.
// (THREADS: 256 for a run of a few thousand max-blocks — four waves of a dozen registers find room on a CU that a matchfinder workgroup of another run
// fills, which a 1024-thread workgroup does not: it waited 2.5 ms for one on the 100 MB step; 1024 for the tens of thousands of inputs of a files batch)
template <uint32_t ZH_PLAN_THREADS>
__global__ void __launch_bounds__(ZH_PLAN_THREADS)
zh_plan_subblocks(const zh_block_t *__restrict__ blocks, uint32_t nblocks, const uint32_t *__restrict__ tok_pos, uint64_t tok_stride, const uint32_t *__restrict__ ntok,
                  const uint32_t *__restrict__ split_tok, const uint32_t *__restrict__ split_cnt, uint32_t *sub_base /* out: exclusive scan of split_cnt */,
                  uint64_t slot_stride, zh_work_t *work, uint2 *taskmap, uint32_t *cnt /* the run's counters */) {
   __shared__ uint32_t wsum[2][ZH_PLAN_THREADS / 64];
   __shared__ uint32_t carry[2];
   const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
   if (tid < 2) carry[tid] = 0;
   __syncthreads();
   for (uint32_t b0 = 0; b0 < nblocks; b0 += ZH_PLAN_THREADS) {
      const uint32_t b = b0 + tid;
      const bool on = b < nblocks;
      zh_block_t blk = {0, 0, 0};
      uint32_t n_sub = 0, n_task = 0, nt = 0;
      const uint32_t *tp = tok_pos, *st = split_tok;
      if (on) {
         blk = blocks[b];
         tp = tok_pos + (uint64_t)b * tok_stride;
         st = split_tok + (uint64_t)b * (ZH_MAX_SPLITS + 1);
         n_sub = split_cnt[b];
         nt = ntok[b];
         uint32_t p0 = (st[0] < nt) ? tp[st[0]] : blk.prev + blk.n;
         for (uint32_t k = 0; k < n_sub; k++) {
            const uint32_t t1 = st[k + 1];
            const uint32_t p1 = (t1 < nt) ? tp[t1] : blk.prev + blk.n;
            n_task += (p1 - p0 + ZH_TASK - 1) / ZH_TASK;
            p0 = p1;
         }
      }
      // exclusive prefix over the workgroup's max-blocks, of the sub-blocks and of the tasks
      const uint32_t xs = zh_wave_excl_sum(n_sub), xt = zh_wave_excl_sum(n_task);
      if (lane == 63) {
         wsum[0][wave] = xs + n_sub;
         wsum[1][wave] = xt + n_task;
      }
      __syncthreads();
      uint32_t sub0 = carry[0] + xs, task0 = carry[1] + xt;
      for (uint32_t w = 0; w < wave; w++) {
         sub0 += wsum[0][w];
         task0 += wsum[1][w];
      }
      __syncthreads();
      if (tid == ZH_PLAN_THREADS - 1) {
         carry[0] = sub0 + n_sub;
         carry[1] = task0 + n_task;
      }
      if (on) {
         sub_base[b] = sub0;
         uint32_t p0 = (st[0] < nt) ? tp[st[0]] : blk.prev + blk.n;
         for (uint32_t k = 0; k < n_sub; k++) {
            const uint32_t t0 = st[k], t1 = st[k + 1];
            const uint32_t p1 = (t1 < nt) ? tp[t1] : blk.prev + blk.n;
            zh_work_t w;
            w.block = b;
            w.start = p0;
            w.size = p1 - p0;
            w.tok0 = t0;
            w.tok1 = t1;
            const uint32_t rel = p0 - blk.prev;
            w.out_off = (uint64_t)b * slot_stride + ((rel + 3u) & ~3u) + 64u * k;
            w.out_cap = ((p1 - p0) + 8u + 3u) & ~3u;
            w.index = sub0 + k;
            w.ntasks = (w.size + ZH_TASK - 1) / ZH_TASK;
            w.task_base = task0;
            w.pad = 0;
            for (uint32_t j = 0; j < w.ntasks; j++) {
               uint2 e;
               e.x = w.index;
               e.y = j;
               taskmap[w.task_base + j] = e;
            }
            work[w.index] = w;
            task0 += w.ntasks;
            p0 = p1;
         }
      }
      __syncthreads();   // the carries are in
   }
   if (tid == 0) {
      cnt[ZH_CNT_NSUBS] = carry[0];
      cnt[ZH_CNT_TASKS] = carry[1];
   }
}

// Files mode: an input below 8192 bytes is never split (blockdeflate.c:646), so a batch of inputs has one sub-block per input by construction and the task ranges
// follow from the inputs' sizes alone — which the host has (it was handed them): task_prefix = the exclusive prefix of ceil(size / ZH_TASK) over the batch's inputs,
// uploaded with the block list. A thread per input, any number of workgroups. (Rounds 4-5 ran zh_plan_subblocks<1024> here: ONE workgroup looping over 32 768 inputs
// with five dependent loads each — 2.9 ms per launch on the 65 536-input batches of configuration 5, 5 ms next to another run's matchfinder, on the path to the run's
// first parse pass.) task_prefix points at the run's first input; sub-block and task indices are local to the run.
#define ZH_PLAN_FILES_THREADS 256
__global__ void __launch_bounds__(ZH_PLAN_FILES_THREADS)
zh_plan_files(const zh_block_t *__restrict__ blocks, uint32_t nblocks, const uint32_t *__restrict__ ntok, const uint32_t *__restrict__ task_prefix, uint32_t *sub_base,
              uint64_t slot_stride, zh_work_t *work, uint2 *taskmap, uint32_t *cnt /* the run's counters */) {
   const uint32_t b = blockIdx.x * ZH_PLAN_FILES_THREADS + threadIdx.x;
   const uint32_t base = task_prefix[0];
   if (b == 0) {
      cnt[ZH_CNT_NSUBS] = nblocks;
      cnt[ZH_CNT_TASKS] = task_prefix[nblocks] - base;
   }
   if (b >= nblocks) return;
   const zh_block_t blk = blocks[b];
   zh_work_t w;
   w.block = b;
   w.start = blk.prev;          // the first token sits on the input's first byte
   w.size = blk.n;
   w.tok0 = 0;
   w.tok1 = ntok[b];
   w.out_off = (uint64_t)b * slot_stride;
   w.out_cap = (blk.n + 8u + 3u) & ~3u;
   w.index = b;
   w.ntasks = (blk.n + ZH_TASK - 1) / ZH_TASK;
   w.task_base = task_prefix[b] - base;
   w.pad = 0;
   for (uint32_t j = 0; j < w.ntasks; j++) taskmap[w.task_base + j] = make_uint2(b, j);
   work[b] = w;
   sub_base[b] = b;
}

// ---- LDS workspace of the per-sub-block kernels -----------------------------------------------------------------
// 8 KB: twenty one-wave workgroups of zh_sb_init / zh_sb_build per CU. The kernels are chains of dependent LDS round trips on one wave (or one
// lane), their throughput is the number of them a CU holds; with every array laid side by side (15 KB) it held ten. What is never live at the same
// time shares its bytes: the histograms, the alternative's tables and the sort scratch are dead once the alternative is decided, which is when the
// twenty header candidates start.
struct zh_sb_ws_t {
   uint8_t lit_len[ZH_NLIT], dist_len[ZH_NDIST];
   uint16_t lit_code[ZH_NLIT], dist_code[ZH_NDIST];
   uint8_t lens[ZH_NLIT + ZH_NDIST];
   zh_cl_t cl;
   uint16_t runs[ZH_NLIT + ZH_NDIST];   // run list of the header's code lengths, shared by the 20 mask candidates
   int32_t tmp;
   union {
      struct {
         int32_t lit_freq[ZH_NLIT], dist_freq[ZH_NDIST];   // after pass 3: smoothed in place for the RLE-friendly alternative (:925-945)
         uint8_t alt_lit_len[ZH_NLIT], alt_dist_len[ZH_NDIST];
         uint16_t alt_lit_code[ZH_NLIT], alt_dist_code[ZH_NDIST];
         zh_huff_scratch_t sc;                              // (sc.sorted doubles as the smoothing's `keep` flags: no build is in flight then)
      };
      zh_cl_t cl_work[20];                                  // the header candidates (:947-992)
   };
};
static_assert(ZH_OBUF_WORDS * 4 >= 256, "zh_post_tasks stages 256 bytes of the task in the emission window");
static_assert(sizeof(zh_sb_ws_t) <= 8192, "twenty workgroups of zh_sb_build per CU (160 KB of LDS)");

__device__ inline void zh_store_codes_wave(zh_sbstate_t *st, const zh_sb_ws_t *ws) {
   const uint32_t lane = zh_lane();
   for (uint32_t s = lane; s < ZH_NLIT; s += 64) {
      st->lit_len[s] = ws->lit_len[s];
      st->lit_code[s] = ws->lit_code[s];
   }
   if (lane < ZH_NDIST) {
      st->dist_len[lane] = ws->dist_len[lane];
      st->dist_code[lane] = ws->dist_code[lane];
   }
}

// ---- zh_sb_init: libzultra.c:317-324 and the start of zultra_block_deflate (blockdeflate.c:832-868) -------------
__device__ __forceinline__ void zh_sb_init_one(zh_sb_ws_t &ws, uint32_t sb, const uint16_t *__restrict__ tok_info, uint64_t tok_stride, const zh_work_t *__restrict__ work, zh_sbstate_t *states) {
   const zh_work_t wk = work[sb];
   zh_sbstate_t *st = states + sb;
   const uint16_t *ti = tok_info + (uint64_t)wk.block * tok_stride;
   const uint32_t lane = zh_lane();

   for (uint32_t s = lane; s < ZH_NLIT; s += 64) ws.lit_freq[s] = 0;
   if (lane < ZH_NDIST) ws.dist_freq[lane] = 0;
   zh_sync();
   zh_token_histogram_wave(ti, wk.tok0, wk.tok1, ws.lit_freq, ws.dist_freq);
   if (lane == 0) ws.lit_freq[ZH_EOB] += 1;
   zh_sync();

   uint32_t sc_part = 0;   // blockdeflate.c:538-566
   for (uint32_t s = lane; s < 257 + 29; s += 64) {
      int xb = (s >= 257) ? zh_lenidx_xbits((int)s - 257) : 0;
      sc_part += (uint32_t)(ws.lit_freq[s] * (zh_static_lit_len((int)s) + xb));
   }
   if (lane < ZH_NDIST) sc_part += (uint32_t)(ws.dist_freq[lane] * (5 + zh_dist_xbits((int)lane)));
   const int static_cost = (int)zh_wave_sum(sc_part) + 3;
   const int dynamic_cost = zh_dynamic_cost_wave(ws.lit_freq, ws.dist_freq, ws.lit_len, ws.dist_len, ws.lens, &ws.cl, &ws.tmp,
                                                 &ws.sc, true);
   const uint32_t is_dynamic = (static_cost <= dynamic_cost) ? 0u : 1u;
   uint32_t failed = 0;

   if (!is_dynamic) {
      // blockdeflate.c:836-858: fixed lengths
      for (uint32_t s = lane; s < ZH_NLIT; s += 64) ws.lit_len[s] = (uint8_t)zh_static_lit_len((int)s);
      if (lane < ZH_NDIST) ws.dist_len[lane] = 5;
      zh_sync();
      zh_huff_static_codes_wave(ws.lit_len, ws.lit_code, ZH_NLIT, &ws.sc);
      zh_huff_static_codes_wave(ws.dist_len, ws.dist_code, ZH_NDIST, &ws.sc);
   }
   else {
      // blockdeflate.c:859-868: tentative codes from the greedy histogram
      if (zh_huff_build_wave(ws.lit_freq, ws.lit_len, ws.lit_code, ZH_NLIT, 15, &ws.sc) < 0) failed = 1;
      if (zh_huff_build_wave(ws.dist_freq, ws.dist_len, ws.dist_code, ZH_NDIST, 15, &ws.sc) < 0) failed = 1;
   }
   zh_store_codes_wave(st, &ws);
   for (uint32_t s = lane; s < ZH_NLIT; s += 64) st->pre_lit_len[s] = ws.lit_len[s];
   if (lane < ZH_NDIST) st->pre_dist_len[lane] = ws.dist_len[lane];
   if (lane == 0) {
      st->is_dynamic = is_dynamic;
      st->failed = failed;
      st->settled = 0;
      st->hdr_bits = 0;
      st->static_cost = static_cost;
      st->dynamic_cost = dynamic_cost;
   }
}

// One wave per sub-block. How many sub-blocks the splitter made is known on the device only (cnt[ZH_CNT_NSUBS], zh_plan_subblocks): the host launches
// the <false> form over a grid of what data usually gives (a few per max-block; surplus workgroups leave at once) and, behind it, the <true> form —
// a few workgroups that stride over whatever lies beyond that grid (`first`), i.e. nearly always over nothing. (One kernel that strides over
// everything keeps three times the registers: the early exits of a sub-block become branches inside a loop.)
template <bool MORE>
__global__ void __launch_bounds__(64)
zh_sb_init(const uint16_t *__restrict__ tok_info, uint64_t tok_stride, const zh_work_t *__restrict__ work, zh_sbstate_t *states, const uint32_t *__restrict__ cnt /* the run's counters */,
           uint32_t first) {
   __shared__ zh_sb_ws_t ws;
   const uint32_t nsubs = cnt[ZH_CNT_NSUBS];
   if (!MORE) {
      if (blockIdx.x < nsubs) zh_sb_init_one(ws, blockIdx.x, tok_info, tok_stride, work, states);
      return;
   }
   for (uint32_t sb = first + blockIdx.x; sb < nsubs; sb += gridDim.x) {
      zh_sync();   // the sub-block before this one is done with the workspace
      zh_sb_init_one(ws, sb, tok_info, tok_stride, work, states);
   }
}

// ---- bit output of the block header (single lane) -------------------------------------------------------------------
struct zh_bitw_t {
   uint32_t *out;        // dword-aligned slot in HBM
   uint32_t cap_bits;
   uint64_t acc;
   uint32_t nacc;
   uint32_t nbits;       // total so far
   __device__ __forceinline__ void put(uint32_t v, uint32_t n) {
      acc |= (uint64_t)v << nacc;
      nacc += n;
      nbits += n;
      if (nacc >= 32) {
         if (nbits - nacc + 32 <= cap_bits) out[(nbits - nacc) >> 5] = (uint32_t)acc;
         acc >>= 32;
         nacc -= 32;
      }
   }
};
struct zh_cl_write_sink {
   const zh_cl_t *h;
   zh_bitw_t *w;
   __device__ __forceinline__ void put(int sym, int xval, int xbits) {
      w->put(h->code[sym], h->len[sym]);
      if (xbits) w->put((uint32_t)xval, (uint32_t)xbits);
   }
};

// ---- zh_sb_build: blockdeflate.c:887-919 for pass 0..3; after pass 3 also :925-992 -----------------------------------
// One sub-block by one wave (all 64 lanes call; the syncs are the wave's: it may be one of several of a workgroup); `ws` is the
// wave's own workspace in LDS. Returns (in every lane) whether the sub-block has failed.
__device__ __forceinline__ uint32_t zh_sb_build_one(zh_sb_ws_t &ws, const zh_work_t wk, zh_sbstate_t *st, const uint32_t *__restrict__ hist_part, uint8_t *payload, int pass, uint32_t *cnt /* the run's counters */) {
   if (st->failed) return 1u;
   if (!st->is_dynamic) return 0u;
   const uint32_t lane = zh_lane();
   uint32_t failed = 0;

   // histogram of the parse = sum over the sub-block's tasks, +1 end-of-block (blockdeflate.c:371-400)
   for (uint32_t s = lane; s < ZH_NSYM; s += 64) {
      uint32_t v = 0;
      for (uint32_t j = 0; j < wk.ntasks; j++) v += hist_part[(uint64_t)(wk.task_base + j) * ZH_NSYM + s];
      if (s < ZH_NLIT)
         ws.lit_freq[s] = (int32_t)v + (s == ZH_EOB ? 1 : 0);
      else
         ws.dist_freq[s - ZH_NLIT] = (int32_t)v;
   }
   zh_wave_sync();
   if (pass == 3 && lane == 0) {   // at least two distance codes among 0..29 (:893-913)
      int used = 0;
      for (int s = 0; used < 2 && s < ZH_NDIST - 2; s++)
         if (ws.dist_freq[s]) used++;
      if (used == 0)
         ws.dist_freq[0] = ws.dist_freq[1] = 1;
      else if (used == 1) {
         if (ws.dist_freq[0])
            ws.dist_freq[1] = 1;
         else
            ws.dist_freq[0] = 1;
      }
   }
   zh_wave_sync();
   if (zh_huff_build_wave(ws.lit_freq, ws.lit_len, ws.lit_code, ZH_NLIT, 15, &ws.sc) < 0) failed = 1;
   if (zh_huff_build_wave(ws.dist_freq, ws.dist_len, ws.dist_code, ZH_NDIST, 15, &ws.sc) < 0) failed = 1;

   if (pass < 3) {
      // A fixed point of the loop at blockdeflate.c:874-901: the next pass prices with these lengths, unused symbols at 9 / 6 bits (:875-878). If
      // that is what this pass priced with, the next parse is this one again (the parse is a function of the rows and the prices), so is
      // its histogram, so are the lengths built from it — for every pass left. The parse kernels then skip the sub-block (st->settled) and
      // leave its parse entries and task histograms alone; the builds still run, on the same sums (pass 3 adds the distance-code fix,
      // the alternative tables and the header).
      uint32_t moved = 0;
      for (uint32_t s = lane; s < ZH_NLIT; s += 64) moved |= (uint32_t)((ws.lit_len[s] ? ws.lit_len[s] : 9) != (st->lit_len[s] ? st->lit_len[s] : 9));
      if (lane < ZH_NDIST) moved |= (uint32_t)((ws.dist_len[lane] ? ws.dist_len[lane] : 6) != (st->dist_len[lane] ? st->dist_len[lane] : 6));
      moved = zh_wave_sum(moved);
      zh_store_codes_wave(st, &ws);
      if (lane == 0) {
         if (failed) st->failed = 1;
         if (!moved && !failed && !st->settled) {
            st->settled = 1;
            atomicAdd(&cnt[ZH_CNT_SETTLED], (uint32_t)(3 - pass));
            atomicAdd(&cnt[ZH_CNT_SETTLED_POS], (uint32_t)(3 - pass) * ((wk.size + 512u) >> 10));
         }
      }
      return failed;
   }

   // the prices literalisation will use: the codes of the last pass (:923) (read by later kernels only: plain stores, like the header below)
   for (uint32_t s = lane; s < ZH_NLIT; s += 64) st->pre_lit_len[s] = ws.lit_len[s];
   if (lane < ZH_NDIST) st->pre_dist_len[lane] = ws.dist_len[lane];

   // ---- blockdeflate.c:925-945: RLE-friendlier alternative (histograms are those of the last parse) -----------------
   {
      const int cur_cost = zh_dynamic_cost_wave(ws.lit_freq, ws.dist_freq, ws.lit_len, ws.dist_len, ws.lens, &ws.cl, &ws.tmp,
                                                &ws.sc, false);
      // (the histograms of the last parse have priced the codes in force and are not read again: smoothed where they are)
      zh_wave_sync();
#ifndef ZH_DBG_SKIP_SMOOTH   // (timing experiments with wrong output, tools/build_variant.sh)
      if (lane == 0) {
         zh_smooth_for_rle_lane(ZH_NLIT, ws.lit_freq, (uint8_t *)ws.sc.sorted);
         zh_smooth_for_rle_lane(ZH_NDIST, ws.dist_freq, (uint8_t *)ws.sc.sorted);
      }
#endif
      zh_wave_sync();
      if (zh_huff_build_wave(ws.lit_freq, ws.alt_lit_len, ws.alt_lit_code, ZH_NLIT, 15, &ws.sc) < 0) failed = 1;
      if (zh_huff_build_wave(ws.dist_freq, ws.alt_dist_len, ws.alt_dist_code, ZH_NDIST, 15, &ws.sc) < 0) failed = 1;
      const int alt_cost = zh_dynamic_cost_wave(ws.lit_freq, ws.dist_freq, ws.alt_lit_len, ws.alt_dist_len, ws.lens,
                                                &ws.cl, &ws.tmp, &ws.sc, false);
      if (alt_cost < cur_cost) {
         for (uint32_t s = lane; s < ZH_NLIT; s += 64) {
            ws.lit_len[s] = ws.alt_lit_len[s];
            ws.lit_code[s] = ws.alt_lit_code[s];
         }
         if (lane < ZH_NDIST) {
            ws.dist_len[lane] = ws.alt_dist_len[lane];
            ws.dist_code[lane] = ws.alt_dist_code[lane];
         }
      }
      zh_wave_sync();
   }

   // ---- blockdeflate.c:947-992: header ---------------------------------------------------------------------------------
   const int nlit = zh_defined_count(ws.lit_len, ZH_NLIT, 257);
   const int ndist = zh_defined_count(ws.dist_len, ZH_NDIST, 1);
   for (int s = (int)lane; s < nlit; s += 64) ws.lens[s] = ws.lit_len[s];
   if ((int)lane < ndist) ws.lens[nlit + (int)lane] = ws.dist_len[lane];
   zh_wave_sync();

   const int nruns = zh_cl_make_runs_wave(ws.lens, nlit + ndist, ws.runs, ws.sc.keys);
   uint32_t mkey = 0xFFFFFFFFu;
#ifdef ZH_DBG_ONE_MASK
   if (lane < 1) {
#else
   if (lane < 20) {
#endif
      // 20 lanes, one mask each, all walking the same run list: the loop trip counts agree, only the token choice differs
      const unsigned mask = lane < 8 ? lane : 9 + 2 * (lane - 8);   // 0..7, 9, 11, ..., 31 (:959)
      zh_cl_t *h = &ws.cl_work[lane];
      zh_cl_reset(h);
      zh_cl_count_sink cs{h};
      zh_cl_tokenize_runs(ws.runs, nruns, mask, cs);
      if (zh_cl_build_lane(h, 7) < 0)
         mkey = 0xFFFFFFFEu;
      else {
         zh_cl_size_sink ss{h, 0};
         zh_cl_tokenize_runs(ws.runs, nruns, mask, ss);
         mkey = ((uint32_t)ss.bits << 6) | (63u - lane);   // cheapest; among equals the last tried (:966)
      }
   }
   zh_wave_sync();
   const uint32_t mbest = zh_wave_min(mkey);
   const uint32_t anybad = zh_wave_sum(mkey == 0xFFFFFFFEu ? 1u : 0u);
   if (anybad) failed = 1;
   const uint32_t best_lane = 63u - (mbest & 63u);
   const unsigned best_mask = best_lane < 8 ? best_lane : 9 + 2 * (best_lane - 8);

   if (lane == 0) {
      const zh_cl_t *h = &ws.cl_work[best_lane < 20 ? best_lane : 0];
      const int ncl = zh_cl_raw_table_size(h);
      if (nlit > 286 || ndist > 30 || ncl > ZH_NCL) ws.tmp = -1;
      else {
         uint32_t *out = (uint32_t *)(payload + wk.out_off);
         zh_bitw_t w{out, wk.out_cap * 8, 0, 0, 0};
         w.put((uint32_t)(nlit - 257), 5);
         w.put((uint32_t)(ndist - 1), 5);
         w.put((uint32_t)(ncl - 4), 4);
         for (int k = 0; k < ncl; k++) w.put(h->len[zh_cl_order(k)], 3);
         zh_cl_write_sink sink{h, &w};
         zh_cl_tokenize_runs(ws.runs, nruns, best_mask, sink);
         // the partial dword: the slot is zero-filled and the first task ORs its bits in later
         if (w.nacc && w.nbits - w.nacc + 32 <= w.cap_bits) out[(w.nbits - w.nacc) >> 5] = (uint32_t)w.acc;
         ws.tmp = (int32_t)w.nbits;
      }
   }
   zh_wave_sync();
   if (ws.tmp < 0) failed = 1;
   zh_store_codes_wave(st, &ws);
   if (lane == 0) {
      st->hdr_bits = ws.tmp < 0 ? 0u : (uint32_t)ws.tmp;
      if (failed) st->failed = 1;
   }
   return failed;
}

// (one wave per sub-block; <false> over the usual grid, <true> strides over what lies beyond it: see zh_sb_init)
template <bool MORE>
__global__ void __launch_bounds__(64)
zh_sb_build(const zh_work_t *__restrict__ work, zh_sbstate_t *states, const uint32_t *__restrict__ hist_part, uint8_t *payload, int pass, uint32_t *cnt, uint32_t first) {
   __shared__ zh_sb_ws_t ws;
   if (zh_run_is_void(cnt)) return;
   const uint32_t nsubs = cnt[ZH_CNT_NSUBS];
   if (!MORE) {
      if (blockIdx.x < nsubs) (void)zh_sb_build_one(ws, work[blockIdx.x], states + blockIdx.x, hist_part, payload, pass, cnt);
      return;
   }
   for (uint32_t sb = first + blockIdx.x; sb < nsubs; sb += gridDim.x) {
      zh_sync();   // the sub-block before this one is done with the workspace
      (void)zh_sb_build_one(ws, work[sb], states + sb, hist_part, payload, pass, cnt);
   }
}

// ---- prices / sizes of the codes in a sub-block state, staged in LDS by the task kernels -------------------------
struct zh_task_ws_t {
   uint8_t lit_len[ZH_NLIT], dist_len[ZH_NDIST];
   uint16_t lit_code[ZH_NLIT], dist_code[ZH_NDIST];
   uint8_t pre_lit_len[ZH_NLIT];
   uint8_t lencost[256];       // pre-alternative prices of lengths / distance symbols (literalisation)
   uint8_t distcost[ZH_NDIST];
   uint32_t obuf[ZH_OBUF_WORDS];
   int32_t tmp;
};

// ---- zh_post_tasks: matches that are cheaper as literals (blockdeflate.c:410-458), then the bit count of the task --
__device__ __forceinline__ void zh_post_task_one(zh_task_ws_t &ws, uint32_t gt, const uint8_t *__restrict__ data, const zh_block_t *__restrict__ blocks, const uint64_t *__restrict__ bars,
                                                 uint64_t bar_stride, const zh_work_t *__restrict__ work, const uint2 *__restrict__ taskmap, const zh_sbstate_t *__restrict__ states,
                                                 uint32_t *best_all, uint64_t best_stride, uint32_t *task_bits, const uint2 *__restrict__ taskinfo) {
   const uint2 tm = taskmap[gt];
   const zh_work_t wk = work[tm.x];
   const zh_sbstate_t *st = states + tm.x;
   if (st->failed) return;
   const zh_block_t blk = blocks[wk.block];
   const uint8_t *win = data + blk.win_off;
   const uint32_t prev = blk.prev;
   const uint64_t *bar = bars + (uint64_t)wk.block * bar_stride;
   uint32_t *best = best_all + (uint64_t)wk.block * best_stride;
   const uint32_t lane = zh_lane();
   const uint32_t t0 = taskinfo[gt].x, t1 = taskinfo[gt].y & 0x7fffffffu;   // the task's range (zh_list_huge)
   (void)bar;
   const bool dynamic = st->is_dynamic != 0;

   for (uint32_t s = lane; s < ZH_NLIT; s += 64) {
      ws.lit_len[s] = st->lit_len[s];
      ws.pre_lit_len[s] = st->pre_lit_len[s];
   }
   if (lane < ZH_NDIST) {
      ws.dist_len[lane] = st->dist_len[lane];
      ws.distcost[lane] = (uint8_t)(st->pre_dist_len[lane] + zh_dist_xbits((int)lane));
   }
   zh_sync();
   for (uint32_t e = lane; e < 256; e += 64) {
      const int idx = zh_len_idx(e + 3);
      ws.lencost[e] = (uint8_t)(ws.pre_lit_len[257 + idx] + zh_lenidx_xbits(idx));
   }
   zh_sync();

   if (dynamic) {   // the static path has no post-optimisation (blockdeflate.c:836-858)
      // (a task can be tens of thousands of positions — a whole chain — and a tile's work hangs on one global load: the next tile's is
      // issued before this tile is worked on. A tile's literalisations may reach into the tiles loaded ahead, but only at positions the
      // chain jumps over: what the walk reads of a position that is a token start is never changed under it.)
      // (round 6) the bytes a match is priced against as literals come from LDS: the task's bytes are staged tile by tile, two tiles ahead, in a 256-byte ring (the
      // emission window of zh_emit_tasks, idle here) — a match is literalised or kept within its first few bytes (its price is a few dozen bits at most), and every
      // one of those was a dependent load from global memory in a loop that a whole tile waits for: ~2800 cycles per tile, and one wave walks a whole task (a chain
      // task: tens of thousands of positions — this kernel lasted as long as its longest task, 1.0 ms for one run alone). Bytes beyond the staged tiles: from memory.
      uint8_t *stage = (uint8_t *)ws.obuf;
      uint32_t carry = 0;
      uint32_t b_next = (t0 + lane < t1) ? best[t0 + lane - prev] : 0;
      uint32_t y_next = (t0 + 64 + lane < t1) ? (uint32_t)win[t0 + 64 + lane] : 0u;   // the bytes of the tile after the next one loaded
      if (t0 + lane < t1) stage[lane] = win[t0 + lane];
      for (uint32_t base = t0; base < t1; base += 64) {
         const uint32_t limit = min(64u, t1 - base);
         const uint32_t pos = base + lane;
         const uint32_t b = b_next;
         b_next = (pos + 64 < t1) ? best[pos + 64 - prev] : 0;
         stage[(pos + 64 - t0) & 255u] = (uint8_t)y_next;   // tile base + 64 .. base + 127 (the ring holds base - 64 .. base + 191)
         y_next = (pos + 128 < t1) ? (uint32_t)win[pos + 128] : 0u;
         zh_sync();
         const uint32_t staged_end = min(t1, base + 128u);
         const uint32_t len = b & 0xffffu;
         const uint64_t mask = zh_chain_mask(len, carry, limit);
         if (((mask >> lane) & 1ull) && len >= ZH_MIN_MATCH) {
            const uint32_t off = b >> 16;
            if (off >= 1 && off <= ZH_MAX_DIST) {
               const uint32_t mcost = (uint32_t)ws.lencost[min(len - ZH_MIN_MATCH, 255u)] + (uint32_t)ws.distcost[zh_dist_sym(off)];
               uint32_t lcost = 0, j = 0;
               bool usable = true;
               for (; j < len && lcost < mcost; j++) {
                  const uint32_t pj = pos + j;
                  const uint32_t l = ws.pre_lit_len[pj < staged_end ? (uint32_t)stage[(pj - t0) & 255u] : (uint32_t)win[pj]];
                  if (l == 0) {
                     usable = false;   // a byte without a code keeps the match (:436-440)
                     break;
                  }
                  lcost += l;
               }
               if (usable && lcost < mcost)
                  for (j = 0; j < len; j++) best[pos - prev + j] &= 0xffff0000u;   // length := 0 (:449-451)
            }
         }
         zh_sync();   // (the next tile's bytes go over the ring's oldest)
      }
      __threadfence_block();
      zh_sync();
   }

   // bits of the task's tokens under the final codes
   uint32_t bits = 0, carry = 0;
   if (t0 < t1) ZH_WALK_BEGIN(best, win, prev, t0, t1)   // (tiles in flight tracked by hand, zh_parse.h: this walk issues no vector-memory operation of its own)
      const uint32_t len = b & 0xffffu;
      const uint64_t mask = zh_chain_mask(len, carry, limit);
      if ((mask >> lane) & 1ull) {
         if (len >= ZH_MIN_MATCH) {
            const int li = zh_len_idx(len), ds = zh_dist_sym(b >> 16);
            bits += (uint32_t)ws.lit_len[257 + li] + (uint32_t)zh_lenidx_xbits(li) + (uint32_t)ws.dist_len[ds] + (uint32_t)zh_dist_xbits(ds);
         }
         else
            bits += ws.lit_len[byte];
      }
   ZH_WALK_END
   bits = zh_wave_sum(bits);
   if (lane == 0) task_bits[gt] = bits;
}

// One wave per task; the length of the run's task list is known on the device only: <false> over a grid of what data usually gives (surplus
// workgroups leave at once), <true> a few workgroups that stride over what lies beyond it (see zh_sb_init).
template <bool MORE>
__global__ void __launch_bounds__(64)
zh_post_tasks(const uint8_t *__restrict__ data, const zh_block_t *__restrict__ blocks, const uint64_t *__restrict__ bars, uint64_t bar_stride,
              const zh_work_t *__restrict__ work, const uint2 *__restrict__ taskmap, const uint32_t *__restrict__ ntasks_total,
              const zh_sbstate_t *__restrict__ states, uint32_t *best_all, uint64_t best_stride, uint32_t *task_bits, const uint2 *__restrict__ taskinfo, uint32_t first) {
   __shared__ zh_task_ws_t ws;
   if (zh_run_is_void(ntasks_total)) return;   // (ntasks_total = the run's counter block, ZH_CNT_TASKS first)
   const uint32_t ntasks = *ntasks_total;
   if (!MORE) {
      if (blockIdx.x < ntasks) zh_post_task_one(ws, blockIdx.x, data, blocks, bars, bar_stride, work, taskmap, states, best_all, best_stride, task_bits, taskinfo);
      return;
   }
   for (uint32_t gt = first + blockIdx.x; gt < ntasks; gt += gridDim.x) {
      zh_sync();   // the task before this one is done with the workspace
      zh_post_task_one(ws, gt, data, blocks, bars, bar_stride, work, taskmap, states, best_all, best_stride, task_bits, taskinfo);
   }
}

// ---- zh_emit_tasks: token emission (blockdeflate.c:471-507) -------------------------------------------------------
// Adds `nb` bits of `code` per lane, in lane order, at bit position bitpos of the slot; complete dwords leave the LDS
// window for HBM (the task's first dword may be shared with its predecessor: atomic OR). Returns the new position.
__device__ inline uint32_t zh_emit_tile(uint32_t *obuf, uint32_t *out, uint32_t cap_bits, uint32_t first_word, uint32_t bitpos,
                                        uint64_t code, uint32_t nb) {
   const uint32_t lane = zh_lane();
   const uint32_t offs = zh_wave_excl_sum(nb);
   const uint32_t total = zh_wave_sum(nb);
   const uint32_t word0 = bitpos >> 5;
   if (nb) {
      // a token is at most 15+5+15+13 = 48 bits; shifted by < 32 it spans at most 3 dwords of the window
      const uint32_t bp = bitpos + offs;
      const uint32_t w = (bp >> 5) - word0, sh = bp & 31;
      const uint64_t lo = code << sh;
      atomicOr(&obuf[w], (uint32_t)lo);
      atomicOr(&obuf[w + 1], (uint32_t)(lo >> 32));
      if (sh) atomicOr(&obuf[w + 2], (uint32_t)(code >> (64 - sh)));
   }
   zh_sync();
   const uint32_t newpos = bitpos + total;
   const uint32_t nfull = (newpos >> 5) - word0;
   for (uint32_t k = lane; k < nfull; k += 64) {
      const uint32_t wi = word0 + k;
      if (((wi + 1) << 5) <= cap_bits) {
         if (wi == first_word)
            atomicOr(&out[wi], obuf[k]);
         else
            out[wi] = obuf[k];
      }
   }
   zh_sync();
   const uint32_t partial = obuf[nfull];
   zh_sync();
   for (uint32_t k = lane; k < ZH_OBUF_WORDS; k += 64) obuf[k] = (k == 0) ? partial : 0;
   zh_sync();
   return newpos;
}

__device__ __forceinline__ void zh_emit_task_one(zh_task_ws_t &ws, uint32_t gt, const uint8_t *__restrict__ data, const zh_block_t *__restrict__ blocks, const uint64_t *__restrict__ bars,
                                                 uint64_t bar_stride, const zh_work_t *__restrict__ work, const uint2 *__restrict__ taskmap, const zh_sbstate_t *__restrict__ states,
                                                 const uint32_t *__restrict__ best_all, uint64_t best_stride, const uint32_t *__restrict__ task_bits, uint8_t *payload,
                                                 zh_subblock_t *results, const uint2 *__restrict__ taskinfo) {
   const uint2 tm = taskmap[gt];
   const zh_work_t wk = work[tm.x];
   const zh_sbstate_t *st = states + tm.x;
   const zh_block_t blk = blocks[wk.block];
   const uint32_t prev = blk.prev;
   const uint32_t lane = zh_lane();
   const bool last = tm.y + 1 == wk.ntasks;
   uint32_t failed = st->failed;
   uint32_t nbits = 0;

   if (!failed) {
      const uint8_t *win = data + blk.win_off;
      const uint64_t *bar = bars + (uint64_t)wk.block * bar_stride;
      const uint32_t *best = best_all + (uint64_t)wk.block * best_stride;
      uint32_t *out = (uint32_t *)(payload + wk.out_off);
      const uint32_t cap_bits = wk.out_cap * 8;
      const uint32_t t0 = taskinfo[gt].x, t1 = taskinfo[gt].y & 0x7fffffffu;   // the task's range (zh_list_huge)
      (void)bar;

      for (uint32_t s = lane; s < ZH_NLIT; s += 64) {
         ws.lit_len[s] = st->lit_len[s];
         ws.lit_code[s] = st->lit_code[s];
      }
      if (lane < ZH_NDIST) {
         ws.dist_len[lane] = st->dist_len[lane];
         ws.dist_code[lane] = st->dist_code[lane];
      }
      for (uint32_t k = lane; k < ZH_OBUF_WORDS; k += 64) ws.obuf[k] = 0;
      // bit offset of the task = header + the tasks before it
      uint32_t before = 0;
      for (uint32_t j = lane; j < tm.y; j += 64) before += task_bits[wk.task_base + j];
      uint32_t bitpos = st->hdr_bits + zh_wave_sum(before);
      const uint32_t first_word = bitpos >> 5;
      zh_sync();

      uint32_t carry = 0;
      uint32_t b_next = 0, byte_next = 0;
      if (t0 + lane < t1) {
         b_next = best[t0 + lane - prev];
         byte_next = win[t0 + lane];
      }
      for (uint32_t base = t0; base < t1; base += 64) {
         const uint32_t limit = min(64u, t1 - base);
         const uint32_t pos = base + lane;
         const uint32_t b = b_next, byte = byte_next;
         b_next = 0;
         if (pos + 64 < t1) {   // the next tile's loads, under this tile's work
            b_next = best[pos + 64 - prev];
            byte_next = win[pos + 64];
         }
         const uint32_t len = b & 0xffffu;
         const uint64_t mask = zh_chain_mask(len, carry, limit);
         uint64_t code = 0;
         uint32_t nb = 0;
         if ((mask >> lane) & 1ull) {
            if (len >= ZH_MIN_MATCH) {
               const uint32_t off = b >> 16;
               const int li = zh_len_idx(len), ds = zh_dist_sym(off);
               const uint32_t lx = (uint32_t)zh_lenidx_xbits(li), dx = (uint32_t)zh_dist_xbits(ds);
               code = ws.lit_code[257 + li];
               nb = ws.lit_len[257 + li];
               code |= (uint64_t)(len - zh_lenidx_base(li)) << nb;
               nb += lx;
               code |= (uint64_t)ws.dist_code[ds] << nb;
               nb += ws.dist_len[ds];
               code |= (uint64_t)(off - zh_dist_base(ds)) << nb;
               nb += dx;
            }
            else {
               code = ws.lit_code[byte];
               nb = ws.lit_len[byte];
            }
         }
         bitpos = zh_emit_tile(ws.obuf, out, cap_bits, first_word, bitpos, code, nb);
      }
      if (last)   // end-of-block symbol
         bitpos = zh_emit_tile(ws.obuf, out, cap_bits, first_word, bitpos, lane == 0 ? (uint64_t)ws.lit_code[ZH_EOB] : 0ull,
                               lane == 0 ? (uint32_t)ws.lit_len[ZH_EOB] : 0u);
      // the partial dword at the end is shared with the next task
      if (lane == 0 && (bitpos & 31) && (((bitpos >> 5) + 1) << 5) <= cap_bits) atomicOr(&out[bitpos >> 5], ws.obuf[0]);
      nbits = bitpos;
      if (nbits > cap_bits) failed = 1;   // outgrew the slot: the stitcher stores the sub-block instead
   }

   if (last && lane == 0) {
      zh_subblock_t r;
      r.block = wk.block;
      r.start = wk.start - prev;
      r.size = wk.size;
      r.is_dynamic = st->is_dynamic;
      r.static_cost = st->static_cost;
      r.dynamic_cost = st->dynamic_cost;
      r.failed = failed;
      r.reserved = 0;
      r.nbits = nbits;
      r.bits_off = wk.out_off;
      results[wk.index] = r;
   }
}

template <bool MORE>
__global__ void __launch_bounds__(64)
zh_emit_tasks(const uint8_t *__restrict__ data, const zh_block_t *__restrict__ blocks, const uint64_t *__restrict__ bars, uint64_t bar_stride,
              const zh_work_t *__restrict__ work, const uint2 *__restrict__ taskmap, const uint32_t *__restrict__ ntasks_total,
              const zh_sbstate_t *__restrict__ states, const uint32_t *__restrict__ best_all, uint64_t best_stride,
              const uint32_t *__restrict__ task_bits, uint8_t *payload, zh_subblock_t *results, const uint2 *__restrict__ taskinfo, uint32_t first) {
   __shared__ zh_task_ws_t ws;
   if (zh_run_is_void(ntasks_total)) return;   // (ntasks_total = the run's counter block, ZH_CNT_TASKS first)
   const uint32_t ntasks = *ntasks_total;
   if (!MORE) {
      if (blockIdx.x < ntasks) zh_emit_task_one(ws, blockIdx.x, data, blocks, bars, bar_stride, work, taskmap, states, best_all, best_stride, task_bits, payload, results, taskinfo);
      return;
   }
   for (uint32_t gt = first + blockIdx.x; gt < ntasks; gt += gridDim.x) {
      zh_sync();   // the task before this one is done with the workspace
      zh_emit_task_one(ws, gt, data, blocks, bars, bar_stride, work, taskmap, states, best_all, best_stride, task_bits, payload, results, taskinfo);
   }
}
