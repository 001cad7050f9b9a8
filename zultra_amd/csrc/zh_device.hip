// zh_device.hip — host side of the device layer: context, HBM layout, launches (C ABI of include/zultra_hip.h).
//
// HBM layout of one context (B = max_blocks, N = max_block_size; S = matchfinder segments per max-block, 1 up to 64 KiB,
// Ws = segment window, <= 96 KiB: zh_common.h):
//   d_data     N + 32768 + (B-1)*N   input bytes when the caller hands over host memory (read in place otherwise)
//   d_blocks   B * 16                max-block descriptors;  d_segs  B * S * 32  matchfinder segments
//   d_sort_a/b B * S * Ws * 4  each  window positions in trigram / 4-gram-hash order (ping-pong of the radix sorts)
//   d_prev3    B * S * Ws * 8        previous occurrence of the trigram, 4-gram and 5-gram at every window position
//   d_runs     B * S * (Ws + 576) * 4  byte-run table of every segment window
//   d_match    B * N * 32            match rows, 8 x {u16 length, u16 offset} per block position
//   d_tok_pos  B * N * 4, d_tok_info B * N * 2   greedy token chain: position / packed symbols
//   d_bars     B * N / 8             barrier bitmap of every max-block (zh_parse.h)
//   d_best     B * N * 4             final parse per position
//   d_payload  B * (N + 4224)        per-sub-block bit strings (slot of sub-block k of a block starts at its offset + 64k)
//   d_states   B * 64 * 1.4 KB       per-sub-block coder state between the kernels of stage 3 (64 -> 1 in files mode)
//   d_taskmap, d_hist_part, d_task_bits   per task (T = B * (N / 2048 + 64)): owner, 320-counter histogram, bit count
//   d_stream   B * (N + ...)         stitched deflate bytes of the last batch;  d_crc / d_adler  per max-block checksums
//   small: ntok, split boundaries, counts, work items, results (+ pinned host mirrors of everything the host reads)
#include <zh_platform.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <thread>
#include <vector>

#include "../../include/zultra_hip.h"
#include "zh_common.h"
#include "zh_encode.h"
#include "zh_huffman.h"
#include "zh_matchfinder.h"
#include "zh_parse.h"
#include "zh_parse_chain.h"
#include "zh_parse_lanes.h"
#include "zh_split.h"
#include "zh_stitch.h"

#ifdef ZH_EMU
#include <mutex>
// the lock-step emulator (tests/emu) runs one kernel at a time on one OS thread's fibers: host threads that drive contexts of their
// own (libzultra.cpp: lanes of zultra_memory_compress) take turns here. The product build has no such lock.
static std::recursive_mutex g_emu_mutex;
#define ZH_EMU_SERIALIZE() std::lock_guard<std::recursive_mutex> emu_lock_(g_emu_mutex)
#else
#define ZH_EMU_SERIALIZE()
#endif

#define ZH_MAX_RUNS 8           // staggered runs of a batch (ZULTRA_HIP_STREAMS)
#define ZH_TOK_SMALL_BATCH 32u  // a batch of at most this many max-blocks follows its token chain in small chunks (zh_split.h)
#define ZH_NCNT ((uint32_t)ZH_MAX_RUNS * ZH_CNT_STRIDE)   // device counters: one block of ZH_CNT_* words per run

static_assert(sizeof(zultra_hip_block_t) == sizeof(zh_block_t), "ABI");
static_assert(sizeof(zultra_hip_subblock_t) == sizeof(zh_subblock_t), "ABI");

struct zultra_hip_ctx_s {
   int device;
   uint32_t num_cus;            // persistent kernels launch one workgroup per CU
   uint32_t total_cus;          // CUs of the device
   uint32_t max_block, max_blocks;
   uint64_t W, sort_stride, match_stride, tok_stride, best_stride, slot_stride;
   size_t data_cap;
   size_t device_bytes;         // sum of the context's device allocations (zultra_hip_ctx_info)
   hipStream_t stream;
   hipEvent_t ev[8];

   uint8_t *d_data;
   zh_block_t *d_blocks;
   uint32_t *d_sort_a, *d_sort_b, *d_runs;
   uint2 *d_prev3;              // per window position: previous trigram occurrence | distances of the previous 4-gram / 5-gram occurrence
   uint64_t run_stride;
   // matchfinder segments (zh_common.h): max-blocks above 64 KiB are cut so that every segment window fits the LDS
   uint32_t seg_n, segs_per_block, seg_W;
   zh_seg_t *d_segs;
   uint32_t *d_chunk_ctr;       // per run: per segment {next chunk of the 4-gram order to hand out, workgroups serving it}, then the
                                // segment tickets of zh_mf_frontier and zh_mf_group
   std::vector<zh_seg_t> segs;
   std::vector<uint32_t> seg_base;
   zh_match_t *d_match;
   // a stitch enqueued with its batch (zultra_hip_stitch_with_batch): armed for the next batch; what the last batch was stitched with, if it was
   int ab_armed, ab_final, stitched_valid, stitched_final, stitched_rc;
   uint32_t ab_phase, stitched_phase;
   // what the runs of the LAST batch of max-blocks listed for zh_parse_chain (their counters, read back with the batch's results): a context whose last batch had no chain
   // at all in run k enqueues run k of the next batch without chain kernels (zh_enqueue_run; zh_parse.h: zh_run_is_void)
   int chain_seen_runs;                  // runs of that batch (0: no batch yet)
   uint32_t chain_redone;                // batches run again because a run without chain kernels listed chains
   uint32_t chain_seen[ZH_MAX_RUNS];     // per run: tasks listed + cut tasks
   int chain_skip;                       // ZULTRA_HIP_CHAIN_SKIP (default 1): such a run is enqueued without chain kernels (0: always with them)
   uint32_t seen_nsubs[ZH_MAX_RUNS], seen_ntasks[ZH_MAX_RUNS];   // ... and how many sub-blocks and tasks they had: a run that stayed inside the <false> grids last time gets no <true> forms
   bool run_nomore[ZH_MAX_RUNS];         // this batch: run k was enqueued without them (its grids are in its counters, zh_run_is_void)
   uint32_t run_grids[ZH_MAX_RUNS][2];   // ... the grids (sub-blocks, tasks) of its <false> forms, as enqueued
   uint32_t *h_grids;                    // pinned: what goes into the counters (ZH_CNT_SBGRID, ZH_CNT_TASKGRID) of such runs
   bool run_nochains[ZH_MAX_RUNS];       // this batch: run k was enqueued without
   uint32_t streams_respread;   // streams replaced at creation because they shared a hardware queue with a more important one (zh_spread_streams)
   uint32_t grid_cap;           // ZULTRA_HIP_GRID_CAP (tests): the <false> grids of the per-sub-block / per-task kernels are capped here, so that the <true> forms behind them get work
   uint32_t lane_tasks;         // zh_parse_lanes: tasks per wave when forced (0: chosen per run)
   uint32_t lane_tasks_last;    // ... of the batch's last run, whose passes are the tail of the step (0: like the others)
   uint32_t run_share[ZH_MAX_RUNS];   // shares of the runs of a batch in per mille of its max-blocks (run_share[0] == 0: equal shares, see first_run_pct / last_run_pct)
   uint32_t mf_lds_cap;         // zh_mf_group: chunk size of the refinement in LDS, 0 = through HBM
   uint32_t *d_pay;             // zh_mf_group: 3 x sort_stride words per persistent workgroup (payload of the refining sort passes)
   uint32_t *d_longest;         // (round 2: a copy of slot 0 of every match row; no longer written — its readers take the rows)
   uint32_t *d_tok_pos;
   uint16_t *d_tok_info;
   uint32_t *d_ntok, *d_split_tok, *d_split_cnt, *d_sub_base;
   uint32_t *d_chunkmax, *d_spanstart, *d_spancnt;   // per chunk of ZH_TOK_CHUNK positions (zh_split.h: barriers and token chain)
   uint32_t chunks_per_block;
   uint32_t *d_best;
   uint16_t *d_cost;            // zh_parse_lanes: cost of every position from its piece's end (best_stride per max-block)
   zh_work_t *d_work;
   zh_subblock_t *d_results;
   uint8_t *d_payload;
   uint64_t *d_bars;
   uint64_t bar_stride, max_tasks;
   zh_sbstate_t *d_states;
   uint2 *d_taskmap;
   uint2 *d_taskinfo;           // per task: its range and whether zh_list_huge listed it (zh_parse_chain.h)
   uint32_t *d_ntasks, *d_hist_part, *d_task_bits;   // d_ntasks: the runs' counters, ZH_CNT_STRIDE words per run, fields ZH_CNT_* (zh_parse.h): tasks, listed tasks by
                                                     // length class and their positions, cut tasks / segments, the tickets of the persistent kernels per pass
   uint32_t *h_ntasks;          // pinned mirror, read after the batch (zultra_hip_last_stats)
   uint32_t *d_hugelist;
   uint4 *d_segtasks;           // tasks cut into speculative segments (zh_parse_chain.h): per max-block seg_tasks_per_block entries
   uint2 *d_segitems;           // their segments, as jobs of zh_parse_chain: per max-block seg_items_per_block entries
   uint2 *d_segwaves;           // ... or as segment waves of zh_parse_lanes' launch (four segments each), likewise
   uint32_t cut_len;            // ... into segments of about this many positions
   uint32_t demote_min;         // a cut task with this many failed cuts in a pass is parsed as one chain in the passes left (ZULTRA_HIP_DEMOTE; 0: never)
   uint32_t coop_tasks;         // ... a run of at most this many tasks counts as small (ZULTRA_HIP_COOP_TASKS, default: the number of CUs)
   uint32_t coop_small;         // runs of fewer tasks than CUs: tasks with a barrier-free piece longer than this go to the chain kernel (ZULTRA_HIP_COOP_SMALL; ZH_COOP_MIN otherwise)
   uint32_t cut_min;            // tasks of at least this many positions are cut into segments
   uint32_t seg_whole;          // ... with fewer, zh_parse_chain takes the segments — and the cut tasks shorter than this whole (ZULTRA_HIP_SEG_WHOLE)
   int auto_runs;               // ZULTRA_HIP_STREAMS not set: the number of runs follows the batch size
   int last_runs;               // runs the last batch was cut into
   uint32_t last_run_b0[ZH_MAX_RUNS];   // ... and the first max-block of each (diagnostics: zultra_hip_cut_tasks)
   uint32_t lane_waves;         // zh_parse_lanes waves per CU that stay next to chains (zh_parse_lanes.h)
   uint32_t mf_cu_pct;          // share of the CUs the matchfinder kernels' grids cover, percent
   uint32_t split_waves;        // waves per splitter workgroup, 0 = by max-block size
   int stagger_ev;              // event of the previous run that a run's matchfinder waits for (0: none)
   uint32_t last_run_pct;       // share of the last run, likewise
   uint32_t first_run_pct;      // share of the first run of a batch in percent of an equal share
   uint32_t seg_wide;           // a run with at least this many segments parses them in the segment workgroups of zh_parse_lanes' launch (ZULTRA_HIP_SEG_WIDE)
   int16_t *d_vecs;             // two cost vectors per segment
   uint64_t seg_tasks_per_block, seg_items_per_block;
   uint64_t *d_chain_trace;     // diagnostics (ZULTRA_HIP_CHAIN_TRACE=1): [run][pass][ticket] {positions, start, end}
   hipEvent_t ev2[16];
   // sub-batch pipelining: a batch runs as up to ZH_MAX_LANES contiguous runs of max-blocks, each on its own stream
   // "files" mode (zultra_hip_create_files): every max-block is a whole small input (< 8192 bytes, so the splitter can
   // never cut it, blockdeflate.c:646): no history, one sub-block and one task per block, no host decision anywhere in
   // the sequence -> stages 1-3 are captured once in a hipGraph and replayed per batch.
   int files_mode;
   uint32_t max_file_size;      // files mode: the size the context was created for (inputs above it are rejected: from 8192 bytes on
                                // the reference's splitter may cut an input, which the files pipeline never does)
   uint32_t max_subs;           // sub-blocks a max-block can have: 64, or 1 in files mode
   hipGraph_t graph;
   hipGraphExec_t graph_exec;
   // files mode as several runs: two graphs per run — up to the end of its first matchfinder kernel, and the rest — captured on the run's own stream
   // (zh_run_files). Two sets are kept: a caller's batches are all of one size except the last (1 000 000 inputs in batches of 65 536), and capturing
   // a set costs several milliseconds.
   struct zh_run_graphs_t {
      hipGraph_t graph[2 * ZH_MAX_RUNS];
      hipGraphExec_t exec[2 * ZH_MAX_RUNS];
      uint32_t nblocks;
      int runs;
      const uint8_t *data;
      uint64_t used;   // (tick of the last launch: the older set is the one replaced)
   } rg[2];
   uint64_t rg_tick;
   uint32_t files_chain_grid;                  // files mode: workgroups of zh_parse_chain per run and pass (ZULTRA_HIP_FILES_CHAIN_GRID; the count of chain tasks is not known to the host: the graph is fixed)
   int files_run_graphs;                       // ZULTRA_HIP_FILES_RUN_GRAPHS (default 1): 0 = several runs are launched kernel by kernel, and large batches stay one run
   uint32_t graph_nblocks;
   int graph_runs;
   const uint8_t *graph_data;
   std::vector<uint64_t> file_off;
   int nlanes;
   hipStream_t lane_stream[ZH_MAX_RUNS];
   hipEvent_t lane_ev[ZH_MAX_RUNS][24];
   hipStream_t side_stream[ZH_MAX_RUNS];     // per run: zh_parse_chain runs next to zh_parse_lanes
   hipEvent_t side_ev[ZH_MAX_RUNS][8];       // per pass: fork, join
   hipEvent_t ev_input;
   zh_subblock_t *d_results_compact;
   uint8_t *h_stage[2];         // pinned staging for callers that hand over pageable host memory (zultra_hip_staging)
   size_t h_stage_size[2];
   // pinned host mirrors: async copies to pageable memory would block the host and serialise the runs
   uint32_t *h_crc;
   zh_block_t *h_blocks;      // pinned staging of the batch's descriptors and segment list: an async copy from pageable memory is a blocking, staged one
   zh_seg_t *h_segs;
   zh_subblock_t *h_results;
   zh_stitch_item_t *d_items;
   // stream assembly on the device (zh_stitch.h): the runs' descriptors laid end to end (d_results_compact, d_nsubs: total and per run), the first
   // sub-block of every max-block, what the scan reports (pinned mirror: read after the one synchronisation of a stitch)
   uint32_t *d_nsubs, *h_nsubs, *d_blk_start;
   zh_scan_out_t *d_scan_out, *h_scan_out;
   uint64_t *d_file_off;      // files mode: first byte of every input's stream
   uint64_t *h_file_off;      // ... read back with the batch (the stitch of a files batch goes out with it, zh_enqueue_files_tail)
   int files_stitched, files_stitch_rc;   // the last files batch was stitched with its kernels; zh_stitch_verdict of that
   uint32_t *d_task_prefix, *h_task_prefix;   // files mode: exclusive prefix of the inputs' task counts, computed by the host from the sizes it was handed (zh_plan_files); B + 1 entries
   uint32_t *d_stream;        // stitched deflate bits of the last batch
   size_t stream_cap;         // bytes
   uint32_t *d_crc, *d_crc_tables, *d_adler;
   uint32_t *h_adler;
   std::vector<uint32_t> adler;
   std::vector<zh_stitch_item_t> items;
   std::vector<uint32_t> crc;
   int payload_on_host;       // lazily copied

   // host mirrors of the last batch
   std::vector<zh_block_t> blocks;
   std::vector<zh_subblock_t> results;
   uint8_t *h_payload;   // pinned
   size_t payload_size;
   const uint8_t *cur_data;   // device pointer the kernels read
   uint32_t nblocks, nsubs;
   zultra_hip_timing_t timing;
   char err[256];
};

#define ZH_CHECK(ctx, call)                                                                              \
   do {                                                                                                  \
      hipError_t e_ = (call);                                                                            \
      if (e_ != hipSuccess) {                                                                            \
         snprintf((ctx)->err, sizeof((ctx)->err), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                  __FILE__, __LINE__);                                                                   \
         return -1;                                                                                      \
      }                                                                                                  \
   } while (0)

// (64-bit min / max by name: hipcc's host-side `min` picks the int overload for two 64-bit arguments — 0xFFFFFFFF became -1)
static inline uint64_t zh_min64(uint64_t a, uint64_t b) { return a < b ? a : b; }
static inline uint64_t zh_max64(uint64_t a, uint64_t b) { return a > b ? a : b; }

static uint32_t zh_clamp_block(uint32_t n) {
   if (!n) n = 1048576;   // libzultra.c:87-92
   if (n < ZH_MIN_BLOCK) n = ZH_MIN_BLOCK;
   if (n > ZH_MAX_BLOCK) n = ZH_MAX_BLOCK;
   return n;
}

// ---- wave primitive self-check -------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) zh_selftest_kernel(uint32_t seed, uint32_t *bad) {
   __shared__ uint32_t v[64];
   const uint32_t lane = zh_lane();
   uint32_t errors = 0;
   for (uint32_t round = 0; round < 64; round++) {
      uint32_t x = (seed + round * 977u + lane * 2654435761u) * 2246822519u;
      x ^= x >> 15;
      if (round & 1) x &= 0xffffu;      // small values: sums stay exact
      if (round == 5) x = 0xFFFFFFFFu;  // identity of the min reduction in every lane
      v[lane] = x;
      zh_sync();
      uint32_t mn = 0xFFFFFFFFu, sm = 0, ex = 0, rmn = 0xFFFFFFFFu;
      for (uint32_t k = 0; k < 64; k++) {
         mn = min(mn, v[k]);
         sm += v[k];
         if (k < lane) ex += v[k];
         if ((k >> 4) == (lane >> 4)) rmn = min(rmn, v[k]);
      }
      if (zh_wave_min(x) != mn) errors++;
      if (zh_wave_min_bcast(x) != mn) errors++;
      if (zh_wave_sum(x) != sm) errors++;
      if (zh_wave_excl_sum(x) != ex) errors++;
      {
         uint32_t im = 0;
         for (uint32_t k = 0; k <= lane; k++) im = max(im, v[k]);
         if (zh_wave_incl_max(x) != im) errors++;
      }
      if (zh_row_min(x) != rmn) errors++;
      if (zh_readlane(x, (int)(round & 63)) != v[round & 63]) errors++;
      if (zh_shfl(x, (int)((lane * 7 + round) & 63)) != v[(lane * 7 + round) & 63]) errors++;
      uint64_t b = zh_ballot((x & 4) != 0), bref = 0;
      for (uint32_t k = 0; k < 64; k++) bref |= (uint64_t)((v[k] & 4) != 0) << k;
      if (b != bref) errors++;
      if (zh_readfirstlane(x) != v[0]) errors++;
      if (zh_wave_shr1(x, 0xABCD0000u + round) != (lane ? v[lane - 1] : 0xABCD0000u + round)) errors++;
      if (zh_row_shr<1>(x) != v[(lane & 15) >= 1 ? lane - 1 : lane]) errors++;
      if (zh_row_shr<4>(x) != v[(lane & 15) >= 4 ? lane - 4 : lane]) errors++;
      if (zh_row_shl<1>(x) != v[(lane & 15) + 1 < 16 ? lane + 1 : lane]) errors++;
      if (zh_row_shl<2>(x) != v[(lane & 15) + 2 < 16 ? lane + 2 : lane]) errors++;
      {
         // reads at any byte address of LDS (zh_load32_any / zh_load128_any: the matchfinder's string probes)
         const uint8_t *vb = (const uint8_t *)v;
         const uint32_t o = (lane * 5u + round) % 237u;   // 0..236: 16 bytes from there stay inside the 256
         uint32_t w[4];
         for (uint32_t k = 0; k < 4; k++)
            w[k] = (uint32_t)vb[o + 4 * k] | ((uint32_t)vb[o + 4 * k + 1] << 8) | ((uint32_t)vb[o + 4 * k + 2] << 16) | ((uint32_t)vb[o + 4 * k + 3] << 24);
         if (zh_load32_any(vb + o) != w[0]) errors++;
         const zh_u128_any_t q = zh_load128_any(vb + o);
         if (q.x != w[0] || q.y != w[1] || q.z != w[2] || q.w != w[3]) errors++;
      }
      {
         // Returning LDS atomics hand out their values in lane order among the lanes that meet at an address, and in program order between
         // instructions: the stable counting passes of the matchfinder take a lane's place among the lanes of its digit from one of them
         // (zh_mf_group_lds.h, zh_mf_sort_pass). Digits from alphabets of 1 to 256 symbols, plain counters and two 16-bit counters to a word.
         __shared__ uint32_t cnt32[256], cnt16[128];
         for (uint32_t k = lane; k < 256; k += 64) cnt32[k] = 0;
         for (uint32_t k = lane; k < 128; k += 64) cnt16[k] = 0;
         zh_sync();
         const uint32_t alpha = 1u << (round % 9u);
         for (uint32_t step = 0; step < 3; step++) {
            const uint32_t y = (x ^ (step * 0x9e3779b9u)) * 2654435761u;
            const uint32_t d = ((y >> 11) % alpha) * (256u / alpha);
            const bool valid = (y & 0x7u) != 0;
            uint32_t got32 = 0, got16 = 0;
            if (valid) {
               got32 = zh_atomic_add_lds(&cnt32[d], 1u);
               got16 = (zh_atomic_add_lds(&cnt16[d >> 1], 1u << ((d & 1u) << 4)) >> ((d & 1u) << 4)) & 0xffffu;
            }
            zh_lockstep_point();
            v[lane] = valid ? d : 0xffffffffu;
            zh_sync();
            uint32_t below = 0, all = 0;
            for (uint32_t k = 0; k < 64; k++) {
               if (v[k] == d && k < lane) below++;
               if (v[k] == d) all++;
            }
            // the counter held `all` less before this step's lanes arrived; this lane is the (below)-th of them
            if (valid && (got32 != cnt32[d] - all + below || got16 != got32)) errors++;
            {
               // zh_peers8: the lanes that hold this lane's digit
               uint64_t pref = 0;
               for (uint32_t k = 0; k < 64; k++)
                  if (v[k] == d) pref |= 1ull << k;
               const uint64_t peers = zh_peers8(d, valid);
               if (valid && peers != pref) errors++;
            }
            zh_sync();
         }
      }
      zh_sync();
   }
   // RFC 1951 symbol arithmetic against first principles
   if (lane == 0) {
      uint32_t d = 1;
      for (int s = 0; s < 30; s++) {
         int xb = s < 4 ? 0 : s / 2 - 1;
         if (zh_dist_base(s) != d || zh_dist_xbits(s) != xb) errors++;
         for (uint32_t k = 0; k < (1u << xb); k += (xb > 6 ? 37 : 1))
            if (zh_dist_sym(d + k) != s) errors++;
         if (zh_dist_sym(d + (1u << xb) - 1) != s) errors++;
         d += 1u << xb;
      }
      for (uint32_t len = 3; len <= 258; len++) {
         int idx = zh_len_idx(len);
         uint32_t base = zh_lenidx_base(idx);
         if (len < base || len - base >= (1u << zh_lenidx_xbits(idx)) || idx > 28) errors++;
         if (idx < 28 && zh_lenidx_base(idx + 1) <= len && !(idx == 27 && len < 258)) errors++;
      }
   }
   if (errors) atomicAdd(bad, errors);
}

extern "C" int zultra_hip_selftest(void) {
   int ndev = 0;
   if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return -1;
   uint32_t *d_bad = NULL, bad = 0;
   if (hipMalloc((void **)&d_bad, sizeof(uint32_t)) != hipSuccess) return -2;
   if (hipMemset(d_bad, 0, sizeof(uint32_t)) != hipSuccess) return -3;
   ZH_LAUNCH(zh_selftest_kernel, 8, 64, 0, 12345u, d_bad);
   if (hipMemcpy(&bad, d_bad, sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) return -4;
   (void)hipFree(d_bad);
   return (int)bad;
}

// ---- PMC calibration probe ---------------------------------------------------------------------------------------
// A streaming dword copy of a known size, in the access width the hot kernels use (4 B per lane). Run under
// rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE it tells how many bytes the counters report per byte moved for this width
// (MI355X_MICROARCH.md: only 16 B/lane streams are calibrated; other widths must be calibrated in place).
__global__ void __launch_bounds__(256) zh_probe_copy_dword(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, size_t nwords) {
   for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < nwords; k += (size_t)gridDim.x * 256) dst[k] = src[k] + 1u;
}

extern "C" int zultra_hip_traffic_probe(size_t nbytes) {
   uint32_t *a = NULL, *b = NULL;
   const size_t nwords = nbytes / 4;
   if (hipMalloc((void **)&a, nwords * 4) != hipSuccess) return -1;
   if (hipMalloc((void **)&b, nwords * 4) != hipSuccess) {
      (void)hipFree(a);
      return -1;
   }
   (void)hipMemset(a, 1, nwords * 4);
   ZH_LAUNCH(zh_probe_copy_dword, 256 * 16, 256, 0, (const uint32_t *)a, b, nwords);
   const hipError_t e = hipDeviceSynchronize();
   (void)hipFree(a);
   (void)hipFree(b);
   return e == hipSuccess ? 0 : -2;
}

// Streaming copy with 16 B per lane (the widest access, what the guide's 6.29 TB/s "measured peak" was taken with): the
// second denominator of the roofline. Returns GB/s of bytes read + bytes written, negative on errors.
// (four 16-byte loads in flight per lane, non-temporal both ways: a copy that is read and written once has no use for the caches.
// Round 2's one-load-per-iteration loop reached 4.5-4.9 TB/s, 23 % under the 6.29 TB/s the guide quotes for a float4 copy.)
#ifdef ZH_EMU   // (the emulator build compiles with g++: plain accesses)
struct zh_copy16_t {
   uint32_t v[4];
};
#define __builtin_nontemporal_load(p_) (*(p_))
#define __builtin_nontemporal_store(v_, p_) (*(p_) = (v_))
#else
typedef uint32_t zh_copy16_t __attribute__((ext_vector_type(4)));
#endif
__global__ void __launch_bounds__(256) zh_probe_copy_x4(const uint4 *__restrict__ src4, uint4 *__restrict__ dst4, size_t n16) {
   const zh_copy16_t *src = (const zh_copy16_t *)src4;
   zh_copy16_t *dst = (zh_copy16_t *)dst4;
   const size_t stride = (size_t)gridDim.x * 256;
   size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
   for (; k + 3 * stride < n16; k += 4 * stride) {
      const zh_copy16_t a = __builtin_nontemporal_load(src + k), b = __builtin_nontemporal_load(src + k + stride);
      const zh_copy16_t c = __builtin_nontemporal_load(src + k + 2 * stride), d = __builtin_nontemporal_load(src + k + 3 * stride);
      __builtin_nontemporal_store(a, dst + k);
      __builtin_nontemporal_store(b, dst + k + stride);
      __builtin_nontemporal_store(c, dst + k + 2 * stride);
      __builtin_nontemporal_store(d, dst + k + 3 * stride);
   }
   for (; k < n16; k += stride) dst[k] = src[k];
}
#ifdef ZH_EMU
#undef __builtin_nontemporal_load
#undef __builtin_nontemporal_store
#endif
// the plainest form, what the guide's figure is quoted for: one 16-byte element per thread, default cache policy, a grid that covers the buffer
__global__ void __launch_bounds__(256) zh_probe_copy_plain(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
   const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
   if (k < n16) dst[k] = src[k];
}

extern "C" double zultra_hip_copy_bandwidth(size_t nbytes, int iters) {
   uint4 *a = NULL, *b = NULL;
   const size_t n16 = nbytes / 16;
   if (!n16 || iters < 1) return -1.0;
   if (hipMalloc((void **)&a, n16 * 16) != hipSuccess) return -1.0;
   if (hipMalloc((void **)&b, n16 * 16) != hipSuccess) {
      (void)hipFree(a);
      return -1.0;
   }
   (void)hipMemset(a, 1, n16 * 16);
   hipEvent_t e0 = NULL, e1 = NULL;
   (void)hipEventCreate(&e0);
   (void)hipEventCreate(&e1);
   // the better of two forms, each its own timed loop: four non-temporal loads in flight per lane from persistent workgroups (round 3: 5.0-5.1
   // TB/s on this pool), and one element per thread from a grid over the whole buffer (the guide's float4 copy: 6.29 TB/s)
   float best_ms = 0;
   hipError_t e = hipSuccess;
   for (int form = 0; form < 2 && e == hipSuccess; form++) {
      const uint32_t plain_grid = (uint32_t)((n16 + 255) / 256);
      if (form == 0)
         ZH_LAUNCH(zh_probe_copy_x4, 256 * 32, 256, 0, (const uint4 *)a, b, n16);   // warm-up
      else
         ZH_LAUNCH(zh_probe_copy_plain, plain_grid, 256, 0, (const uint4 *)a, b, n16);
      (void)hipEventRecord(e0, 0);
      for (int i = 0; i < iters; i++) {
         if (form == 0)
            ZH_LAUNCH(zh_probe_copy_x4, 256 * 32, 256, 0, (const uint4 *)a, b, n16);
         else
            ZH_LAUNCH(zh_probe_copy_plain, plain_grid, 256, 0, (const uint4 *)a, b, n16);
      }
      (void)hipEventRecord(e1, 0);
      e = hipEventSynchronize(e1);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (e == hipSuccess && ms > 0 && (best_ms == 0 || ms < best_ms)) best_ms = ms;
   }
   (void)hipEventDestroy(e0);
   (void)hipEventDestroy(e1);
   (void)hipFree(a);
   (void)hipFree(b);
   if (e != hipSuccess || best_ms <= 0) return -2.0;
   return 2.0 * (double)(n16 * 16) * iters / (best_ms * 1e-3) / 1e9;
}

// The pipeline uses up to five streams per context (two runs, a side stream each for zh_parse_chain, one for the stitcher) next
// to the application's own. The HIP runtime multiplexes streams onto 4 hardware queues by default, and two streams that
// share a queue run strictly one after the other: measured, a run's zh_parse_chain then blocks the other run's kernels
// (source code, three runs: 89 ms per 50 MB with 4 queues, 60 ms with 8). The variable is read when the runtime
// initialises, so it is set when the library is loaded — unless the application chose a value or initialised HIP earlier.
__attribute__((constructor)) static void zh_runtime_hints(void) { (void)setenv("GPU_MAX_HW_QUEUES", "8", 0); }

extern "C" int zultra_hip_device_count(void) {
   int n = 0;
   if (hipGetDeviceCount(&n) != hipSuccess) return 0;
   return n;
}

template <typename T>
static int zh_alloc(zultra_hip_ctx_t *c, T **p, size_t count) {
   ZH_CHECK(c, hipMalloc((void **)p, count * sizeof(T)));
   c->device_bytes += count * sizeof(T);
   return 0;
}

extern "C" void zultra_hip_destroy(zultra_hip_ctx_t *c) {
   if (!c) return;
   (void)hipSetDevice(c->device);
   (void)hipFree(c->d_data);
   (void)hipFree(c->d_blocks);
   (void)hipFree(c->d_sort_a);
   (void)hipFree(c->d_sort_b);
   (void)hipFree(c->d_prev3);
   (void)hipFree(c->d_runs);
   (void)hipFree(c->d_segs);
   (void)hipFree(c->d_chunk_ctr);
   (void)hipFree(c->d_match);
   (void)hipFree(c->d_longest);
   (void)hipFree(c->d_pay);
   (void)hipFree(c->d_tok_pos);
   (void)hipFree(c->d_tok_info);
   (void)hipFree(c->d_ntok);
   (void)hipFree(c->d_chunkmax);
   (void)hipFree(c->d_spanstart);
   (void)hipFree(c->d_spancnt);
   (void)hipFree(c->d_split_tok);
   (void)hipFree(c->d_split_cnt);
   (void)hipFree(c->d_sub_base);
   (void)hipFree(c->d_best);
   (void)hipFree(c->d_cost);
   (void)hipFree(c->d_work);
   (void)hipFree(c->d_results);
   (void)hipFree(c->d_payload);
   (void)hipFree(c->d_bars);
   (void)hipFree(c->d_states);
   (void)hipFree(c->d_taskmap);
   (void)hipFree(c->d_taskinfo);
   (void)hipFree(c->d_nsubs);
   (void)hipFree(c->d_blk_start);
   (void)hipFree(c->d_scan_out);
   (void)hipFree(c->d_file_off);
   (void)hipFree(c->d_task_prefix);
   if (c->h_task_prefix) (void)hipHostFree(c->h_task_prefix);
   if (c->h_grids) (void)hipHostFree(c->h_grids);
   if (c->h_file_off) (void)hipHostFree(c->h_file_off);
   if (c->h_nsubs) (void)hipHostFree(c->h_nsubs);
   if (c->h_scan_out) (void)hipHostFree(c->h_scan_out);
   (void)hipFree(c->d_ntasks);
   (void)hipFree(c->d_hugelist);
   (void)hipFree(c->d_segtasks);
   (void)hipFree(c->d_segwaves);
   (void)hipFree(c->d_segitems);
   (void)hipFree(c->d_vecs);
   (void)hipFree(c->d_chain_trace);
   (void)hipFree(c->d_hist_part);
   (void)hipFree(c->d_task_bits);
   for (int i = 0; i < 16; i++)
      if (c->ev2[i]) (void)hipEventDestroy(c->ev2[i]);
   for (int k = 0; k < ZH_MAX_RUNS; k++) {
      for (int i = 0; i < 24; i++)
         if (c->lane_ev[k][i]) (void)hipEventDestroy(c->lane_ev[k][i]);
      if (c->lane_stream[k]) (void)hipStreamDestroy(c->lane_stream[k]);
      for (int i = 0; i < 8; i++)
         if (c->side_ev[k][i]) (void)hipEventDestroy(c->side_ev[k][i]);
      if (c->side_stream[k]) (void)hipStreamDestroy(c->side_stream[k]);
   }
   if (c->ev_input) (void)hipEventDestroy(c->ev_input);
   for (int i = 0; i < 2; i++)
      for (int k = 0; k < 2 * ZH_MAX_RUNS; k++) {
         if (c->rg[i].exec[k]) (void)hipGraphExecDestroy(c->rg[i].exec[k]);
         if (c->rg[i].graph[k]) (void)hipGraphDestroy(c->rg[i].graph[k]);
      }
   if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
   if (c->graph) (void)hipGraphDestroy(c->graph);
   for (int k = 0; k < 2; k++)
      if (c->h_stage[k]) (void)hipHostFree(c->h_stage[k]);
   if (c->h_crc) (void)hipHostFree(c->h_crc);
   if (c->h_blocks) (void)hipHostFree(c->h_blocks);
   if (c->h_segs) (void)hipHostFree(c->h_segs);
   if (c->h_ntasks) (void)hipHostFree(c->h_ntasks);
   if (c->h_results) (void)hipHostFree(c->h_results);
   (void)hipFree(c->d_results_compact);
   (void)hipFree(c->d_items);
   (void)hipFree(c->d_stream);
   (void)hipFree(c->d_crc);
   (void)hipFree(c->d_adler);
   if (c->h_adler) (void)hipHostFree(c->h_adler);
   (void)hipFree(c->d_crc_tables);
   if (c->h_payload) (void)hipHostFree(c->h_payload);
   for (int i = 0; i < 8; i++)
      if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
   if (c->stream) (void)hipStreamDestroy(c->stream);
   delete c;
}

// Environment switches. zh_env: part of the shipped library's surface. zh_knob: tuning experiments — compiled to their defaults
// unless the library is a probe build (-DZH_TUNING_KNOBS): a process environment cannot move the product off its measured settings.
static int zh_env(const char *name, int dflt) {
   const char *e = getenv(name);
   return e ? atoi(e) : dflt;
}
static int zh_knob(const char *name, int dflt) {
#ifdef ZH_TUNING_KNOBS
   return zh_env(name, dflt);
#else
   (void)name;
   return dflt;
#endif
}

#ifndef ZH_EMU
// ---- every run's stream on a hardware queue of its own -----------------------------------------------------------------------------------------------------
// The HIP runtime hands a stream one of a few hardware queues when it is created and keeps it there; two streams on one queue run strictly one after the
// other. With eight queues and a fresh process the library's streams land on queues of their own (zh_runtime_hints); behind other contexts, or in a process that
// initialised HIP with the runtime's default of four before the library was loaded, two runs of a batch can share a queue — the 100 MB step then takes 49 ms
// instead of 39 (profiles/r05_timeline_second_context_static_queues.txt; round 6: tools/ab_step.py under torch without the variable). The runtime does not say
// which queue a stream has, but it shows: a context lets all its run and chain streams spin for 0.2 ms at once and looks at who ran WHEN. A stream that did not
// overlap one ahead of it in the order of importance (the runs' streams, then their chain streams) is replaced by a fresh one — created before the old one is
// destroyed, so the runtime's least-used-queue rule puts it elsewhere — and the test repeated, a few times at most. ~0.3 ms of context creation when all is well.
// Opt-in (see zh_spread_streams): an application that cannot export GPU_MAX_HW_QUEUES before it initialises HIP, and runs one context at a time, sets ZULTRA_HIP_SPREAD_STREAMS=1.
__global__ void zh_queue_probe(uint64_t *out, uint64_t ticks) {
   const uint64_t t0 = zh_wall_clock();
   uint64_t t = t0;
   for (uint32_t guard = 0; t - t0 < ticks && guard < (1u << 20); guard++) {
      __builtin_amdgcn_s_sleep(64);
      t = zh_wall_clock();
   }
   out[0] = t0;
   out[1] = t;
}

static std::mutex g_probe_mutex;   // one context probes at a time: two probes on a shared queue would read each other as a collision

static int zh_spread_streams(zultra_hip_ctx_t *c) {
   // OFF unless asked for (ZULTRA_HIP_SPREAD_STREAMS=1, 2 = and say what was done): it repairs a context that is alone on the device in a process with too few queues
   // (49 -> 40 ms per 100 MB step), but where several contexts run side by side each one's streams on queues of their own means the contexts share every queue with
   // each other — measured: three jobs in flight 2.64-2.71 GB/s without it, 2.45-2.64 with (tools/r06_3jobs*.sh) — and a probe taken on a busy device can misread.
   if (zh_env("ZULTRA_HIP_SPREAD_STREAMS", 0) <= 0) return 0;
   std::lock_guard<std::mutex> probe_lock(g_probe_mutex);
   const int n = c->nlanes;
   hipStream_t *order[2 * ZH_MAX_RUNS];
   bool high[2 * ZH_MAX_RUNS];
   int m = 0;
   // (the first three runs, their chain streams, then the rest: what a batch of less than 256 MiB uses comes first)
   const int scope = zh_env("ZULTRA_HIP_SPREAD_SCOPE", 0);   // 0: the runs' streams only; 1: their chain streams too
   for (int k = 0; k < n && k < 3; k++) { order[m] = &c->lane_stream[k]; high[m++] = false; }
   if (scope >= 1) for (int k = 0; k < n && k < 3; k++) { order[m] = &c->side_stream[k]; high[m++] = true; }
   for (int k = 3; k < n; k++) { order[m] = &c->lane_stream[k]; high[m++] = false; }
   if (scope >= 1) for (int k = 3; k < n; k++) { order[m] = &c->side_stream[k]; high[m++] = true; }
   uint64_t *d_t = NULL, *h_t = NULL;
   ZH_CHECK(c, hipMalloc((void **)&d_t, (size_t)m * 2 * sizeof(uint64_t)));
   ZH_CHECK(c, hipHostMalloc((void **)&h_t, (size_t)m * 2 * sizeof(uint64_t), 0));
   int lo_prio = 0, hi_prio = 0;
   (void)hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio);
   int rc = 0;
   c->streams_respread = 0;
   for (int attempt = 0; attempt < 12; attempt++) {
      for (int i = 0; i < m; i++) hipLaunchKernelGGL(zh_queue_probe, dim3(1), dim3(1), 0, *order[i], d_t + 2 * i, (uint64_t)20000);   // 0.2 ms of the 100 MHz clock
      for (int i = 0; i < m; i++) if (hipStreamSynchronize(*order[i]) != hipSuccess) rc = -1;
      if (rc != 0 || hipMemcpy(h_t, d_t, (size_t)m * 2 * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) { rc = -1; break; }
      int bad = -1;
      for (int i = 1; i < m && bad < 0; i++)
         for (int j = 0; j < i; j++)
            // i ran right BEHIND j (or j right behind i), not next to it: the signature of one queue. (A probe that merely started late — the device busy with another
            // context's kernels — starts when THEY end, not when j does: no reason to move it. Measured with three contexts created and run side by side: replacing on
            // mere non-overlap cost the three-jobs-in-flight leg 6 %.)
            if (!(h_t[2 * i] < h_t[2 * j + 1] && h_t[2 * j] < h_t[2 * i + 1])) {
               const uint64_t gap_ij = h_t[2 * i] >= h_t[2 * j + 1] ? h_t[2 * i] - h_t[2 * j + 1] : ~0ull, gap_ji = h_t[2 * j] >= h_t[2 * i + 1] ? h_t[2 * j] - h_t[2 * i + 1] : ~0ull;
               if (gap_ij <= 3000u || gap_ji <= 3000u) { bad = i; break; }   // (30 us of the 100 MHz clock)
            }
      if (bad < 0) break;
      hipStream_t fresh = NULL;
      const hipError_t e = high[bad] ? hipStreamCreateWithPriority(&fresh, hipStreamNonBlocking, hi_prio) : hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking);
      if (e != hipSuccess) { rc = -1; break; }
      (void)hipStreamDestroy(*order[bad]);
      *order[bad] = fresh;
      c->streams_respread++;
   }
   (void)hipFree(d_t);
   (void)hipHostFree(h_t);
   if (zh_env("ZULTRA_HIP_SPREAD_STREAMS", 0) == 2) fprintf(stderr, "zultra_amd: %u of %d streams replaced for a hardware queue of their own\n", c->streams_respread, m);
   if (rc != 0) snprintf(c->err, sizeof(c->err), "stream placement probe failed");
   return rc;
}
#endif

static int zh_enqueue_stitch(zultra_hip_ctx_t *c, hipStream_t st, uint32_t phase, int final_block, int files, bool scan_only, bool clear);
static int zh_stitch_verdict(zultra_hip_ctx_t *c, bool scan_only);

// streams (and payload areas) a context holds for the staggered runs of a batch: ZULTRA_HIP_STREAMS, 0 or unset = auto = four (zh_create_buffers and
// zultra_hip_context_bytes_on must agree: the host layer budgets batches with the latter)
static int zh_env_lanes(void) {
   const int streams = zh_env("ZULTRA_HIP_STREAMS", 0);
   return streams > 0 ? min(streams, (int)ZH_MAX_RUNS) : 4;
}

static int zh_create_buffers(zultra_hip_ctx_t *c) {
   const uint64_t B = c->max_blocks, N = c->max_block;
   ZH_CHECK(c, hipSetDevice(c->device));
   {
      // once per device and process: the wave primitives and the LDS behaviour the kernels rely on (zh_selftest_kernel) — a device that
      // fails gets no context (the library has no other path)
      static std::mutex checked_mutex;
      static uint64_t checked_ok[4] = {0, 0, 0, 0};
      std::lock_guard<std::mutex> lock(checked_mutex);
      const uint32_t dv = (uint32_t)c->device & 255u;
      if (!((checked_ok[dv >> 6] >> (dv & 63u)) & 1ull)) {
         const int bad = zultra_hip_selftest();
         if (bad != 0) {
            snprintf(c->err, sizeof(c->err), "device %d fails the self-check of the wave and LDS primitives (%d)", c->device, bad);
            return -1;
         }
         checked_ok[dv >> 6] |= 1ull << (dv & 63u);
      }
   }
   {
      int n = 0;
      if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || n <= 0) n = 256;
      c->total_cus = (uint32_t)n;
      c->num_cus = c->total_cus;   // what the persistent kernels may fill (keeping 8..32 CUs free for the chain kernels was measured in round 3: +3 ms per 100 MB)
      // the matchfinder kernels take all their LDS dynamically (zh_matchfinder.h): more than the 64 KiB default limit
      ZH_CHECK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&zh_mf_group<true>), hipFuncAttributeMaxDynamicSharedMemorySize, ZH_MF_GROUP_LDS));
      ZH_CHECK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&zh_mf_frontier<true>), hipFuncAttributeMaxDynamicSharedMemorySize, ZH_MF_FRONTIER_LDS));
      ZH_CHECK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&zh_mf_group_big), hipFuncAttributeMaxDynamicSharedMemorySize, ZH_MF_GROUP_LDS));
   }
   ZH_CHECK(c, hipStreamCreate(&c->stream));
   for (int i = 0; i < 8; i++) ZH_CHECK(c, hipEventCreate(&c->ev[i]));
   for (int i = 0; i < 16; i++) ZH_CHECK(c, hipEventCreate(&c->ev2[i]));
   {
      // ---- switches of the shipped library (INTEGRATION.md lists them): they pick between product paths that the size and kind of the data
      //      would otherwise pick, so that tests can force every path on small inputs; and one diagnostic
      if (zh_env("ZULTRA_HIP_CHAIN_TRACE", 0)) {
         if (zh_alloc(c, &c->d_chain_trace, (size_t)3 * ZH_TRACE_SLOTS * 4 * ZH_MAX_RUNS)) return -1;   // (the getter returns the first four runs)
         ZH_CHECK(c, hipMemset(c->d_chain_trace, 0, (size_t)3 * ZH_TRACE_SLOTS * 4 * ZH_MAX_RUNS * sizeof(uint64_t)));
      }
      c->demote_min = (uint32_t)zh_env("ZULTRA_HIP_DEMOTE", 2);        // a cut task with this many failed cuts in a pass becomes one chain (0: never)
      c->coop_small = (uint32_t)zh_env("ZULTRA_HIP_COOP_SMALL", 256);  // small runs: tasks with a barrier-free piece above this go to the chain kernel
      if (c->coop_small < 64u) c->coop_small = 64u;
      if (c->coop_small > ZH_COOP_MIN) c->coop_small = ZH_COOP_MIN;
      c->seg_whole = (uint32_t)zh_env("ZULTRA_HIP_SEG_WHOLE", 16384);  // cut tasks shorter than this are parsed whole when zh_parse_chain takes the segments
      c->chain_skip = zh_env("ZULTRA_HIP_CHAIN_SKIP", 1);   // 1: a run whose counterpart in the context's last batch listed no chain is enqueued without chain kernels (0: always with them)
      c->chain_seen_runs = 0;
      c->grid_cap = (uint32_t)max(0, zh_env("ZULTRA_HIP_GRID_CAP", 0));   // tests: cap of the <false> grids of zh_sb_init / zh_sb_build / zh_list_huge / zh_post_tasks / zh_emit_tasks (0: none) — the
                                                                          // <true> forms that stride over what lies beyond a grid are otherwise reached by heavily splitting data only
      c->seg_wide = (uint32_t)zh_env("ZULTRA_HIP_SEG_WIDE", 1024);     // a run with at least this many segments parses them in the segment workgroups of zh_parse_lanes' launch
      c->mf_lds_cap = (uint32_t)max(0, zh_env("ZULTRA_HIP_MF_CAP", (int)ZH_MFL_CAP_LIMIT));   // elements per chunk of zh_mf_group's refinement in LDS (zh_mf_group_lds.h)
      const int streams = zh_env("ZULTRA_HIP_STREAMS", 0);              // staggered runs per batch; not set: three, four for batches of 256 MiB and more
      c->nlanes = zh_env_lanes();
      c->auto_runs = streams > 0 ? 0 : 1;   // (measured with the stagger below, runs = 2 / 3 / 4 / 6: 100 MB of real text 49.8 / 49.6 / 51.2 / - ms; configuration 3
                                        // 31.0 / 29.1 / 29.5 / -; 1 GiB of configuration 4 910 / 883 / 789 / 812)
      if (c->nlanes < 1) c->nlanes = 1;
      if (c->nlanes > ZH_MAX_RUNS) c->nlanes = ZH_MAX_RUNS;
      c->files_run_graphs = zh_env("ZULTRA_HIP_FILES_RUN_GRAPHS", 1);  // files mode: 0 = several runs are launched kernel by kernel, and large batches stay one run
      // ---- tuning knobs: read in probe builds only (-DZH_TUNING_KNOBS, tools/build_variant.sh); the shipped library has the defaults compiled in
      c->cut_len = (uint32_t)zh_knob("ZULTRA_HIP_CUT_LEN", (int)ZH_CUT_LEN);   // positions per segment, about (ZH_CUT_WARM .. ZH_CUT_LEN)
      if (c->cut_len < ZH_CUT_WARM) c->cut_len = ZH_CUT_WARM;
      if (c->cut_len > ZH_CUT_LEN) c->cut_len = ZH_CUT_LEN;   // (the buffers are sized for ZH_CUT_WARM, the smallest)
      c->files_chain_grid = (uint32_t)max(1, min((int)ZH_CHAIN_GRID, zh_knob("ZULTRA_HIP_FILES_CHAIN_GRID", (int)ZH_CHAIN_GRID)));
      c->coop_tasks = (uint32_t)zh_knob("ZULTRA_HIP_COOP_TASKS", (int)c->num_cus);   // a run of at most this many tasks counts as small
      c->cut_min = (uint32_t)zh_knob("ZULTRA_HIP_CUT_MIN", (int)ZH_CUT_MIN);         // tasks of at least this many positions are cut (>= 2 * ZH_CUT_WARM)
      if (c->cut_min < 2u * ZH_CUT_WARM) c->cut_min = 2u * ZH_CUT_WARM;
      c->split_waves = (uint32_t)zh_knob("ZULTRA_HIP_SPLIT_WAVES", 0);               // waves per splitter workgroup (2, 4, 8, 16; default by max-block size)
      c->lane_waves = (uint32_t)max(1, min(16, zh_knob("ZULTRA_HIP_LANE_WAVES", 12)));
      c->lane_tasks = (uint32_t)max(0, min((int)ZH_LP_TASKS, zh_knob("ZULTRA_HIP_LANE_TASKS", 0)));   // tasks per wave of zh_parse_lanes; 0: by the size of the run
      c->lane_tasks_last = (uint32_t)max(0, min((int)ZH_LP_TASKS, zh_knob("ZULTRA_HIP_LANE_TASKS_LAST", 0)));
      memset(c->run_share, 0, sizeof(c->run_share));
#ifdef ZH_TUNING_KNOBS
      if (const char *rs = getenv("ZULTRA_HIP_RUN_SHARES")) {   // probe builds: "500,300,150,50" — as many runs, of these shares (per mille)
         int n = 0;
         while (*rs && n < ZH_MAX_RUNS) {
            c->run_share[n++] = (uint32_t)atoi(rs);
            while (*rs && *rs != ',' && *rs != ':') rs++;
            if (*rs) rs++;
         }
         if (n >= 1 && streams == 0) {
            c->nlanes = n;
            c->auto_runs = 0;
         }
      }
#endif
      c->mf_cu_pct = (uint32_t)max(1, min(100, zh_knob("ZULTRA_HIP_MF_CUS", 100)));  // share of the CUs the matchfinder's persistent workgroups take, in percent
      c->stagger_ev = zh_knob("ZULTRA_HIP_STAGGER", 2);   // which stage of the previous run a run's matchfinder waits for: 0 none, 2 zh_mf_group, 3 zh_mf_frontier, 4 the splitter
                                                          // (measured, 2 instead of 3: 100 MB of real text 51.9 -> 49.8 ms, configuration 3 31.6 -> 30.5, configuration 4 972 -> 910)
      if (c->stagger_ev != 0 && (c->stagger_ev < 2 || c->stagger_ev > 4)) c->stagger_ev = 3;
      c->first_run_pct = (uint32_t)max(10, min(100, zh_knob("ZULTRA_HIP_FIRST_RUN", 100)));   // share of the first run, in percent of an equal share
      c->last_run_pct = (uint32_t)max(10, min(100, zh_knob("ZULTRA_HIP_LAST_RUN", 100)));     // share of the last run, likewise (three runs and more)
      for (int k = 0; k < c->nlanes; k++) {
         ZH_CHECK(c, hipStreamCreateWithFlags(&c->lane_stream[k], hipStreamNonBlocking));
         for (int i = 0; i < 24; i++) ZH_CHECK(c, hipEventCreate(&c->lane_ev[k][i]));
         {
            // zh_parse_chain is a few workgroups following long chains: it should never queue behind the wide kernels
            int lo_prio = 0, hi_prio = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio);
            ZH_CHECK(c, hipStreamCreateWithPriority(&c->side_stream[k], hipStreamNonBlocking, hi_prio));
         }
         for (int i = 0; i < 8; i++) ZH_CHECK(c, hipEventCreate(&c->side_ev[k][i]));
      }
      ZH_CHECK(c, hipEventCreate(&c->ev_input));
#ifndef ZH_EMU
      if (zh_spread_streams(c) != 0) return -1;
#endif
      if (zh_alloc(c, &c->d_results_compact, B * c->max_subs)) return -1;
      ZH_CHECK(c, hipHostMalloc((void **)&c->h_nsubs, (1 + ZH_MAX_RUNS) * sizeof(uint32_t), 0));
      ZH_CHECK(c, hipHostMalloc((void **)&c->h_scan_out, sizeof(zh_scan_out_t), 0));
      ZH_CHECK(c, hipHostMalloc((void **)&c->h_grids, 2 * ZH_MAX_RUNS * sizeof(uint32_t), 0));
      memset(c->h_nsubs, 0, (1 + ZH_MAX_RUNS) * sizeof(uint32_t));
      ZH_CHECK(c, hipHostMalloc((void **)&c->h_crc, B * sizeof(uint32_t), 0));
      ZH_CHECK(c, hipHostMalloc((void **)&c->h_blocks, B * sizeof(zh_block_t), 0));
      if (c->files_mode) ZH_CHECK(c, hipHostMalloc((void **)&c->h_task_prefix, (B + 1) * sizeof(uint32_t), 0));
      if (c->files_mode) ZH_CHECK(c, hipHostMalloc((void **)&c->h_file_off, (B + 1) * sizeof(uint64_t), 0));
      ZH_CHECK(c, hipHostMalloc((void **)&c->h_segs, B * c->segs_per_block * sizeof(zh_seg_t), 0));
      ZH_CHECK(c, hipHostMalloc((void **)&c->h_ntasks, 2 * ZH_NCNT * sizeof(uint32_t), 0));   // a mirror of d_ntasks + per-run readbacks
      memset(c->h_ntasks, 0, 2 * ZH_NCNT * sizeof(uint32_t));
      ZH_CHECK(c, hipHostMalloc((void **)&c->h_adler, 2 * B * sizeof(uint32_t), 0));
      ZH_CHECK(c, hipHostMalloc((void **)&c->h_results, B * c->max_subs * sizeof(zh_subblock_t), 0));
   }
   c->bar_stride = c->tok_stride / 64;
   c->max_tasks = B * (N / ZH_TASK + c->max_subs);
   // a cut task has at least 2 * ZH_CUT_WARM positions (the floor of ZULTRA_HIP_CUT_MIN) and lies inside one max-block
   c->seg_tasks_per_block = c->files_mode ? 1 : N / (2u * ZH_CUT_WARM) + 1;
   c->seg_items_per_block = c->files_mode ? 1 : N / ZH_CUT_WARM + ZH_CUT_ROWS * (N / (2u * ZH_CUT_WARM) + 1) + 2;   // a task of len positions has at most len / ZH_CUT_LEN + ZH_CUT_ROWS segments
   if (zh_alloc(c, &c->d_bars, B * c->bar_stride) || zh_alloc(c, &c->d_states, B * c->max_subs) || zh_alloc(c, &c->d_taskmap, c->max_tasks) || zh_alloc(c, &c->d_taskinfo, c->max_tasks) ||
       zh_alloc(c, &c->d_nsubs, 1 + ZH_MAX_RUNS) || zh_alloc(c, &c->d_blk_start, B + 1) || zh_alloc(c, &c->d_scan_out, 1) || (c->files_mode && (zh_alloc(c, &c->d_file_off, B + 1) || zh_alloc(c, &c->d_task_prefix, B + 1))) ||
       zh_alloc(c, &c->d_prev3, B * c->segs_per_block * c->sort_stride) || zh_alloc(c, &c->d_runs, B * c->segs_per_block * c->run_stride) ||
       zh_alloc(c, &c->d_segs, B * c->segs_per_block) || zh_alloc(c, &c->d_chunk_ctr, 2 * B * c->segs_per_block + 3 * ZH_MAX_RUNS) || zh_alloc(c, &c->d_ntasks, ZH_NCNT) || zh_alloc(c, &c->d_hugelist, 4 * c->max_tasks) ||
       zh_alloc(c, &c->d_segtasks, B * c->seg_tasks_per_block) || zh_alloc(c, &c->d_segwaves, B * c->seg_items_per_block) || zh_alloc(c, &c->d_segitems, B * c->seg_items_per_block) ||
       zh_alloc(c, &c->d_vecs, B * c->seg_items_per_block * 2 * ZH_VEC) || zh_alloc(c, &c->d_hist_part, c->max_tasks * ZH_NSYM) || zh_alloc(c, &c->d_task_bits, c->max_tasks))
      return -1;
   if (zh_alloc(c, &c->d_data, c->data_cap + 64) || zh_alloc(c, &c->d_blocks, B) || zh_alloc(c, &c->d_sort_a, B * c->segs_per_block * c->sort_stride) ||
       zh_alloc(c, &c->d_sort_b, B * c->segs_per_block * c->sort_stride) || zh_alloc(c, &c->d_match, B * c->match_stride) || zh_alloc(c, &c->d_longest, 64) ||
       zh_alloc(c, &c->d_pay, (size_t)c->nlanes * zh_min64(c->total_cus, B * c->segs_per_block) * 3 * c->sort_stride) ||   // (the runs' kernels may overlap)
       zh_alloc(c, &c->d_tok_pos, B * c->tok_stride) || zh_alloc(c, &c->d_tok_info, B * c->tok_stride) ||
       zh_alloc(c, &c->d_ntok, B) || zh_alloc(c, &c->d_chunkmax, B * c->chunks_per_block) || zh_alloc(c, &c->d_spanstart, B * c->chunks_per_block) ||
       zh_alloc(c, &c->d_spancnt, B * c->chunks_per_block) || zh_alloc(c, &c->d_split_tok, B * (ZH_MAX_SPLITS + 1)) || zh_alloc(c, &c->d_split_cnt, B) ||
       zh_alloc(c, &c->d_sub_base, B) || zh_alloc(c, &c->d_best, B * c->best_stride) || zh_alloc(c, &c->d_cost, B * c->best_stride) || zh_alloc(c, &c->d_work, B * c->max_subs) ||
       zh_alloc(c, &c->d_results, B * c->max_subs) || zh_alloc(c, &c->d_payload, B * c->slot_stride) ||
       zh_alloc(c, &c->d_items, B * c->max_subs) || zh_alloc(c, &c->d_crc, B) || zh_alloc(c, &c->d_adler, 2 * B) || zh_alloc(c, &c->d_crc_tables, 256 + 1024))
      return -1;
   ZH_CHECK(c, hipMemset(c->d_ntasks, 0, ZH_NCNT * sizeof(uint32_t)));
   c->stream_cap = (size_t)(B * (N + 5 * (N / 65535 + 1) + 8) + 64) & ~(size_t)3;
   ZH_CHECK(c, hipMalloc((void **)&c->d_stream, c->stream_cap + 16));
   {
      // CRC tables: byte table of 0xEDB88320 and the 'append ZH_CRC_SLICE zero bytes' operator, one table per state byte
      std::vector<uint32_t> t(256 + 1024);
      for (uint32_t i = 0; i < 256; i++) {
         uint32_t v = i;
         for (int k = 0; k < 8; k++) v = (v >> 1) ^ ((v & 1) ? 0xEDB88320u : 0);
         t[i] = v;
      }
      for (int b = 0; b < 4; b++)
         for (uint32_t i = 0; i < 256; i++) {
            uint32_t v = i << (8 * b);
            for (int k = 0; k < ZH_CRC_SLICE; k++) v = (v >> 8) ^ t[v & 0xff];
            t[256 + 256 * b + i] = v;
         }
      ZH_CHECK(c, hipMemcpy(c->d_crc_tables, t.data(), t.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
   }
   ZH_CHECK(c, hipHostMalloc((void **)&c->h_payload, B * c->slot_stride, 0));
   (void)N;
   return 0;
}

static zultra_hip_ctx_t *zh_create(int device, uint32_t max_block, uint32_t max_blocks, int files_mode) {
   int ndev = zultra_hip_device_count();
   if (ndev <= 0 || device < 0 || device >= ndev || max_blocks == 0) return NULL;
   zultra_hip_ctx_t *c = new zultra_hip_ctx_s();
   memset((void *)&c->timing, 0, sizeof(c->timing));
   c->device = device;
   c->files_mode = files_mode;
   c->max_block = max_block;
   c->max_blocks = max_blocks;
   c->max_subs = files_mode ? 1u : (uint32_t)ZH_MAX_SPLITS;
   c->W = (uint64_t)c->max_block + (files_mode ? 0u : (uint32_t)ZH_HISTORY);
   if (c->W <= ZH_SEG_WINDOW) {   // a max-block is one matchfinder segment
      c->seg_n = c->max_block;
      c->segs_per_block = 1;
      c->seg_W = (uint32_t)c->W;
   }
   else {
      c->seg_n = ZH_SEG_POSITIONS;
      c->segs_per_block = (c->max_block + ZH_SEG_POSITIONS - 1) / ZH_SEG_POSITIONS;
      c->seg_W = ZH_SEG_WINDOW;
   }
   c->sort_stride = ((uint64_t)c->seg_W + 63) & ~63ull;
   c->run_stride = c->sort_stride + 576;   // start[Q] length[Q] first[256] end[256] count, Q = W/4 + 1
   c->match_stride = (uint64_t)c->max_block * ZH_NMATCH;
   c->tok_stride = ((uint64_t)c->max_block + 63) & ~63ull;
   c->chunks_per_block = (c->max_block + ZH_TOK_CHUNK_SMALL - 1) / ZH_TOK_CHUNK_SMALL;   // (what the per-chunk arrays hold: the small chunks of a small batch)
   c->best_stride = c->tok_stride;
   c->slot_stride = (((uint64_t)c->max_block + 64 * c->max_subs + 64) + 63) & ~63ull;
   c->data_cap = (size_t)c->W + (size_t)(max_blocks - 1) * c->max_block;
   c->err[0] = 0;
   if (zh_create_buffers(c) != 0) {
      fprintf(stderr, "zultra_hip_create: %s\n", c->err);
      zultra_hip_destroy(c);
      return NULL;
   }
   return c;
}

extern "C" zultra_hip_ctx_t *zultra_hip_create(int device, uint32_t max_block_size, uint32_t max_blocks) {
   return zh_create(device, zh_clamp_block(max_block_size), max_blocks, 0);
}

extern "C" zultra_hip_ctx_t *zultra_hip_create_files(int device, uint32_t max_file_size, uint32_t max_files) {
   if (max_file_size == 0 || max_file_size >= 8192) return NULL;   // larger inputs can be split: use the block interface
   zultra_hip_ctx_t *c = zh_create(device, (max_file_size + 63u) & ~63u, max_files, 1);
   if (c) c->max_file_size = max_file_size;
   return c;
}

// Pinned host buffers owned by the context (which = 0: input staging, 1: output staging), grown on demand and kept for
// the context's lifetime: host<->device copies from pageable memory run at a fraction of the PCIe rate and block the host.
extern "C" void *zultra_hip_staging(zultra_hip_ctx_t *c, int which, size_t size) {
   if (!c || which < 0 || which > 1) return NULL;
   if (c->h_stage_size[which] < size) {
      if (hipSetDevice(c->device) != hipSuccess) return NULL;
      if (c->h_stage[which]) (void)hipHostFree(c->h_stage[which]);
      c->h_stage[which] = NULL;
      c->h_stage_size[which] = 0;
      if (hipHostMalloc((void **)&c->h_stage[which], size, 0) != hipSuccess) return NULL;
      c->h_stage_size[which] = size;
   }
   return c->h_stage[which];
}

extern "C" const char *zultra_hip_last_error(const zultra_hip_ctx_t *c) { return c ? c->err : "no context"; }
extern "C" void zultra_hip_ctx_info(const zultra_hip_ctx_t *c, int *device, uint32_t *max_block_size, uint32_t *max_blocks, size_t *device_bytes) {
   if (!c) return;
   if (device) *device = c->device;
   if (max_block_size) *max_block_size = c->max_block;
   if (max_blocks) *max_blocks = c->max_blocks;
   if (device_bytes) *device_bytes = c->device_bytes;
}

// Device bytes a context on `device` for batches of `max_blocks` max-blocks of `max_block_size` bytes allocates: the strides and the list
// of arrays of zh_create / zh_create_buffers (the layout comment at the top of this file) restated without allocating — a second
// formula, held to the real layout by test_context_cache_and_size_estimate (-5 % .. +10 %); files mode (one sub-block per block)
// allocates less than this. The device matters for one term only: the matchfinder's payload, per compute unit.
extern "C" size_t zultra_hip_context_bytes_on(int device, uint32_t max_block_size, uint32_t max_blocks) {
   const uint64_t N = zh_clamp_block(max_block_size), B = max_blocks;
   const uint64_t W = N + ZH_HISTORY;
   const uint64_t seg_W = W <= ZH_SEG_WINDOW ? W : (uint64_t)ZH_SEG_WINDOW;
   const uint64_t S = W <= ZH_SEG_WINDOW ? 1 : (N + ZH_SEG_POSITIONS - 1) / ZH_SEG_POSITIONS;
   const uint64_t sort_stride = (seg_W + 63) & ~63ull, run_stride = sort_stride + 576, tok_stride = (N + 63) & ~63ull;
   const uint64_t slot_stride = ((N + 64 * ZH_MAX_SPLITS + 64) + 63) & ~63ull, cpb = (N + ZH_TOK_CHUNK_SMALL - 1) / ZH_TOK_CHUNK_SMALL;
   const uint64_t tasks = B * (N / ZH_TASK + ZH_MAX_SPLITS), subs = B * ZH_MAX_SPLITS;
   uint64_t bytes = 0;
   bytes += W + (B - 1) * N + 64;                                    // d_data
   bytes += B * S * sort_stride * (4 + 4 + 8) + B * S * run_stride * 4;   // sort ping-pong, prev records, run tables
   bytes += B * N * ZH_NMATCH * sizeof(zh_match_t) + B * tok_stride * (4 + 2 + 4 + 2);   // rows, token chain, parse, costs
   bytes += B * (tok_stride / 64) * 8;                               // barrier bitmap
   bytes += B * slot_stride;                                         // payload slots
   bytes += (B * (N + 5 * (N / 65535 + 1) + 8) + 80);                // stitched stream
   bytes += subs * (sizeof(zh_sbstate_t) + sizeof(zh_work_t) + 2 * sizeof(zh_subblock_t) + sizeof(zh_stitch_item_t));
   bytes += tasks * (2 * sizeof(uint2) + 4 * 4 + 4 + ZH_NSYM * 4);   // task map and ranges, chain lists, bit counts, histograms
   bytes += B * (S * (sizeof(zh_seg_t) + 8) + sizeof(zh_block_t) + cpb * 12 + (ZH_MAX_SPLITS + 1) * 4 + 6 * 4) + 8192;
   {
      // payload of the matchfinder's refining passes: per run (ZULTRA_HIP_STREAMS) and persistent workgroup (one per CU)
      int cus = 0;
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
      const uint64_t lanes = (uint64_t)zh_env_lanes();   // (what zh_create_buffers allocates for: 0 or unset = four)
      bytes += lanes * zh_min64((uint64_t)cus, B * S) * 3 * sort_stride * 4;
      // cut tasks (zh_parse.h): lists and two cost vectors per segment
      const uint64_t seg_tasks = N / (2u * ZH_CUT_WARM) + 1, seg_items = N / ZH_CUT_WARM + ZH_CUT_ROWS * (N / (2u * ZH_CUT_WARM) + 1) + 2;
      bytes += B * (seg_tasks * sizeof(uint4) + seg_items * (2 * sizeof(uint2) + 2 * ZH_VEC * sizeof(int16_t)));
   }
   return (size_t)bytes;
}
extern "C" size_t zultra_hip_context_bytes(uint32_t max_block_size, uint32_t max_blocks) {   // on the calling thread's current device
   int dev = 0;
   if (hipGetDevice(&dev) != hipSuccess) dev = 0;
   return zultra_hip_context_bytes_on(dev, max_block_size, max_blocks);
}
extern "C" size_t zultra_hip_data_capacity(const zultra_hip_ctx_t *c) { return c ? c->data_cap : 0; }

// Cuts the max-blocks of a batch into matchfinder segments (zh_common.h) and uploads the list.
static int zh_build_segments(zultra_hip_ctx_t *c, const zultra_hip_block_t *blocks, uint32_t nblocks, hipStream_t st) {
   c->segs.clear();
   c->seg_base.resize(nblocks + 1);
   for (uint32_t b = 0; b < nblocks; b++) {
      c->seg_base[b] = (uint32_t)c->segs.size();
      const zultra_hip_block_t &blk = blocks[b];
      if ((uint64_t)blk.prev + blk.n <= ZH_SEG_WINDOW && c->segs_per_block == 1) {
         zh_seg_t sg = {blk.win_off, blk.prev, blk.n, 0, b, 0};
         c->segs.push_back(sg);
         continue;
      }
      if ((uint64_t)blk.prev + blk.n <= ZH_SEG_WINDOW) {   // a short max-block of a large-block context
         zh_seg_t sg = {blk.win_off, blk.prev, blk.n, 0, b, 0};
         c->segs.push_back(sg);
         continue;
      }
      for (uint32_t pos = 0; pos < blk.n;) {
         const uint32_t len = blk.n - pos < c->seg_n ? blk.n - pos : c->seg_n;
         const uint32_t hist = blk.prev + pos < ZH_HISTORY ? blk.prev + pos : (uint32_t)ZH_HISTORY;
         const uint32_t rest = blk.n - pos - len;
         zh_seg_t sg = {blk.win_off + blk.prev + pos - hist, hist, len, rest < ZH_MAX_MATCH ? rest : (uint32_t)ZH_MAX_MATCH, b, pos};
         c->segs.push_back(sg);
         pos += len;
      }
   }
   c->seg_base[nblocks] = (uint32_t)c->segs.size();
   if (c->segs.size() > (size_t)c->max_blocks * c->segs_per_block) {
      snprintf(c->err, sizeof(c->err), "segment list overflows");
      return -1;
   }
   memcpy(c->h_segs, c->segs.data(), c->segs.size() * sizeof(zh_seg_t));
   ZH_CHECK(c, hipMemcpyAsync(c->d_segs, c->h_segs, c->segs.size() * sizeof(zh_seg_t), hipMemcpyHostToDevice, st));
   return 0;
}

// tasks a wave of zh_parse_lanes takes (zh_parse_lanes.h): as many as ZH_LP_TASKS, as few as it takes to fill the chip's wave slots
static uint32_t zh_tasks_per_wave(const zultra_hip_ctx_t *c, uint32_t ntasks, bool last_run) {
   if (last_run && c->lane_tasks_last) return c->lane_tasks_last;
   if (c->lane_tasks) return c->lane_tasks;   // ZULTRA_HIP_LANE_TASKS, read once at context creation
   const uint32_t slots = c->total_cus * 8u;   // (measured: eight tasks per wave at 16 K tasks per run beat four by 2 % of the step)
   return max(1u, min((uint32_t)ZH_LP_TASKS, (ntasks + slots - 1) / slots));
}

// barrier bitmap and greedy token chain of `nb` max-blocks starting at batch block b0, in chunks (zh_split.h)
static int zh_enqueue_tokenize(zultra_hip_ctx_t *c, hipStream_t st, const zh_block_t *blk, uint32_t b0, uint32_t nb) {
   // (chunks of an eighth for a BATCH of a few max-blocks — all its runs alike: they share the per-chunk arrays, which are sized for the small chunks; files mode:
   // an input is one chunk)
   const uint32_t chunk = (!c->files_mode && c->nblocks <= ZH_TOK_SMALL_BATCH) ? (uint32_t)ZH_TOK_CHUNK_SMALL : (uint32_t)ZH_TOK_CHUNK;
   const uint32_t cpb = (c->max_block + chunk - 1) / chunk;
   const uint32_t *match = (const uint32_t *)(c->d_match + (uint64_t)b0 * c->match_stride);   // slot 0 of a position's row = its longest match
   uint64_t *bars = c->d_bars + (uint64_t)b0 * c->bar_stride;
   uint32_t *tp = c->d_tok_pos + (uint64_t)b0 * c->tok_stride;
   uint16_t *ti = c->d_tok_info + (uint64_t)b0 * c->tok_stride;
   uint32_t *cmax = c->d_chunkmax + (uint64_t)b0 * cpb, *sstart = c->d_spanstart + (uint64_t)b0 * cpb, *scnt = c->d_spancnt + (uint64_t)b0 * cpb;
   uint32_t *slot0 = c->d_best + (uint64_t)b0 * c->best_stride;   // (free until the run's first parse pass: zh_split.h)
   ZH_LAUNCH(zh_barriers, nb * cpb, 64, st, blk, match, c->match_stride, bars, c->bar_stride, cmax, cpb, chunk, slot0, c->best_stride);
   if (cpb > 1) ZH_LAUNCH(zh_barriers_fix, (nb + 63) / 64, 64, st, blk, nb, bars, c->bar_stride, (const uint32_t *)cmax, cpb, chunk);
   ZH_LAUNCH(zh_tokenize_spans, nb * cpb, 64, st, c->cur_data, blk, (const uint32_t *)slot0, c->best_stride, tp, ti, c->tok_stride, (const uint64_t *)bars, c->bar_stride, sstart, scnt,
             cpb, chunk);
   ZH_LAUNCH(zh_tokens_compact, nb, ZH_COMPACT_THREADS, st, blk, tp, ti, c->tok_stride, (const uint32_t *)sstart, (const uint32_t *)scnt, cpb, chunk, c->d_ntok + b0);
   return 0;
}

// files mode: what zh_split would report for an input below its 8192-byte threshold — one sub-block spanning all tokens
__global__ void zh_nosplit(uint32_t nblocks, const uint32_t *__restrict__ ntok, uint32_t *split_tok, uint32_t *split_cnt) {
   const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
   if (b >= nblocks) return;
   split_tok[(uint64_t)b * (ZH_MAX_SPLITS + 1)] = 0;
   split_tok[(uint64_t)b * (ZH_MAX_SPLITS + 1) + 1] = ntok[b];
   split_cnt[b] = 1;
}

// The kernel sequence of one run of a batch — max-blocks b0 .. b0 + nb with `total_n` input bytes, matchfinder segments sg0 .. sg0 + nsg — with
// NO HOST DECISION in it: run k on stream st, its chains on `side` (a cut task's segments, when a run has many, in the first workgroups of zh_parse_lanes' grid; not in files
// mode: inputs below 8192 bytes are never cut). Everything the splitter decides — how many sub-blocks, hence how many tasks, chains, segments — stays on the device: zh_plan_subblocks
// sums it up into the run's counters, every later kernel takes its bounds from there, and the grids here are sized from the input bytes alone, as
// bounded grids that stride over what there is (rounds 1-4 read the counts back twice per run, in the middle of the pipeline: 0.8-2.2 ms each on the
// 100 MB step, profiles/r04_timeline_c2.txt). Per-block buffers are addressed as base + block * stride, so a run sees the base pointers advanced to its
// first max-block; sub-block and task indices are local to the run (zh_compact_results shifts the descriptors).
// part: 0 the whole run; 1 only up to the end of its first matchfinder kernel (where the next run's matchfinder may start), 2 only what follows — the two
// halves of a files-mode run captured as graphs of their own (zh_run_files).
// A kernel whose item count only the device knows (sub-blocks, tasks, segment waves of a run): the <false> form over `grid_` workgroups, one item each (what data
// usually gives; surplus workgroups leave at once), then the <true> form, ZH_MORE_GRID workgroups striding over what lies beyond grid_ — nearly always nothing.
// (bound_: what the item count cannot exceed — a grid that covers it needs no second launch: a call on a few max-blocks is a matter of launches)
#define ZH_MORE_GRID 256u
#define ZH_LAUNCH_BOTH(kernel_, grid_, bound_, stream_, ...)                                                                  \
   do {                                                                                                                       \
      ZH_LAUNCH(kernel_<false>, (grid_), 64, stream_, __VA_ARGS__, 0u);                                                       \
      if ((uint64_t)(grid_) < (uint64_t)(bound_) && !no_more) ZH_LAUNCH(kernel_<true>, ZH_MORE_GRID, 64, stream_, __VA_ARGS__, (uint32_t)(grid_)); \
   } while (0)

static int zh_enqueue_run(zultra_hip_ctx_t *c, int k, uint32_t b0, uint32_t nb, uint64_t total_n, uint32_t max_n, uint32_t sg0, uint32_t nsg, hipStream_t st, hipStream_t side, int part) {
   const bool files = c->files_mode != 0;
   hipEvent_t *ev = c->lane_ev[k];
   const zh_block_t *blk = c->d_blocks + b0;
   const uint64_t tasks_per_block = c->max_tasks / c->max_blocks;
   const uint64_t s0 = (uint64_t)b0 * c->max_subs, t0 = (uint64_t)b0 * tasks_per_block;   // per-sub-block and per-task buffers of the run start at its worst-case offset
   const zh_seg_t *sgs = c->d_segs + sg0;
   uint32_t *sa = c->d_sort_a + (uint64_t)sg0 * c->sort_stride, *sb = c->d_sort_b + (uint64_t)sg0 * c->sort_stride;
   uint2 *p3 = c->d_prev3 + (uint64_t)sg0 * c->sort_stride;
   uint32_t *rn = c->d_runs + (uint64_t)sg0 * c->run_stride;
   uint32_t *ctr = c->d_chunk_ctr + (size_t)sg0 * 2 + 3 * (size_t)k;   // this run's counters: 2 per segment + the three tickets
   uint32_t *pay = c->d_pay + (size_t)k * zh_min64(c->total_cus, (uint64_t)c->max_blocks * c->segs_per_block) * 3 * c->sort_stride;
   uint32_t *cnt = c->d_ntasks + (size_t)k * ZH_CNT_STRIDE;   // the run's counters (ZH_CNT_*)
   zh_work_t *work = c->d_work + s0;
   zh_sbstate_t *states = c->d_states + s0;
   uint2 *taskmap = c->d_taskmap + t0;
   uint2 *taskinfo = c->d_taskinfo + t0;
   uint32_t *hist_part = c->d_hist_part + t0 * ZH_NSYM, *task_bits = c->d_task_bits + t0;
   uint32_t *hugelist = c->d_hugelist + 4 * t0;   // four lists of `cap` entries each: three by zh_list_huge, the cut tasks given up on by the segment workgroups
   const uint32_t cap = (uint32_t)zh_min64((uint64_t)nb * tasks_per_block, 0xFFFFFFFFull);
   uint4 *segtasks = c->d_segtasks + (uint64_t)b0 * c->seg_tasks_per_block;
   uint2 *segwaves = c->d_segwaves + (uint64_t)b0 * c->seg_items_per_block, *segitems = c->d_segitems + (uint64_t)b0 * c->seg_items_per_block;
   int16_t *vecs = c->d_vecs + (uint64_t)b0 * c->seg_items_per_block * 2 * ZH_VEC;
   uint8_t *payload = c->d_payload + (uint64_t)b0 * c->slot_stride;
   uint32_t *best = c->d_best + (uint64_t)b0 * c->best_stride;
   uint16_t *cost = c->d_cost + (uint64_t)b0 * c->best_stride;
   const uint64_t *bars = c->d_bars + (uint64_t)b0 * c->bar_stride;
   const zh_match_t *match = c->d_match + (uint64_t)b0 * c->match_stride;
   const uint32_t mf_grid = min(nsg, max(1u, c->num_cus * c->mf_cu_pct / 100u));   // persistent workgroups, one per CU (zh_matchfinder.h)
   // grids: bounded by what the input bytes allow, sized for what data usually gives; the kernels stride
   const uint32_t est_tasks = (uint32_t)zh_min64(cap, total_n / ZH_TASK + 2ull * nb);                                   // tasks: ~ bytes / 2048 + one per sub-block
   uint32_t task_grid = cap <= 2048u ? cap : (uint32_t)zh_min64(cap, total_n / ZH_TASK + 4ull * nb);               // one wave per task (zh_list_huge, zh_post_tasks, zh_emit_tasks)
   const uint64_t sb_bound = (uint64_t)nb * c->max_subs;
   uint32_t sb_grid = (uint32_t)zh_min64(sb_bound, zh_max64(4ull * nb, 1024));   // one wave per sub-block (zh_sb_init, zh_sb_build)
   if (c->grid_cap) {   // (tests: everything beyond the cap goes through the <true> forms)
      task_grid = max(1u, min(task_grid, c->grid_cap));
      sb_grid = max(1u, min(sb_grid, c->grid_cap));
   }
   // no <true> overflow forms where the context's last batch stayed inside these grids (zh_parse.h, zh_run_is_void): the grids go into the run's counters, a run that
   // outgrows them is void and the batch is run again with the forms
   const bool no_more = !files && c->chain_skip && c->chain_seen_runs == c->last_runs && c->seen_nsubs[k] <= sb_grid && c->seen_ntasks[k] <= task_grid &&
                        ((uint64_t)sb_grid < sb_bound || (uint64_t)task_grid < (uint64_t)cap);
   c->run_nomore[k] = no_more;
   c->run_grids[k][0] = sb_grid;
   c->run_grids[k][1] = task_grid;
   if (no_more) {
      c->h_grids[2 * k] = sb_grid;
      c->h_grids[2 * k + 1] = task_grid;
      ZH_CHECK(c, hipMemcpyAsync(cnt + ZH_CNT_SBGRID, c->h_grids + 2 * k, 2 * sizeof(uint32_t), hipMemcpyHostToDevice, st));   // (behind the clearing of the run's counters, on this stream)
   }
   const uint64_t seg_bound = (uint64_t)nb * c->seg_items_per_block;                     // entries of segwaves (zh_list_huge)
   if (part != 2) {
      ZH_LAUNCH_LDS(zh_mf_group<true>, mf_grid, ZH_MF_THREADS, ZH_MF_GROUP_LDS, st, c->cur_data, sgs, sa, sb, p3, rn, c->sort_stride, c->run_stride, 0, nsg, ctr + (size_t)nsg * 2 + 1, pay,
                    c->mf_lds_cap);
      // the bigram classes that fit no chunk of zh_mf_group, noted by it (zh_mf_group_lds.h): a kernel of their own (inputs of <= 4 KiB are one chunk: nothing is ever noted)
      if (c->mf_lds_cap && c->seg_W > (files ? (uint32_t)ZH_MFL_MAXCAP : min((uint32_t)ZH_MFL_MAXCAP, max(c->mf_lds_cap, 16u))))
         ZH_LAUNCH_LDS(zh_mf_group_big, files ? min(mf_grid, 32u) : mf_grid, ZH_MF_THREADS, ZH_MF_GROUP_LDS, st, c->cur_data, sgs, sa, sb, p3, (const uint32_t *)rn, c->sort_stride, c->run_stride, nsg,
                       ctr + (size_t)nsg * 2 + 2, pay);
   }
   if (part == 1) return 0;
   if (part == 0) ZH_CHECK(c, hipEventRecord(ev[2], st));   // the next run's matchfinder starts here (part 2: the caller records it between the two graphs)
   // (segment descriptors carry batch-wide block indices: the rows go to d_match + block * match_stride)
   // a run of fewer segments than CUs (one call on a few max-blocks: latency): the workgroups beyond one per segment find the tickets gone and help —
   // a segment of a 64 KiB max-block is ~1500 chunks, shared while a helper's share stays above ZH_MF_HELP_MIN of them (small inputs: nothing worth sharing)
   const uint32_t fr_grid = files ? mf_grid : min(max(1u, c->num_cus * c->mf_cu_pct / 100u), mf_grid * 8u);
   ZH_LAUNCH_LDS(zh_mf_frontier<true>, fr_grid, ZH_MF_THREADS, ZH_MF_FRONTIER_LDS, st, c->cur_data, sgs, (const uint32_t *)sa, (const uint2 *)p3, (const uint32_t *)rn, c->sort_stride,
                 c->run_stride, c->d_match, c->match_stride, c->d_longest, c->tok_stride, ctr, nsg, files ? 0u : 1u);
   if (!files) ZH_CHECK(c, hipEventRecord(ev[3], st));   // (timing marks)
   if (zh_enqueue_tokenize(c, st, blk, b0, nb) != 0) return -1;
   if (files)
      ZH_LAUNCH(zh_nosplit, (nb + 255) / 256, 256, st, nb, (const uint32_t *)(c->d_ntok + b0), c->d_split_tok + (uint64_t)b0 * (ZH_MAX_SPLITS + 1), c->d_split_cnt + b0);
   else {
#define ZH_LAUNCH_SPLIT(W_)                                                                                                                                   \
   ZH_LAUNCH(zh_split<W_>, nb, 64 * W_, st, blk, (const uint32_t *)(c->d_tok_pos + b0 * c->tok_stride), (const uint16_t *)(c->d_tok_info + b0 * c->tok_stride), \
             c->tok_stride, (const uint32_t *)(c->d_ntok + b0), c->d_split_tok + (uint64_t)b0 * (ZH_MAX_SPLITS + 1), c->d_split_cnt + b0)
      const uint32_t sw = c->split_waves ? c->split_waves : (max_n > 131072 ? 16u : 8u);   // (by the run's largest max-block, not the context's limit)
      if (sw >= 16)
         ZH_LAUNCH_SPLIT(16);
      else if (sw >= 8)
         ZH_LAUNCH_SPLIT(8);
      else if (sw >= 4)
         ZH_LAUNCH_SPLIT(4);
      else
         ZH_LAUNCH_SPLIT(2);
#undef ZH_LAUNCH_SPLIT
   }
   if (!files) ZH_CHECK(c, hipEventRecord(ev[4], st));   // (timing marks)
   // ---- stage 3: the sub-block coder, one kernel per step over the run (zh_encode.h) -----------------------------------------------------
#define ZH_LAUNCH_PLAN(T_)                                                                                                                                                               \
   ZH_LAUNCH(zh_plan_subblocks<T_>, 1, T_, st, blk, nb, (const uint32_t *)(c->d_tok_pos + (uint64_t)b0 * c->tok_stride), c->tok_stride, (const uint32_t *)(c->d_ntok + b0),              \
             (const uint32_t *)(c->d_split_tok + (uint64_t)b0 * (ZH_MAX_SPLITS + 1)), (const uint32_t *)(c->d_split_cnt + b0), c->d_sub_base + b0, c->slot_stride, work, taskmap, cnt)
   if (files)   // one sub-block per input, the task ranges from the sizes the host was handed (zh_plan_files)
      ZH_LAUNCH(zh_plan_files, (nb + ZH_PLAN_FILES_THREADS - 1) / ZH_PLAN_FILES_THREADS, ZH_PLAN_FILES_THREADS, st, blk, nb, (const uint32_t *)(c->d_ntok + b0), (const uint32_t *)(c->d_task_prefix + b0),
                c->d_sub_base + b0, c->slot_stride, work, taskmap, cnt);
   else if (nb > 4096u)
      ZH_LAUNCH_PLAN(1024u);
   else
      ZH_LAUNCH_PLAN(256u);
#undef ZH_LAUNCH_PLAN
   ZH_LAUNCH_BOTH(zh_sb_init, sb_grid, sb_bound, st, (const uint16_t *)(c->d_tok_info + (uint64_t)b0 * c->tok_stride), c->tok_stride, (const zh_work_t *)work, states, (const uint32_t *)cnt);
   // (inputs of a files batch are never cut into speculative segments: seg_min = all ones; a run of at most coop_tasks tasks counts as small)
   ZH_LAUNCH_BOTH(zh_list_huge, task_grid, cap, st, blk, bars, c->bar_stride, (const zh_work_t *)work, (const uint2 *)taskmap, (const uint32_t *)(c->d_match + (uint64_t)b0 * c->match_stride),
             c->match_stride, hugelist, cap, segtasks, segitems, segwaves, files ? 0xFFFFFFFFu : c->cut_min, files ? (uint32_t)ZH_CUT_LEN : c->cut_len, cnt, taskinfo, (uint32_t)ZH_COOP_MIN,
             files ? (uint32_t)ZH_COOP_MIN : c->coop_small, files ? 0u : c->coop_tasks);
   if (!files) ZH_CHECK(c, hipEventRecord(ev[5], st));   // (timing marks)
   // Persistent workgroups of zh_parse_chain take the listed chains from a ticket (none listed: they leave at once); zh_parse_lanes takes the task
   // list in groups, as a grid that fills the chip's wave slots — next to chains only `lane_waves` per CU stay, so that the chain workgroups find
   // room the moment they are launched (the run's counters tell the kernel which); the first workgroups of zh_parse_lanes' grid take the cut tasks' segments when there are many.
   const uint32_t tpw = zh_tasks_per_wave(c, est_tasks, !files && c->last_runs > 1 && k == c->last_runs - 1);
   const uint32_t lane_grid = max(1u, min((est_tasks + tpw - 1) / tpw, c->num_cus * 16u));
   // (two chain workgroups fit a CU — 169 registers, four waves — and they are persistent: a third per CU would only queue behind them, find the tickets
   // gone and leave; and in a run without chains every workgroup of this grid has to find a slot among the quad kernel's waves before the pass can end)
   uint32_t chain_grid = files ? min(nb, c->files_chain_grid) : (uint32_t)zh_min64(zh_min64(ZH_CHAIN_GRID, 2u * c->num_cus), total_n / 256u + nb);
   // A stream without chains must not pay for them (round 6; zh_parse.h, zh_run_is_void): a run whose counterpart in the context's last batch listed nothing for
   // zh_parse_chain gets no chain kernel at all — no fork, no grid to schedule among the quad kernels' waves, no join — and a mark in its counters that says so.
   const bool chains_idle = !files && c->run_nochains[k];
   if (chains_idle) ZH_CHECK(c, hipMemsetAsync(cnt + ZH_CNT_NOCHAINS, 1, sizeof(uint32_t), st));   // (non-zero: zh_run_is_void; the run's counters were cleared on this stream before)
   const uint32_t seg_grid = (uint32_t)zh_min64(c->num_cus * 8u, zh_max64(1, seg_bound));
   if (chains_idle) {
      // what the side stream was given at the start of the batch (checksums, clearing the payload slots and the stream buffer) is otherwise joined with the chains
      ZH_CHECK(c, hipEventRecord(c->side_ev[k][1], side));
      ZH_CHECK(c, hipStreamWaitEvent(st, c->side_ev[k][1], 0));
   }
   for (int pass = 0; pass <= 3; pass++) {
      // (a lambda: ZH_LAUNCH returns from the function it stands in)
      auto launch_chain = [&](hipStream_t cs) -> int {
         ZH_LAUNCH(zh_parse_chain, chain_grid, ZH_CHAIN_THREADS, cs, c->cur_data, blk, match, c->match_stride, bars, c->bar_stride, (const zh_work_t *)work, (const uint2 *)taskmap,
                   (const uint32_t *)hugelist, cap, segtasks, (const uint2 *)segitems, vecs, c->seg_wide, c->seg_whole, cnt, (const zh_sbstate_t *)states, best, c->best_stride, hist_part, pass,
                   cnt + ZH_CNT_CHAIN_TICKET + pass, (c->d_chain_trace && !files) ? c->d_chain_trace + 3 * (uint64_t)ZH_TRACE_SLOTS * (4 * k + pass) : (uint64_t *)NULL);
         return 0;
      };
      if (!chains_idle) {
         ZH_CHECK(c, hipEventRecord(c->side_ev[k][2 * pass], st));
         ZH_CHECK(c, hipStreamWaitEvent(side, c->side_ev[k][2 * pass], 0));
         if (launch_chain(side) != 0) return -1;
         ZH_CHECK(c, hipEventRecord(c->side_ev[k][2 * pass + 1], side));
      }
      {
         zh_seg_args_t sg;
         sg.segtasks = segtasks;
         sg.segwaves = (const uint2 *)segwaves;
         sg.vecs = vecs;
         sg.demote_list = hugelist + 3 * (size_t)cap;
         sg.demote_min = c->demote_min;
         sg.seg_wide_min = c->seg_wide;
         sg.seg_grid = files ? 0u : seg_grid;
         ZH_LAUNCH(zh_parse_lanes, sg.seg_grid + lane_grid, 64, st, c->cur_data, blk, match, c->match_stride, bars, c->bar_stride, (const zh_work_t *)work, (const uint2 *)taskmap, cnt,
                   (const zh_sbstate_t *)states, best, c->best_stride, cost, hist_part, pass, cnt + ZH_CNT_TASK_TICKET + pass, (const uint2 *)taskinfo, tpw,
                   files ? 0xFFFFFFFFu : c->num_cus * c->lane_waves, sg);
      }
      if (!chains_idle) ZH_CHECK(c, hipStreamWaitEvent(st, c->side_ev[k][2 * pass + 1], 0));
      if (!files) ZH_CHECK(c, hipEventRecord(ev[6 + 2 * pass], st));   // (timing marks)
      ZH_LAUNCH_BOTH(zh_sb_build, sb_grid, sb_bound, st, (const zh_work_t *)work, states, (const uint32_t *)hist_part, payload, pass, cnt);
      if (!files) ZH_CHECK(c, hipEventRecord(ev[7 + 2 * pass], st));   // (timing marks)
   }
   ZH_LAUNCH_BOTH(zh_post_tasks, task_grid, cap, st, c->cur_data, blk, bars, c->bar_stride, (const zh_work_t *)work, (const uint2 *)taskmap, (const uint32_t *)cnt, (const zh_sbstate_t *)states, best,
             c->best_stride, task_bits, (const uint2 *)taskinfo);
   if (!files) ZH_CHECK(c, hipEventRecord(ev[14], st));   // (timing marks)
   ZH_LAUNCH_BOTH(zh_emit_tasks, task_grid, cap, st, c->cur_data, blk, bars, c->bar_stride, (const zh_work_t *)work, (const uint2 *)taskmap, (const uint32_t *)cnt, (const zh_sbstate_t *)states,
             (const uint32_t *)best, c->best_stride, (const uint32_t *)task_bits, payload, c->d_results + s0, (const uint2 *)taskinfo);
   if (!files) ZH_CHECK(c, hipEventRecord(ev[15], st));   // (timing marks)
   // per-max-block CRC-32 (linear part) and Adler-32 for the framing's footer (a batch of max-blocks computes them next to its matchfinder: zultra_hip_compress_blocks)
   if (files) {
      // (several inputs to a workgroup: zh_crc32_small)
      uint32_t spg = 1;
      while (spg < (c->max_block + ZH_CRC_SLICE - 1) / ZH_CRC_SLICE) spg <<= 1;
      if (spg <= ZH_CRC_THREADS / 2)
         ZH_LAUNCH(zh_crc32_small, (nb + ZH_CRC_THREADS / spg - 1) / (ZH_CRC_THREADS / spg), ZH_CRC_THREADS, st, c->cur_data, blk, nb, (const uint32_t *)c->d_crc_tables, c->d_crc + b0, c->d_adler + 2 * (size_t)b0, spg);
      else
         ZH_LAUNCH(zh_crc32_blocks, nb, ZH_CRC_THREADS, st, c->cur_data, blk, (const uint32_t *)c->d_crc_tables, c->d_crc + b0, c->d_adler + 2 * (size_t)b0);
   }
   return 0;
}

// Behind the last run: the runs' descriptors laid end to end in batch coordinates (zh_compact_results), the totals next to them. On `st`, which has
// waited for every run.
static int zh_enqueue_compact(zultra_hip_ctx_t *c, const uint32_t *run_b0, int runs, hipStream_t st) {
   zh_runs_t R;
   memset(&R, 0, sizeof(R));
   R.nruns = (uint32_t)runs;
   for (int k = 0; k < runs; k++) R.b0[k] = run_b0[k];
   R.cnt_stride = ZH_CNT_STRIDE;
   R.nsubs_field = ZH_CNT_NSUBS;
   R.max_subs = c->max_subs;
   R.slot_stride = c->slot_stride;
   ZH_LAUNCH(zh_compact_results, 64, ZH_COMPACT_RESULTS_THREADS, st, R, (const zh_subblock_t *)c->d_results, (const uint32_t *)c->d_ntasks, c->d_results_compact, c->d_nsubs);
   ZH_CHECK(c, hipMemcpyAsync(c->h_nsubs, c->d_nsubs, (1 + ZH_MAX_RUNS) * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
   ZH_CHECK(c, hipMemcpyAsync(c->h_ntasks, c->d_ntasks, ZH_NCNT * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
   return 0;
}

// A files-mode batch: runs of inputs on streams of their own, each run's matchfinder behind the previous run's sort, exactly as a
// batch of max-blocks (zultra_hip_compress_blocks) — but with nothing for the host to decide in between. On stream st0 and the
// streams forked from it; captured into a hipGraph the first time a (batch size, input pointer) pair is seen and replayed
// afterwards: one graph launch per batch.
static uint32_t zh_files_run_lo(const zultra_hip_ctx_t *c, uint32_t nblocks, int k) { return (uint32_t)((uint64_t)nblocks * (uint64_t)k / (uint64_t)c->last_runs); }

// what follows the runs of a files batch, on the stream that has waited for them: the descriptors in batch order, and the copies of what the host reads
static int zh_enqueue_files_tail(zultra_hip_ctx_t *c, uint32_t nblocks, hipStream_t st0) {
   uint32_t run_b0[ZH_MAX_RUNS];
   for (int k = 0; k < c->last_runs; k++) run_b0[k] = zh_files_run_lo(c, nblocks, k);
   if (zh_enqueue_compact(c, run_b0, c->last_runs, st0) != 0) return -1;
   ZH_CHECK(c, hipMemcpyAsync(c->h_adler, c->d_adler, 2 * (size_t)nblocks * sizeof(uint32_t), hipMemcpyDeviceToHost, st0));
   ZH_CHECK(c, hipMemcpyAsync(c->h_results, c->d_results_compact, (size_t)nblocks * sizeof(zh_subblock_t), hipMemcpyDeviceToHost, st0));   // (one sub-block per input)
   ZH_CHECK(c, hipMemcpyAsync(c->h_crc, c->d_crc, (size_t)nblocks * sizeof(uint32_t), hipMemcpyDeviceToHost, st0));
   // The assembly of the inputs' streams (zh_stitch_scan in files form, zh_stitch) goes out with the batch (round 6; part of the captured graph where there is one): every
   // input starts on a byte boundary, nothing depends on what the caller does between the batch and zultra_hip_stitch_files — which then only hands out the offsets read
   // back here. (Before: a second call, a second synchronisation, 0.4 ms between the two on 65 536-input batches.) The clearing covers the worst case of the batch's
   // input COUNT — a replayed graph sees other sizes.
   ZH_CHECK(c, hipMemsetAsync(c->d_stream, 0, (size_t)zh_min64((uint64_t)c->stream_cap + 16, ((uint64_t)nblocks * ((uint64_t)c->max_block + 16) + 16 + 3) & ~3ull), st0));
   if (zh_enqueue_stitch(c, st0, 0, -1, 1, false, false) != 0) return -1;
   ZH_CHECK(c, hipMemcpyAsync(c->h_file_off, c->d_file_off, ((size_t)nblocks + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, st0));
   return 0;
}

static int zh_enqueue_files(zultra_hip_ctx_t *c, uint32_t nblocks, hipStream_t st0) {
   const int runs = c->last_runs;
   // what the runs need cleared, and (below) the copies of what they produce, on the stream the runs fork from
   ZH_CHECK(c, hipMemsetAsync(c->d_chunk_ctr, 0, ((size_t)nblocks * 2 + 3 * (size_t)runs) * sizeof(uint32_t), st0));
   ZH_CHECK(c, hipMemsetAsync(c->d_ntasks, 0, ZH_NCNT * sizeof(uint32_t), st0));
   ZH_CHECK(c, hipMemsetAsync(c->d_payload, 0, (size_t)nblocks * c->slot_stride, st0));
   ZH_CHECK(c, hipEventRecord(c->ev2[1], st0));   // fork
   for (int k = 0; k < runs; k++) {
      hipStream_t st = k ? c->lane_stream[k] : st0;
      if (k) {
         ZH_CHECK(c, hipStreamWaitEvent(st, c->ev2[1], 0));
         if (c->stagger_ev) ZH_CHECK(c, hipStreamWaitEvent(st, c->lane_ev[k - 1][2], 0));
      }
      const uint32_t b0 = zh_files_run_lo(c, nblocks, k), b1 = zh_files_run_lo(c, nblocks, k + 1);
      if (zh_enqueue_run(c, k, b0, b1 - b0, (uint64_t)(b1 - b0) * c->max_block, c->max_block, b0, b1 - b0, st, c->side_stream[k], 0) != 0) return -1;
      if (k) ZH_CHECK(c, hipEventRecord(c->lane_ev[k][17], st));
   }
   for (int k = 1; k < runs; k++) ZH_CHECK(c, hipStreamWaitEvent(st0, c->lane_ev[k][17], 0));   // join
   return zh_enqueue_files_tail(c, nblocks, st0);
}

static int zh_run_files(zultra_hip_ctx_t *c, uint32_t nblocks) {
   hipStream_t st = c->lane_stream[0];
   {
      // one run — the captured graph — unless ZULTRA_HIP_STREAMS asks for more (down to four inputs per run). Measured, 1 M inputs of
      // 4 KiB in batches of 65 536: 621 k files/s as one run, 582 / 634 / 637 k as 2 / 3 / 4 runs: not worth leaving the graph for.
      // End of round 4: with a graph per run (below) a large batch is two runs — the second run's matchfinder fills the ~12 ms the first
      // spends in its last code build, literalisation and emission: 1 080 k -> 1 167 k files/s.
      const uint64_t want = c->auto_runs ? ((c->files_run_graphs && nblocks >= 8192u && c->nlanes >= 2) ? 2u : 1u) : zh_min64((uint64_t)c->nlanes, (uint64_t)nblocks / 4u);
      c->last_runs = (int)zh_max64(1, want);
   }
#ifndef ZH_EMU
   if (c->last_runs > 1 && c->files_run_graphs) {
      // Several runs, each a captured graph of its own on its own stream (the capture of ONE stream that forks into the runs' streams is what
      // crashes next to torch's runtime; a run's capture only forks to its chain stream, like the single graph's). What the runs share — clearing
      // the counters and the payload slots before, the copies of the results after — is launched around them. A run is TWO graphs, cut behind its
      // first matchfinder kernel: the event the next run's matchfinder waits for (the stagger of DESIGN.md 3.6) is recorded between them, by
      // the stream — without it the runs march in step, parse on parse, and two runs are no faster than one.
      const int runs = c->last_runs;
      zultra_hip_ctx_s::zh_run_graphs_t *G = NULL;
      for (int i = 0; i < 2; i++)
         if (c->rg[i].nblocks == nblocks && c->rg[i].data == c->cur_data && c->rg[i].runs == runs) G = &c->rg[i];
      if (!G) {
         G = c->rg[0].used <= c->rg[1].used ? &c->rg[0] : &c->rg[1];
         for (int k = 0; k < 2 * ZH_MAX_RUNS; k++) {
            if (G->exec[k]) (void)hipGraphExecDestroy(G->exec[k]);
            if (G->graph[k]) (void)hipGraphDestroy(G->graph[k]);
            G->exec[k] = NULL;
            G->graph[k] = NULL;
         }
         G->nblocks = 0;
         ZH_CHECK(c, hipStreamSynchronize(st));   // the input upload is not part of the graphs
         for (int k = 0; k < runs; k++) {
            hipStream_t sk = c->lane_stream[k];
            const uint32_t b0 = zh_files_run_lo(c, nblocks, k), b1 = zh_files_run_lo(c, nblocks, k + 1);
            ZH_CHECK(c, hipStreamSynchronize(sk));
            for (int part = 1; part <= 2; part++) {
               const int g = 2 * k + part - 1;
               ZH_CHECK(c, hipStreamBeginCapture(sk, hipStreamCaptureModeThreadLocal));
               const int rc = zh_enqueue_run(c, k, b0, b1 - b0, (uint64_t)(b1 - b0) * c->max_block, c->max_block, b0, b1 - b0, sk, c->side_stream[k], part);
               const hipError_t e = hipStreamEndCapture(sk, &G->graph[g]);
               if (rc != 0) return -1;
               ZH_CHECK(c, e);
               ZH_CHECK(c, hipGraphInstantiate(&G->exec[g], G->graph[g], NULL, NULL, 0));
            }
         }
         G->nblocks = nblocks;
         G->data = c->cur_data;
         G->runs = runs;
      }
      G->used = ++c->rg_tick;
      ZH_CHECK(c, hipEventRecord(c->lane_ev[0][1], st));
      ZH_CHECK(c, hipMemsetAsync(c->d_chunk_ctr, 0, ((size_t)nblocks * 2 + 3 * (size_t)runs) * sizeof(uint32_t), st));
      ZH_CHECK(c, hipMemsetAsync(c->d_ntasks, 0, ZH_NCNT * sizeof(uint32_t), st));
      ZH_CHECK(c, hipMemsetAsync(c->d_payload, 0, (size_t)nblocks * c->slot_stride, st));
      ZH_CHECK(c, hipEventRecord(c->ev2[1], st));
      for (int k = 0; k < runs; k++) {
         hipStream_t sk = c->lane_stream[k];
         if (k) {
            ZH_CHECK(c, hipStreamWaitEvent(sk, c->ev2[1], 0));
            if (c->stagger_ev) ZH_CHECK(c, hipStreamWaitEvent(sk, c->lane_ev[k - 1][2], 0));   // behind the previous run's first matchfinder kernel
         }
         ZH_CHECK(c, hipGraphLaunch(G->exec[2 * k], sk));
         ZH_CHECK(c, hipEventRecord(c->lane_ev[k][2], sk));
         ZH_CHECK(c, hipGraphLaunch(G->exec[2 * k + 1], sk));
         if (k) ZH_CHECK(c, hipEventRecord(c->lane_ev[k][17], sk));
      }
      for (int k = 1; k < runs; k++) ZH_CHECK(c, hipStreamWaitEvent(st, c->lane_ev[k][17], 0));
      if (zh_enqueue_files_tail(c, nblocks, st) != 0) return -1;
   }
   else if (c->last_runs > 1) {
      // Several runs: launched directly, ~35 launches per run and batch. (Forking the runs' streams inside a stream capture crashes in
      // hipStreamEndCapture when another HIP runtime — torch's — lives in the process; the launches of a batch are 0.5 ms of host time.)
      ZH_CHECK(c, hipEventRecord(c->lane_ev[0][1], st));
      if (zh_enqueue_files(c, nblocks, st) != 0) return -1;
   }
   else {
   if (!c->graph_exec || c->graph_nblocks != nblocks || c->graph_data != c->cur_data || c->graph_runs != c->last_runs) {
      if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
      if (c->graph) (void)hipGraphDestroy(c->graph);
      c->graph_exec = NULL;
      c->graph = NULL;
      ZH_CHECK(c, hipStreamSynchronize(st));   // the input upload is not part of the graph
      ZH_CHECK(c, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      const int rc = zh_enqueue_files(c, nblocks, st);
      const hipError_t e = hipStreamEndCapture(st, &c->graph);
      if (rc != 0) return -1;
      ZH_CHECK(c, e);
      ZH_CHECK(c, hipGraphInstantiate(&c->graph_exec, c->graph, NULL, NULL, 0));
      c->graph_nblocks = nblocks;
      c->graph_data = c->cur_data;
      c->graph_runs = c->last_runs;
   }
   ZH_CHECK(c, hipEventRecord(c->lane_ev[0][1], st));
   ZH_CHECK(c, hipGraphLaunch(c->graph_exec, st));
   }
#else
   ZH_CHECK(c, hipEventRecord(c->lane_ev[0][1], st));
   if (zh_enqueue_files(c, nblocks, st) != 0) return -1;
#endif
   ZH_CHECK(c, hipEventRecord(c->lane_ev[0][16], st));
   ZH_CHECK(c, hipStreamSynchronize(st));
   ZH_CHECK(c, hipGetLastError());
   c->results.assign(c->h_results, c->h_results + nblocks);   // (batch coordinates: zh_compact_results)
   memcpy(c->crc.data(), c->h_crc, nblocks * sizeof(uint32_t));
   c->adler.assign(c->h_adler, c->h_adler + 2 * (size_t)nblocks);
   c->nsubs = nblocks;
   c->files_stitch_rc = zh_stitch_verdict(c, false);
   c->files_stitched = 1;   // (no stitch_ms of its own: inside a captured graph an event is a node, not a time stamp; the batch's encode_ms covers the assembly)
   (void)hipEventElapsedTime(&c->timing.h2d_ms, c->lane_ev[0][0], c->ev_input);
   (void)hipEventElapsedTime(&c->timing.encode_ms, c->lane_ev[0][1], c->lane_ev[0][16]);   // the whole graph
   (void)hipEventElapsedTime(&c->timing.total_ms, c->lane_ev[0][0], c->lane_ev[0][16]);
   return (int)nblocks;
}

extern "C" int zultra_hip_compress_blocks(zultra_hip_ctx_t *c, const void *data, size_t data_size, int data_on_device,
                                          const zultra_hip_block_t *blocks, uint32_t nblocks) {
   ZH_EMU_SERIALIZE();
   if (!c || !data || !blocks || nblocks == 0 || nblocks > c->max_blocks) {
      if (c) snprintf(c->err, sizeof(c->err), "bad arguments");
      return -1;
   }
   for (uint32_t b = 0; b < nblocks; b++) {
      if (blocks[b].n == 0 || blocks[b].n > c->max_block || (c->files_mode && (blocks[b].n > c->max_file_size || blocks[b].prev != 0)) || blocks[b].prev > ZH_HISTORY || (uint64_t)blocks[b].prev + blocks[b].n > c->W ||
          blocks[b].win_off + blocks[b].prev + blocks[b].n > data_size) {
         snprintf(c->err, sizeof(c->err), "block %u out of range", b);
         return -1;
      }
   }
   if (data_on_device != 1 && data_size > c->data_cap) {
      snprintf(c->err, sizeof(c->err), "batch of %zu bytes exceeds the context's staging capacity %zu", data_size, c->data_cap);
      return -1;
   }
   ZH_CHECK(c, hipSetDevice(c->device));
   const bool stitch_now = c->ab_armed && !c->files_mode;   // (one batch only)
   c->ab_armed = 0;
   c->stitched_valid = 0;
   c->files_stitched = 0;
   c->nblocks = nblocks;
   c->nsubs = 0;
   c->blocks.assign((const zh_block_t *)blocks, (const zh_block_t *)blocks + nblocks);
   c->crc.resize(nblocks);
   c->payload_size = (size_t)nblocks * c->slot_stride;
   c->payload_on_host = 0;
   memset((void *)&c->timing, 0, sizeof(c->timing));

   // The batch is cut into `lanes` contiguous runs of max-blocks, each driven through the whole kernel sequence on its
   // own stream. Max-blocks are independent, every per-block buffer is addressed as base + block * stride, so a run
   // simply sees base pointers advanced to its first block. The single-wave, latency-bound kernels of one run
   // (zh_split, zh_sb_build, ...) then overlap with the wide kernels of the others.
   uint64_t batch_bytes = 0;
   for (uint32_t b = 0; b < nblocks; b++) batch_bytes += blocks[b].n;
   const uint64_t want_runs = c->auto_runs ? (batch_bytes >= (256ull << 20) ? 4u : 3u) : (uint64_t)c->nlanes;
   const int lanes = c->last_runs = (int)zh_max64(1, zh_min64(want_runs, zh_min64((uint64_t)nblocks / 4u, batch_bytes >> 22)));   // at least four max-blocks and 4 MiB per run
   // run k = blocks [run_lo(k), run_lo(k + 1)): the first run may be given a smaller share (c->first_run_pct of an equal share), so that the
   // other runs' matchfinders start earlier
   auto run_lo = [&](int k) -> uint32_t {
      if (k <= 0) return 0u;
      if (k >= lanes) return nblocks;
      if (c->run_share[0]) {   // explicit shares (set for `lanes` runs); a run is never empty
         uint64_t acc = 0, tot = 0;
         for (int j = 0; j < lanes; j++) tot += c->run_share[j] ? c->run_share[j] : 1u;
         for (int j = 0; j < k; j++) acc += c->run_share[j] ? c->run_share[j] : 1u;
         const uint64_t lo = (uint64_t)nblocks * acc / tot;
         return (uint32_t)zh_min64(zh_max64(lo, (uint64_t)k), (uint64_t)nblocks - (uint64_t)(lanes - k));
      }
      // (a run is never empty: with ZULTRA_HIP_FIRST_RUN / _LAST_RUN below 25 and four max-blocks per run the shares rounded to 0, and a
      // zero-sized grid fails the batch)
      const uint64_t first = zh_max64(1, (uint64_t)nblocks * c->first_run_pct / (100ull * (uint64_t)lanes));
      if (lanes < 3) return (uint32_t)(first + ((uint64_t)nblocks - first) * (uint64_t)(k - 1) / (uint64_t)(lanes - 1));
      const uint64_t last = zh_max64(1, (uint64_t)nblocks * c->last_run_pct / (100ull * (uint64_t)lanes));   // likewise the last run: its passes are the tail of the batch
      const uint64_t mid = (uint64_t)nblocks - first - last;
      if (k == lanes - 1) return (uint32_t)(nblocks - last);
      return (uint32_t)(first + mid * (uint64_t)(k - 1) / (uint64_t)(lanes - 2));
   };
   hipStream_t st0 = c->lane_stream[0];
   bool any_nochains = false;
   for (int k = 0; k < ZH_MAX_RUNS; k++) {
      c->run_nochains[k] = !c->files_mode && c->chain_skip && k < lanes && c->chain_seen_runs == lanes && c->chain_seen[k] == 0;
      any_nochains = any_nochains || c->run_nochains[k];
   }
   // (such a batch may have to be run again, see below — as may one with a run enqueued without its <true> overflow forms)
   (void)any_nochains;
   const bool stitch_with = stitch_now;   // (safe with void runs: zh_compact_results marks the batch, zh_stitch_scan and zh_stitch then write nothing)

   ZH_CHECK(c, hipEventRecord(c->lane_ev[0][0], st0));
   // data_on_device == 2: pageable host memory, and the batch runs as staggered runs of max-blocks — every run's bytes are staged
   // (host copy into the context's pinned buffer) and uploaded on the run's own stream just before its kernels are launched, so the
   // copies of run k+1 go on under the kernels of run k: only the first run's are waited for. (Files mode runs from a captured graph:
   // its input goes up in one piece like mode 0.)
   uint8_t *per_run_stage = NULL;
   if (data_on_device == 2 && !c->files_mode) {
      per_run_stage = (uint8_t *)zultra_hip_staging(c, 0, data_size);
      if (!per_run_stage) {
         snprintf(c->err, sizeof(c->err), "no pinned staging for %zu bytes", data_size);
         return -1;
      }
   }
   if (data_on_device == 1)
      c->cur_data = (const uint8_t *)data;
   else {
      if (!per_run_stage) ZH_CHECK(c, hipMemcpyAsync(c->d_data, data, data_size, hipMemcpyHostToDevice, st0));
      c->cur_data = c->d_data;
   }
   memcpy(c->h_blocks, blocks, nblocks * sizeof(zh_block_t));
   ZH_CHECK(c, hipMemcpyAsync(c->d_blocks, c->h_blocks, nblocks * sizeof(zh_block_t), hipMemcpyHostToDevice, st0));
   if (c->files_mode) {
      // one sub-block per input, ceil(size / ZH_TASK) tasks each: the task ranges of zh_plan_files, from the sizes alone
      uint32_t acc = 0;
      for (uint32_t b = 0; b < nblocks; b++) {
         c->h_task_prefix[b] = acc;
         acc += (blocks[b].n + ZH_TASK - 1) / ZH_TASK;
      }
      c->h_task_prefix[nblocks] = acc;
      ZH_CHECK(c, hipMemcpyAsync(c->d_task_prefix, c->h_task_prefix, ((size_t)nblocks + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, st0));
   }
   if (zh_build_segments(c, blocks, nblocks, st0) != 0) return -1;
   ZH_CHECK(c, hipEventRecord(c->ev_input, st0));
   if (c->files_mode) return zh_run_files(c, nblocks);

   // ---- every run, start to finish: nothing in it waits for the host (zh_enqueue_run) ---------------------------------------------
   uint32_t run_b0[ZH_MAX_RUNS];
   for (int k = 0; k < lanes; k++) {
      hipStream_t st = c->lane_stream[k], side = c->side_stream[k];
      hipEvent_t *ev = c->lane_ev[k];
      const uint32_t b0 = run_lo(k), b1 = run_lo(k + 1);
      const uint32_t nb = b1 - b0;
      run_b0[k] = c->last_run_b0[k] = b0;
      if (k) {
         ZH_CHECK(c, hipStreamWaitEvent(st, c->ev_input, 0));
         // stagger the runs by one stage: this run's wide matchfinder kernels start when the previous run reaches its
         // narrow ones (token chain, splitter), so narrow and wide kernels of different runs share the chip
         if (c->stagger_ev) ZH_CHECK(c, hipStreamWaitEvent(st, c->lane_ev[k - 1][c->stagger_ev], 0));
      }
      uint64_t total_n = 0;
      uint32_t max_n = 0;
      for (uint32_t b = b0; b < b1; b++) {
         total_n += blocks[b].n;
         max_n = max(max_n, blocks[b].n);
      }
      if (per_run_stage) {
         // this run's windows: from the first block's history to the last block's end (the 32 KiB in front of the run go up twice,
         // with the run before it: the same bytes)
         // (min / max over the run's blocks: the windows of a batch need not ascend, nor be contiguous — what lies between them goes up too)
         uint64_t lo = blocks[b0].win_off, hi = blocks[b0].win_off + blocks[b0].prev + blocks[b0].n;
         for (uint32_t b = b0 + 1; b < b1; b++) {
            lo = min(lo, (uint64_t)blocks[b].win_off);
            hi = max(hi, (uint64_t)blocks[b].win_off + blocks[b].prev + blocks[b].n);
         }
         const uint8_t *src = (const uint8_t *)data + lo;
         uint8_t *dst = per_run_stage + lo;
         const size_t len = (size_t)(hi - lo), piece = 8u << 20;
         if (len >= 2 * piece) {   // (a run of tens of MB: the host copy is split over a few threads)
            const size_t nt = len / piece < 4 ? len / piece : 4;
            std::vector<std::thread> th;
            size_t started = 1;   // (nothing thrown here may cross the C ABI: pieces without a thread are copied by this one)
            try {
               th.reserve(nt);
               for (; started < nt; started++) th.emplace_back([=] { memcpy(dst + len * started / nt, src + len * started / nt, len * (started + 1) / nt - len * started / nt); });
            } catch (...) {
            }
            memcpy(dst, src, len / nt);
            if (started < nt) memcpy(dst + len * started / nt, src + len * started / nt, len - len * started / nt);
            for (auto &t : th) t.join();
         }
         else
            memcpy(dst, src, len);
         ZH_CHECK(c, hipMemcpyAsync(c->d_data + lo, dst, len, hipMemcpyHostToDevice, st));
      }
      ZH_CHECK(c, hipEventRecord(ev[1], st));
      const uint32_t sg0 = c->seg_base[b0], nsg = c->seg_base[b1] - sg0;   // this run's matchfinder segments
      ZH_CHECK(c, hipMemsetAsync(c->d_chunk_ctr + (size_t)sg0 * 2 + 3 * (size_t)k, 0, ((size_t)nsg * 2 + 3) * sizeof(uint32_t), st));
      ZH_CHECK(c, hipMemsetAsync(c->d_ntasks + (size_t)k * ZH_CNT_STRIDE, 0, ZH_CNT_STRIDE * sizeof(uint32_t), st));   // the run's counters and tickets
      // per-max-block CRC-32 (linear part) and Adler-32 for the framing's footer need the input only: on the run's side stream, next to its matchfinder
      // (behind the run's last kernel they were the tail of the batch; behind its frontier, round 4, on the path to its first parse pass). The side
      // stream's later kernels — the chains of every pass — are joined by the run's stream: so is this one.
      ZH_CHECK(c, hipStreamWaitEvent(side, ev[1], 0));
      ZH_LAUNCH(zh_crc32_blocks, nb, ZH_CRC_THREADS, side, c->cur_data, (const zh_block_t *)(c->d_blocks + b0), (const uint32_t *)c->d_crc_tables, c->d_crc + b0, c->d_adler + 2 * (size_t)b0);
      // token bits are ORed into the payload slots: cleared there too, long before stage 3 needs them
      ZH_CHECK(c, hipMemsetAsync(c->d_payload + (uint64_t)b0 * c->slot_stride, 0, (size_t)nb * c->slot_stride, side));
      if (stitch_with && k == 0) {
         // ... and so is the stream buffer of a stitch that goes out with the batch (zh_enqueue_stitch): nobody reads it between two batches' stitches
         uint64_t bound = 16;
         for (uint32_t b = 0; b < nblocks; b++) bound += (uint64_t)blocks[b].n + 5ull * (blocks[b].n / 65535u + 1u);
         bound += 5ull * (uint64_t)nblocks * c->max_subs;
         ZH_CHECK(c, hipMemsetAsync(c->d_stream, 0, (size_t)zh_min64((uint64_t)c->stream_cap + 16, (bound + 3) & ~3ull), side));
      }
      if (zh_enqueue_run(c, k, b0, nb, total_n, max_n, sg0, nsg, st, side, 0) != 0) return -1;
      ZH_CHECK(c, hipMemcpyAsync(c->h_crc + b0, c->d_crc + b0, nb * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
      ZH_CHECK(c, hipMemcpyAsync(c->h_adler + 2 * (size_t)b0, c->d_adler + 2 * (size_t)b0, 2 * nb * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
      ZH_CHECK(c, hipEventRecord(ev[16], st));
   }
   // behind the last run: the descriptors in stream order and batch coordinates, the counts — the first thing the host waits for
   for (int k = 1; k < lanes; k++) ZH_CHECK(c, hipStreamWaitEvent(st0, c->lane_ev[k][16], 0));
   if (zh_enqueue_compact(c, run_b0, lanes, st0) != 0) return -1;
   // (the host's copy of the descriptors — the getters' — needs the count: a small batch copies what it can hold with the counts, one wait instead of two)
   const bool copy_bound = (uint64_t)nblocks * c->max_subs * sizeof(zh_subblock_t) <= (256u << 10);
   if (copy_bound) ZH_CHECK(c, hipMemcpyAsync(c->h_results, c->d_results_compact, (size_t)nblocks * c->max_subs * sizeof(zh_subblock_t), hipMemcpyDeviceToHost, st0));
   if (stitch_with) {
      // the stitch behind the batch's last kernel (zultra_hip_stitch_with_batch): scan, bit mover and the scan's report before the host's one wait
      ZH_CHECK(c, hipEventRecord(c->ev[0], st0));
      if (zh_enqueue_stitch(c, st0, c->ab_phase, c->ab_final, 0, false, false) != 0) return -1;
   }
   ZH_CHECK(c, hipStreamSynchronize(st0));
   ZH_CHECK(c, hipGetLastError());
   for (int k = 0; k < lanes; k++) {
      const uint32_t *cnt = c->h_ntasks + (size_t)k * ZH_CNT_STRIDE;
      c->chain_seen[k] = cnt[ZH_CNT_VLONG] + cnt[ZH_CNT_LONG] + cnt[ZH_CNT_SHORT] + cnt[ZH_CNT_SEGTASKS];
   }
   c->chain_seen_runs = lanes;
   for (int k = 0; k < lanes; k++) {
      const uint32_t *cnt = c->h_ntasks + (size_t)k * ZH_CNT_STRIDE;
      c->seen_nsubs[k] = cnt[ZH_CNT_NSUBS];
      c->seen_ntasks[k] = cnt[ZH_CNT_TASKS];
   }
   for (int k = 0; k < lanes; k++) {
      const bool outgrown = c->run_nomore[k] && (c->seen_nsubs[k] > c->run_grids[k][0] || c->seen_ntasks[k] > c->run_grids[k][1]);
      if ((c->run_nochains[k] && c->chain_seen[k] != 0) || outgrown) {
         // a run enqueued without chain kernels lists chains after all (zh_run_is_void: its kernels left at once): the batch again, every run with its chain
         // kernels (the counts just taken say so) — one batch's time, once, where the stream's content changes
         c->chain_redone++;
         c->ab_armed = stitch_now ? 1 : 0;
         return zultra_hip_compress_blocks(c, data, data_size, data_on_device, blocks, nblocks);
      }
   }
   const uint32_t nsubs = c->h_nsubs[0];
   if (nsubs < nblocks || (uint64_t)nsubs > (uint64_t)nblocks * c->max_subs) {
      snprintf(c->err, sizeof(c->err), "the device reports %u sub-blocks for %u max-blocks", nsubs, nblocks);
      return -1;
   }
   if (!copy_bound) ZH_CHECK(c, hipMemcpy(c->h_results, c->d_results_compact, (size_t)nsubs * sizeof(zh_subblock_t), hipMemcpyDeviceToHost));
   c->results.assign(c->h_results, c->h_results + nsubs);
   memcpy(c->crc.data(), c->h_crc, nblocks * sizeof(uint32_t));
   c->adler.assign(c->h_adler, c->h_adler + 2 * (size_t)nblocks);
   c->nsubs = nsubs;

   // device time per kernel group, summed over the runs (they overlap in wall time); total = first launch to last completion
   for (int k = 0; k < lanes; k++) {
      hipEvent_t *ev = c->lane_ev[k];
      float t = 0;
      auto add = [&](float &acc, hipEvent_t a, hipEvent_t b) {
         if (hipEventElapsedTime(&t, a, b) == hipSuccess) acc += t;
      };
      add(c->timing.group_ms, ev[1], ev[2]);
      add(c->timing.frontier_ms, ev[2], ev[3]);
      add(c->timing.tokenize_split_ms, ev[3], ev[4]);
      add(c->timing.init_ms, ev[4], ev[5]);
      for (int pass = 0; pass <= 3; pass++) {
         add(c->timing.parse_ms, ev[5 + 2 * pass], ev[6 + 2 * pass]);
         add(c->timing.build_ms, ev[6 + 2 * pass], ev[7 + 2 * pass]);
      }
      add(c->timing.post_ms, ev[13], ev[14]);
      add(c->timing.emit_ms, ev[14], ev[15]);
      add(c->timing.d2h_ms, ev[15], ev[16]);
      float tot = 0;
      if (hipEventElapsedTime(&tot, c->lane_ev[0][0], ev[16]) == hipSuccess && tot > c->timing.total_ms) c->timing.total_ms = tot;
   }
   (void)hipEventElapsedTime(&c->timing.h2d_ms, c->lane_ev[0][0], c->ev_input);
   (void)hipEventElapsedTime(&c->timing.head_ms, c->lane_ev[0][0], c->lane_ev[0][3]);
   {
      float last_mf = 0;
      if (hipEventElapsedTime(&last_mf, c->lane_ev[0][0], c->lane_ev[lanes - 1][3]) == hipSuccess) c->timing.tail_ms = c->timing.total_ms - last_mf;
   }
   c->timing.matchfinder_ms = c->timing.group_ms + c->timing.frontier_ms;
   c->timing.encode_ms = c->timing.init_ms + c->timing.parse_ms + c->timing.build_ms + c->timing.post_ms + c->timing.emit_ms;
   if (stitch_with) {
      (void)hipEventElapsedTime(&c->timing.stitch_ms, c->ev[0], c->ev[1]);
      c->stitched_rc = zh_stitch_verdict(c, false);
      c->stitched_phase = c->ab_phase;
      c->stitched_final = c->ab_final;
      c->stitched_valid = 1;
   }
   return (int)nsubs;
}

extern "C" const zultra_hip_subblock_t *zultra_hip_subblocks(const zultra_hip_ctx_t *c, uint32_t *count) {
   if (!c) return NULL;
   if (count) *count = c->nsubs;
   return (const zultra_hip_subblock_t *)c->results.data();
}
extern "C" const uint8_t *zultra_hip_payload(const zultra_hip_ctx_t *cc, size_t *size) {
   zultra_hip_ctx_t *c = (zultra_hip_ctx_t *)cc;
   if (!c) return NULL;
   if (!c->payload_on_host && c->nsubs) {   // copied on first request: the device stitcher does not need it on the host
      if (hipSetDevice(c->device) != hipSuccess) return NULL;
      if (hipMemcpy(c->h_payload, c->d_payload, c->payload_size, hipMemcpyDeviceToHost) != hipSuccess) return NULL;
      c->payload_on_host = 1;
   }
   if (size) *size = c->payload_size;
   return c->h_payload;
}

extern "C" int zultra_hip_block_adler32(const zultra_hip_ctx_t *c, uint32_t *out /* 2 per block: A, Bw */) {
   if (!c || !out) return -1;
   memcpy(out, c->adler.data(), 2 * (size_t)c->nblocks * sizeof(uint32_t));
   return (int)c->nblocks;
}

extern "C" int zultra_hip_block_crc32(const zultra_hip_ctx_t *c, uint32_t *out) {
   if (!c || !out) return -1;
   memcpy(out, c->crc.data(), c->nblocks * sizeof(uint32_t));
   return (int)c->nblocks;
}

// The stream assembly of the last batch on the device: the scan that decides where every sub-block goes (zh_stitch_scan), then the kernel that puts
// it there (zh_stitch). files: every max-block a stream of its own. One synchronisation, at the end, for the 80 bytes the scan reports.
// The stitch of the batch whose descriptors zh_compact_results has laid out (or will have, on `st`): clears what the bits are ORed into — unless the caller
// has (clear = false) —, the scan, the bit mover, the 80 bytes the scan reports. The grids need no count from the host: the scan is one workgroup, the mover strides
// over scan->nsubs. Enqueues only.
static int zh_enqueue_stitch(zultra_hip_ctx_t *c, hipStream_t st, uint32_t phase, int final_block, int files, bool scan_only, bool clear) {
   if (!scan_only && clear) {
      // the stream buffer must be zero where bits will be ORed in: everything the batch can fill — no sub-block takes more than its stored form,
      // size + 5 bytes per 65535 + the three header bits
      uint64_t bound = 16;
      for (uint32_t b = 0; b < c->nblocks; b++) bound += (uint64_t)c->blocks[b].n + 5ull * (c->blocks[b].n / 65535u + 1u);
      bound += 5ull * (uint64_t)c->nblocks * c->max_subs;
      const size_t clear_bytes = (size_t)zh_min64((uint64_t)c->stream_cap + 16, (bound + 3) & ~3ull);
      ZH_CHECK(c, hipMemsetAsync(c->d_stream, 0, clear_bytes, st));
   }
   // (the scan's chunk tables are 32-bit: a chunk of ceil(nblocks / 1024) max-blocks must stay below 2^32 bits — only HBM capacity has ruled that out so far)
   if (((uint64_t)c->nblocks + ZH_SCAN_THREADS - 1) / ZH_SCAN_THREADS * zh_stitch_blockbuf_cap(c->max_block < ZH_MIN_BLOCK ? (uint32_t)ZH_MIN_BLOCK : c->max_block) * 8ull >= (1ull << 32)) {
      snprintf(c->err, sizeof(c->err), "batch too large for the stream assembly's 32-bit chunk tables (%u max-blocks of %u bytes)", c->nblocks, c->max_block);
      return -1;
   }
   ZH_LAUNCH(zh_stitch_scan, 1, ZH_SCAN_THREADS, st, (const zh_subblock_t *)c->d_results_compact, (const uint32_t *)c->d_nsubs, c->nblocks, phase,
             files ? (c->max_block < ZH_MIN_BLOCK ? (uint32_t)ZH_MIN_BLOCK : c->max_block) : c->max_block, final_block, files, c->d_blk_start, c->d_items, c->d_file_off, c->d_scan_out);
   if (!scan_only) {
      const uint32_t grid = (uint32_t)zh_min64((uint64_t)c->nblocks * c->max_subs, zh_max64(4ull * c->nblocks, 1024));
      ZH_LAUNCH(zh_stitch, grid, ZH_STITCH_THREADS, st, (const zh_subblock_t *)c->d_results_compact, (const zh_stitch_item_t *)c->d_items, (const zh_block_t *)c->d_blocks, c->cur_data,
                (const uint8_t *)c->d_payload, c->d_stream, (const zh_scan_out_t *)c->d_scan_out, (uint64_t)c->stream_cap);
   }
   ZH_CHECK(c, hipEventRecord(c->ev[1], st));
   ZH_CHECK(c, hipMemcpyAsync(c->h_scan_out, c->d_scan_out, sizeof(zh_scan_out_t), hipMemcpyDeviceToHost, st));
   return 0;
}

// ... and what the host makes of the scan's report, once the stream has been waited for
static int zh_stitch_verdict(zultra_hip_ctx_t *c, bool scan_only) {
   if (c->h_scan_out->nsubs != c->nsubs) {
      snprintf(c->err, sizeof(c->err), "stream assembly saw %u sub-blocks, the batch has %u", c->h_scan_out->nsubs, c->nsubs);
      return -1;
   }
   if (scan_only) return 0;
   if (c->h_scan_out->failed) {
      snprintf(c->err, sizeof(c->err), "stream assembly overflows the per-block buffer bound (ZULTRA_ERROR_DST)");
      return -2;
   }
   if (((c->h_scan_out->end_bit + 7) >> 3) + 8 > c->stream_cap) {
      snprintf(c->err, sizeof(c->err), "stream buffer too small");
      return -1;
   }
   return 0;
}

static int zh_stitch_on_device(zultra_hip_ctx_t *c, uint32_t phase, int final_block, int files, bool scan_only = false) {
   ZH_CHECK(c, hipSetDevice(c->device));
   hipStream_t st = c->stream;
   c->stitched_valid = 0;   // (the items and the scan's report are rewritten)
   c->files_stitched = 0;
   ZH_CHECK(c, hipEventRecord(c->ev[0], st));
   if (zh_enqueue_stitch(c, st, phase, final_block, files, scan_only, true) != 0) return -1;
   ZH_CHECK(c, hipStreamSynchronize(st));
   ZH_CHECK(c, hipGetLastError());
   if (!scan_only) (void)hipEventElapsedTime(&c->timing.stitch_ms, c->ev[0], c->ev[1]);
   return zh_stitch_verdict(c, scan_only);
}

// The next batch of max-blocks (zultra_hip_compress_blocks) is stitched at bit phase `phase` (final_block as for zultra_hip_stitch_device) BEHIND ITS LAST KERNEL,
// before the host is woken: a caller who knows the phase its batch starts at — the first batch of a stream, any batch of a stream compressed batch by batch —
// saves the second synchronisation and what its thread does between the two calls (0.8 ms of a 38 ms step, profiles/r05_timeline_c2.txt). The
// zultra_hip_stitch_device call that follows, with the same phase and final_block, returns that stitch's result without launching anything; any other call
// stitches again as before. One batch only: the setting is consumed by the batch. Not for files contexts.
extern "C" int zultra_hip_stitch_with_batch(zultra_hip_ctx_t *c, int enable, uint32_t phase, int final_block) {
   if (!c || c->files_mode) return -1;
   c->ab_armed = enable ? 1 : 0;
   c->ab_phase = phase & 7u;
   c->ab_final = final_block;
   return 0;
}

// Device stitch of the last batch. state->nacc = pending bits (phase) before the batch; on return state->nacc = pending
// bits after it and *end_bit = total bits from the start of the byte that held the pending bits. The stream buffer
// holds ceil(end_bit / 8) bytes; its first byte carries only this batch's bits (OR the caller's pending bits in).
extern "C" int zultra_hip_stitch_device(zultra_hip_ctx_t *c, zultra_hip_bitstate_t *state, int final_block, uint64_t *end_bit) {
   ZH_EMU_SERIALIZE();
   if (!c || !state || !end_bit || c->nsubs == 0) return -1;
   // (a stitch that went out with the batch, zultra_hip_stitch_with_batch: its result is at hand)
   const int rc = (c->stitched_valid && c->stitched_phase == (state->nacc & 7u) && c->stitched_final == final_block) ? c->stitched_rc : zh_stitch_on_device(c, state->nacc & 7u, final_block, 0);
   if (rc != 0) return rc;
   const uint64_t eb = c->h_scan_out->end_bit;
   state->nacc = (uint32_t)(eb & 7);
   *end_bit = eb;
   return 0;
}

// Where the last batch would end for each of the eight bit phases it could start at: end_bits[p] = its end bit had it started at phase p (origin: the
// byte holding the p pending bits), failed bit p = that start overflows a max-block buffer (the reference fails with ZULTRA_ERROR_DST there). This is the
// table a rank hands its neighbours when one stream is cut over several devices (zultra_amd/sharded.py): the bit length of a shard depends on the phase
// it starts at, and the scan (zh_stitch_scan) computes all eight on the way. Nothing is written to the stream buffer.
extern "C" int zultra_hip_stitch_phase_table(zultra_hip_ctx_t *c, uint64_t *end_bits /* 8 */, uint32_t *failed_mask) {
   ZH_EMU_SERIALIZE();
   if (!c || !end_bits || c->nsubs == 0 || c->files_mode) return -1;
   const int rc = zh_stitch_on_device(c, 0, -1, 0, true);
   if (rc != 0) return rc;
   for (int p = 0; p < 8; p++) end_bits[p] = c->h_scan_out->table_end[p];
   if (failed_mask) *failed_mask = c->h_scan_out->table_failed;
   return 0;
}

// Files mode assembly: every max-block of the last batch becomes its own raw deflate stream (BFINAL on its last
// sub-block, padded to a byte, libzultra.c:414-417), laid end to end in the stream buffer; file_off[b]..file_off[b+1]
// are its bytes. Works for any context, it is what compressing each max-block with its own zultra_memory_compress gives.
extern "C" int zultra_hip_stitch_files(zultra_hip_ctx_t *c, uint64_t *file_off /* nblocks + 1 */) {
   ZH_EMU_SERIALIZE();
   if (!c || !file_off || c->nsubs == 0) return -1;
   if (!c->d_file_off) {   // (a context of max-blocks: allocated on first use)
      ZH_CHECK(c, hipSetDevice(c->device));
      if (zh_alloc(c, &c->d_file_off, (size_t)c->max_blocks + 1)) return -1;
   }
   if (c->files_mode && c->files_stitched) {   // (the batch brought its assembly with it, zh_enqueue_files_tail)
      if (c->files_stitch_rc != 0) return c->files_stitch_rc;
      memcpy(file_off, c->h_file_off, ((size_t)c->nblocks + 1) * sizeof(uint64_t));
      return 0;
   }
   const int rc = zh_stitch_on_device(c, 0, -1, 1);
   if (rc != 0) return rc;
   ZH_CHECK(c, hipMemcpy(file_off, c->d_file_off, ((size_t)c->nblocks + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost));
   return 0;
}

extern "C" int zultra_hip_compress_files(zultra_hip_ctx_t *c, const void *data, size_t data_size, int data_on_device, const uint64_t *offsets,
                                         const uint32_t *sizes, uint32_t nfiles, uint64_t *file_off) {
   if (!c || !offsets || !sizes || !file_off || nfiles == 0) return -1;
   std::vector<zultra_hip_block_t> blk(nfiles);
   for (uint32_t i = 0; i < nfiles; i++) {
      blk[i].win_off = offsets[i];
      blk[i].prev = 0;
      blk[i].n = sizes[i];
   }
   const int rc = zultra_hip_compress_blocks(c, data, data_size, data_on_device, blk.data(), nfiles);
   if (rc <= 0) return rc;
   return zultra_hip_stitch_files(c, file_off) == 0 ? rc : -1;
}

extern "C" const void *zultra_hip_stream_device(const zultra_hip_ctx_t *c) { return c ? c->d_stream : NULL; }

extern "C" int zultra_hip_stream_read(zultra_hip_ctx_t *c, void *out, size_t offset, size_t nbytes) {
   if (!c || offset + nbytes > c->stream_cap) return -1;
   ZH_CHECK(c, hipSetDevice(c->device));
   ZH_CHECK(c, hipMemcpy(out, (const uint8_t *)c->d_stream + offset, nbytes, hipMemcpyDeviceToHost));
   return 0;
}
extern "C" void zultra_hip_last_timing(const zultra_hip_ctx_t *c, zultra_hip_timing_t *t) {
   if (c && t) *t = c->timing;
}

// diagnostics: the chain kernels' per-ticket records of the last batch (ZULTRA_HIP_CHAIN_TRACE=1), [run 0..3][pass 0..3][ZH_TRACE_SLOTS][3]
extern "C" int zultra_hip_chain_trace(zultra_hip_ctx_t *c, uint64_t *out, uint32_t *slots) {
   if (!c || !c->d_chain_trace) return -1;
   if (slots) *slots = ZH_TRACE_SLOTS;
   if (out) ZH_CHECK(c, hipMemcpy(out, c->d_chain_trace, (size_t)3 * ZH_TRACE_SLOTS * 16 * sizeof(uint64_t), hipMemcpyDeviceToHost));
   return 0;
}

// diagnostics: the cut tasks of run `run` of the last batch, {task, K | S / 32 << 12, first vector slot, segment completions | failed cuts << 16 (over the four passes)}
extern "C" int zultra_hip_cut_tasks(zultra_hip_ctx_t *c, uint32_t run, uint32_t *out, uint32_t cap) {
   if (!c || (int)run >= c->last_runs || c->files_mode) return -1;
   const uint32_t n = min(cap, c->h_ntasks[(size_t)run * ZH_CNT_STRIDE + ZH_CNT_SEGTASKS]);
   if (out && n) ZH_CHECK(c, hipMemcpy(out, c->d_segtasks + (uint64_t)c->last_run_b0[run] * c->seg_tasks_per_block, (size_t)n * sizeof(uint4), hipMemcpyDeviceToHost));
   return (int)n;
}

#ifdef ZH_MF_PROFILE
extern "C" int zultra_hip_mf_profile(unsigned long long *out, int reset) {
   if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(zh_mf_prof), sizeof(zh_mf_prof)) != hipSuccess) return -1;
   if (reset) {
      unsigned long long z[16] = {0};
      if (hipMemcpyToSymbol(HIP_SYMBOL(zh_mf_prof), z, sizeof(z)) != hipSuccess) return -1;
   }
   return 0;
}
#endif

#ifdef ZH_MFG_PROFILE
extern "C" int zultra_hip_mfg_profile(unsigned long long *out, int reset) {
   if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(zh_mfg_prof), sizeof(zh_mfg_prof)) != hipSuccess) return -1;
   if (reset) {
      unsigned long long z[32] = {0};
      if (hipMemcpyToSymbol(HIP_SYMBOL(zh_mfg_prof), z, sizeof(z)) != hipSuccess) return -1;
   }
   return 0;
}
#endif

#ifdef ZH_LP_PROFILE
extern "C" int zultra_hip_lp_profile(unsigned long long *out, int reset) {
   if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(zh_lp_prof), sizeof(zh_lp_prof)) != hipSuccess) return -1;
   if (reset) {
      unsigned long long z[16] = {0};
      if (hipMemcpyToSymbol(HIP_SYMBOL(zh_lp_prof), z, sizeof(z)) != hipSuccess) return -1;
   }
   return 0;
}
#endif

#ifdef ZH_LP_TRACE
extern "C" int zultra_hip_lp_trace(unsigned long long *out /* 4 passes x ZH_LP_TRACE_SLOTS x 4 */, unsigned int *slots) {
   if (slots) *slots = ZH_LP_TRACE_SLOTS;
   if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(zh_lp_trace), sizeof(zh_lp_trace)) != hipSuccess) return -1;
   static unsigned long long z[4u * ZH_LP_TRACE_SLOTS * 4u];
   if (out && hipMemcpyToSymbol(HIP_SYMBOL(zh_lp_trace), z, sizeof(z)) != hipSuccess) return -1;
   return 0;
}
#endif

extern "C" void zultra_hip_last_stats(const zultra_hip_ctx_t *c, zultra_hip_stats_t *out) {
   if (!c || !out) return;
   memset(out, 0, sizeof(*out));
   out->blocks = c->nblocks;
   out->subblocks = c->nsubs;
   out->batches_rerun = c->chain_redone;
   for (int k = 0; k < c->last_runs && k < ZH_MAX_RUNS; k++) {   // (a run's counters are cleared when it is launched: only the last batch's runs count)
      const uint32_t *cnt = c->h_ntasks + (size_t)k * ZH_CNT_STRIDE;
      out->tasks += cnt[ZH_CNT_TASKS];
      out->huge_tasks += cnt[ZH_CNT_VLONG] + cnt[ZH_CNT_LONG] + cnt[ZH_CNT_SHORT] + cnt[ZH_CNT_SEGTASKS];
      out->huge_positions += cnt[ZH_CNT_HUGE_POS];
      out->cut_tasks += cnt[ZH_CNT_SEGTASKS];
      out->cut_segments += cnt[ZH_CNT_SEGITEMS];
      out->cut_redone += cnt[ZH_CNT_SEG_FAILED];   // over the four passes
      out->cut_demoted += cnt[ZH_CNT_DEMOTED];
      out->runs_without_chain_kernels += cnt[ZH_CNT_NOCHAINS] ? 1u : 0u;
      out->settled_passes += cnt[ZH_CNT_SETTLED];
      out->settled_kib += cnt[ZH_CNT_SETTLED_POS];
   }
   for (uint32_t b = 0; b < c->nblocks; b++) out->positions += c->blocks[b].n;
   out->runs = (uint32_t)c->last_runs;
}

extern "C" int zultra_hip_get_matches(zultra_hip_ctx_t *c, uint32_t block, uint16_t *out) {
   if (!c || block >= c->nblocks) return -1;
   ZH_CHECK(c, hipSetDevice(c->device));
   // rows live in two planes of four slots (zh_common.h); the second one is only written behind a full first one
   const size_t n = c->blocks[block].n;
   std::vector<uint32_t> lo(n * 4), hi(n * 4);
   const uint32_t *base = (const uint32_t *)(c->d_match + (uint64_t)block * c->match_stride);
   ZH_CHECK(c, hipMemcpy(lo.data(), base, n * 16, hipMemcpyDeviceToHost));
   ZH_CHECK(c, hipMemcpy(hi.data(), base + 4 * ZH_ROW_HI_OFF(c->match_stride), n * 16, hipMemcpyDeviceToHost));
   uint32_t *o = (uint32_t *)out;
   for (size_t i = 0; i < n; i++) {
      const bool more = (lo[4 * i + 3] & 0xffffu) >= ZH_MIN_MATCH;
      for (int k = 0; k < 4; k++) {
         o[8 * i + k] = lo[4 * i + k];
         o[8 * i + 4 + k] = more ? hi[4 * i + k] : 0u;
      }
   }
   return 0;
}

extern "C" int zultra_hip_get_splits(zultra_hip_ctx_t *c, uint32_t block, int *out) {
   if (!c || block >= c->nblocks) return -1;
   ZH_CHECK(c, hipSetDevice(c->device));
   uint32_t st[ZH_MAX_SPLITS + 1], nt = 0, cnt = 0;
   ZH_CHECK(c, hipMemcpy(&cnt, c->d_split_cnt + block, sizeof(cnt), hipMemcpyDeviceToHost));   // (the host never needs the splitter's counts: fetched for this getter)
   if (cnt > ZH_MAX_SPLITS) return -1;
   ZH_CHECK(c, hipMemcpy(st, c->d_split_tok + (uint64_t)block * (ZH_MAX_SPLITS + 1), sizeof(st), hipMemcpyDeviceToHost));
   ZH_CHECK(c, hipMemcpy(&nt, c->d_ntok + block, sizeof(nt), hipMemcpyDeviceToHost));
   for (uint32_t k = 0; k < cnt; k++) {
      uint32_t t1 = st[k + 1], pos = c->blocks[block].prev + c->blocks[block].n;
      if (t1 < nt) ZH_CHECK(c, hipMemcpy(&pos, c->d_tok_pos + (uint64_t)block * c->tok_stride + t1, sizeof(pos), hipMemcpyDeviceToHost));
      out[k] = (int)pos;
   }
   return (int)cnt;
}

extern "C" int zultra_hip_get_parse(zultra_hip_ctx_t *c, uint32_t block, uint16_t *out) {
   if (!c || block >= c->nblocks) return -1;
   ZH_CHECK(c, hipSetDevice(c->device));
   ZH_CHECK(c, hipMemcpy(out, c->d_best + (uint64_t)block * c->best_stride, (size_t)c->blocks[block].n * sizeof(uint32_t),
                         hipMemcpyDeviceToHost));
   return 0;
}
