// zh_parse.h — stage 3a of the hot path: the backward optimal parse (reference src/blockdeflate.c:254-323,
// zultra_optimize_matches_lwd) as a GPU-wide data-parallel kernel, plus the histogram of the chosen parse
// (blockdeflate.c:371-400).
//
// What makes the parse parallel. The recurrence cost[i] = min(literal + cost[i+1], min_k len(k) + dist + cost[i+k])
// is serial in i, but it only ever reads cost[j] for j <= i + (longest match starting at i). Call position q a
// *barrier* when every match that starts before q ends at or before q. Then (1) no position left of q reads a cost
// right of q, so the choices left of q depend on cost[q] only through an additive constant — restarting the recurrence
// with cost[q] = 0 reproduces the reference's choices bit for bit (all comparisons are between sums that share the
// constant); and (2) every parse, greedy or optimal, has a token boundary at q (a token starting before q cannot
// jump over it). zh_barriers (zh_split.h) computes the barrier bitmap of each max-block with a running prefix-max of
// (position + longest length). On text a barrier falls every ~35 positions; on highly repetitive data they are rare
// and the parse degrades gracefully to one serial chain per run.
//
// Work decomposition (no serial dependency anywhere between the units):
//   task  = the positions of one sub-block between the barriers nearest to multiples of ZH_TASK;
//   piece = a barrier-to-barrier run of >= ZH_PIECE positions inside a task.
// Who parses what (since round 3): zh_parse_lanes (zh_parse_lanes.h) takes the task list, a quad of lanes per piece; zh_parse_chain (zh_parse_chain.h) the tasks
// with a barrier-free run too long for that, whole; and THIS file's row machinery — round 2's kernel: each 16-lane DPP row of a wave owns one piece, a wave advances
// four recurrences per step — the speculative SEGMENTS such tasks are cut into when a run has many (below), in the first workgroups of zh_parse_lanes' launch.
// Every forward walk over the chosen parse (histogram, literalisation, bit counting, emission) is a walk over tasks
// too, because a task starts on a token boundary.
//
// One recurrence step for a row (position p, lane s of the row prices length k = 3+s):
//   candidates collapse to one per length: with the slots ordered longest first, the slots able to provide length k
//   are a prefix, and among them only the cheapest distance can win; the reference's evaluation order (slot ascending,
//   length descending, strict improvement) makes the winner the minimum of (cost, slot, -k). Per position a 16-byte
//   record, built with all 64 lanes in parallel when a tile is staged, holds the bitmap of slot lengths and the
//   running minima of (distance price, slot) per slot; lane s finds "how many slots reach k" with one shift+popcount
//   and picks its byte. One row-wise DPP min-reduction yields the row's best match, which is compared with the literal.
//   Lengths 19..39 and slots stored with length >= 40 (tried at full length only, blockdeflate.c:286-297) take a
//   second, rarely executed section.
#pragma once
#include <zh_platform.h>
#include "zh_common.h"
#include "zh_split.h"

#ifndef ZH_TASK
#define ZH_TASK 2048        // target positions per task
#endif
#ifndef ZH_PIECE
#define ZH_PIECE 128        // target positions per piece
#endif
#define ZH_MAXPIECES 64
#ifndef ZH_COOP_MIN
#define ZH_COOP_MIN 1536     // a task with a piece longer than this goes to zh_parse_chain (zh_parse_chain.h), whole or cut into segments
#endif
#define ZH_NSYM (ZH_NLIT + ZH_NDIST)
// Costs are kept modulo 2^16: within the 258 positions a step can look ahead, two costs differ by less than 258 x 15 bits
// (either can be reached from the other by literals), so candidate costs relative to cost[p+1], biased by 2^14, stay
// inside 15 bits and compare exactly like the reference's 32-bit sums.
#define ZH_KEY_BIAS (1u << 14)
// The cost ring of a row: a step reads cost[p+3 .. p+258] and writes cost[p], so 259 entries are live; 320 (not 512) keeps
// the kernel's LDS at 6.7 KB per wave, i.e. 24 instead of 19 resident waves per CU (measured: the kernel's time is
// 8.6 ms + 136 ms / waves per CU). Indices are kept reduced: x < 2 * ZH_RING wraps with one subtract and one unsigned min.
#define ZH_RING 320u
__device__ __forceinline__ uint32_t zh_ring_wrap(uint32_t x) { return min(x, x - ZH_RING); }

// ---- speculative segments -------------------------------------------------------------------------------------------------
// A barrier-free run of tens of thousands of positions is one recurrence: 64 Ki positions are milliseconds of a single wave,
// four times per batch. But the recurrence forgets: started W positions to the right of a cut b with made-up costs (here: as if
// the sub-block ended at b + W), the DIFFERENCES cost[b + i] - cost[b], i = 0..258, usually come out exactly as the true ones —
// on text with long repeats the optimal paths of neighbouring positions funnel through common points within a few hundred
// positions (measured on the bench's corpora, W = 1024: 97 % of the cuts on Python sources, all of them on JSON-like records; the
// exception is data where nearly every position offers a 258-byte match: the costs of positions 258 apart copy each other and
// nothing is ever forgotten). And the choices left of b depend on the costs at b .. b+258 through their differences only (every
// comparison is between sums that share the constant — the argument that makes barriers restart points, see the top of this
// file). So a long task [t0, t1) whose longest matches are mostly shorter than 258 is cut at b_k = t1 - (K-1-k) S into K
// segments, and the segments are PIECES like any other: four of them share a wave (zh_parse_one_task<true>: the segment workgroups of zh_parse_lanes' launch), one per row. Segment k
// starts at b_k + ZH_CUT_WARM, stores no parse entries at or above b_k, records the relative costs of [b_k, b_k + 258] as it
// passes (speculated) and those of [b_{k-1}, b_{k-1} + 258] when it is done (its own left end). The wave that finishes a task last (below; zh_chain_check_task when zh_parse_chain takes the segments)
// (zh_parse_chain.h) then walks each cut task from the right: the last segment is exact by construction; segment k is exact if
// segment k+1 is and its speculated vector equals segment k+1's left one; otherwise it is parsed again from that (exact)
// vector, as a chain. The output is the reference's parse bit for bit either way; speculation only decides how much of it was
// computed in parallel.
#ifndef ZH_CUT_LEN
#define ZH_CUT_LEN 4096u          // positions per segment, about (zh_list_huge picks the number of segments, then their length)
#endif
#ifndef ZH_CUT_WARM
#define ZH_CUT_WARM 1024u         // warm-up positions right of a cut (a multiple of 32, 288 .. ZH_CUT_LEN)
#endif
#define ZH_CUT_MIN ZH_CUT_LEN      // tasks shorter than this stay whole (measured on JSON-like records, 50 MB: 8192 -> 45.4 ms, 4096 -> 43.6 ms, 2048 -> 43.3 ms)
#define ZH_CUT_ROWS 4u            // segments per wave (one per 16-lane row)
// segtasks[].y: number of segments | their length / 32 << 12
#define ZH_CUT_PACK(K, S) ((K) | (((S) >> 5) << 12))
#define ZH_CUT_K(y) ((y) & 0xfffu)
#define ZH_CUT_S(y) ((((y) >> 12) & 0x7ffffu) << 5)
#define ZH_CUT_DEMOTED 0x80000000u   // in segtasks[].y: the task is no longer parsed as segments (its entry in the fourth chain list says by whom)
#define ZH_VEC 264u               // int16 entries per cost vector: cost[x + i] - cost[x], i = 0..258 (+ padding)
#define ZH_VEC_LIVE 259u
#define ZH_VEC_BIAS 4096          // imported costs are (bias + difference) << 9: differences are below 258 x 15 in magnitude

// segtasks[] entry of a cut task (zh_list_huge): x = task, y = number of segments K, z = first vector slot (slot of segment k =
// z + k; a slot holds two vectors: speculated at the segment's right end, computed at its left end); segwaves[] entry: x = index
// into segtasks, y = first segment of the wave.
// Counters of a run (device: uint32 per field; one block of ZH_CNT_STRIDE words per run)
enum {
   ZH_CNT_TASKS = 0, ZH_CNT_VLONG, ZH_CNT_LONG, ZH_CNT_SHORT, ZH_CNT_HUGE_POS, ZH_CNT_SEGTASKS, ZH_CNT_SEGITEMS, ZH_CNT_SEG_FAILED,
   ZH_CNT_CHAIN_TICKET = 8, ZH_CNT_TASK_TICKET = 12, ZH_CNT_FIX_TICKET = 16, ZH_CNT_SEGWAVES = 20,
   ZH_CNT_SETTLED = 21 /* parse passes not run because the sub-block's prices had stopped moving (zh_sb_build_one) */, ZH_CNT_SETTLED_POS = 22 /* ... in KiB of input */,
   ZH_CNT_DEMOTED = 23 /* cut tasks handed to zh_parse_chain as whole chains for the passes left (zh_parse_one_task) */, ZH_CNT_DEMOTED_PASS = 24 /* .. 27: listed in pass p */,
   ZH_CNT_NSUBS = 28 /* sub-blocks of the run (zh_plan_subblocks: the host never sees the splitter's counts) */,
   ZH_CNT_NOCHAINS = 29 /* set by the HOST when it enqueues the run: no chain kernels were launched for it (zh_run_is_void) */,
   ZH_CNT_SBGRID = 30, ZH_CNT_TASKGRID = 31 /* set by the HOST (non-zero) when it launches no <true> overflow forms for the run: the grids of its <false> forms */, ZH_CNT_STRIDE = 32
};
static_assert(ZH_CNT_TASKS == 0, "zh_post_tasks / zh_emit_tasks take the run's counter block as the pointer to its task count");

// A stream without chains must not pay for them (round 6). Whether a run lists anything for zh_parse_chain is known on the device only, and an empty chain grid still
// costs its pass 0.4-2 ms: its four-wave workgroups (169 registers, 24 KB of LDS) have to be scheduled among the quad kernels' waves, which hold every LDS granule of
// a CU, before they can find the lists empty and leave. So the host goes by what the context's LAST batch listed, run by run: where that was nothing it launches no
// chain kernel at all and says so in cnt[ZH_CNT_NOCHAINS]. If the run lists chains after all, its parse is incomplete: every kernel behind zh_list_huge then leaves at
// once (nothing walks a parse that was never written), the host sees both facts in the counters it reads back anyway, and runs the batch again with the chain
// kernels — the price of one batch, once, where a stream's content changes.
// The same bargain for the <true> overflow forms of the per-sub-block / per-task kernels (zh_device.hip, ZH_LAUNCH_BOTH): nearly always they find nothing beyond the <false>
// grid, yet each is a launch whose workgroups must find room on a full chip before they can leave — zh_sb_init<true> (99 registers, 8 KB of LDS) 0.4 ms on average and up to
// 1.7 ms, on every run's path to its first parse pass (profiles/r06_timeline_c2.txt). Where the context's last batch stayed inside this batch's grids the host does not launch
// them and leaves the grids in the counters; a run that outgrows them is void like one that lists chains without chain kernels.
__device__ __forceinline__ bool zh_run_is_void(const uint32_t *cnt) {
   const bool chains = cnt[ZH_CNT_NOCHAINS] != 0 && (cnt[ZH_CNT_VLONG] | cnt[ZH_CNT_LONG] | cnt[ZH_CNT_SHORT] | cnt[ZH_CNT_SEGTASKS]) != 0u;
   const bool outgrown = cnt[ZH_CNT_SBGRID] != 0 && (cnt[ZH_CNT_NSUBS] > cnt[ZH_CNT_SBGRID] || cnt[ZH_CNT_TASKS] > cnt[ZH_CNT_TASKGRID]);
   return chains || outgrown;
}

// sub-block work item produced by zh_plan_subblocks
struct zh_work_t {
   uint32_t block, start, size;   // start = absolute window offset
   uint32_t tok0, tok1;           // greedy token range
   uint32_t out_cap;              // slot capacity in bytes (multiple of 4)
   uint64_t out_off;              // byte offset of the bit slot inside the batch payload (multiple of 4)
   uint32_t index;                // position of this sub-block in stream order (= index of this item)
   uint32_t task_base, ntasks;    // its tasks are [task_base, task_base + ntasks) in the batch's task list
   uint32_t pad;
};

// per-sub-block coder state, lives in HBM between the kernels of the pipeline
struct zh_sbstate_t {
   uint8_t lit_len[ZH_NLIT], dist_len[ZH_NDIST];          // code lengths in force (0 = unused symbol)
   uint16_t lit_code[ZH_NLIT], dist_code[ZH_NDIST];
   uint8_t pre_lit_len[ZH_NLIT], pre_dist_len[ZH_NDIST];  // lengths after the last parse pass, before the RLE-friendly
                                                           // alternative: the prices literalisation uses (:923 vs :926-945)
   uint32_t is_dynamic, failed, hdr_bits;
   int32_t static_cost, dynamic_cost;
   uint32_t settled;        // the prices the next parse pass would use are those of the last one: its parse, and every later one, would come out the same
                            // (zh_sb_build_one); the parse kernels leave the sub-block's parse entries and task histograms as they are
   uint32_t pad[2];
};

// boundary j of a sub-block's task list (window positions): 0 -> start, ntasks -> end
__device__ inline uint32_t zh_task_boundary(const uint64_t *bar, uint32_t prev, uint32_t start, uint32_t end, uint32_t j, uint32_t ntasks) {
   if (j == 0) return start;
   if (j >= ntasks) return end;
   return prev + zh_first_barrier(bar, start + j * ZH_TASK - prev, end - prev);
}

// the pieces of task [t0, t1): boundaries bnd[0..np] (LDS, np <= ZH_MAXPIECES), every inner one a barrier
__device__ inline uint32_t zh_task_pieces(uint32_t *bnd, const uint64_t *bar, uint32_t prev, uint32_t t0, uint32_t t1, uint32_t lane) {
   uint32_t np = 0;
   if (t1 > t0) {
      const uint32_t len = t1 - t0;
      const uint32_t ps = max((uint32_t)ZH_PIECE, (len + ZH_MAXPIECES - 1) / ZH_MAXPIECES);
      np = (len + ps - 1) / ps;
      if (lane < np) bnd[lane] = lane == 0 ? t0 : prev + zh_first_barrier(bar, t0 + lane * ps - prev, t1 - prev);
      if (lane == 0) bnd[np] = t1;
   }
   return np;
}

// after a wave-level sync: does the task hold a piece of more than ZH_COOP_MIN positions?
// (zh_list_huge may be given a smaller bound for a small batch: a piece is one quad's chain of dependent steps, ~0.2 us each, and a call on one
// max-block waits for its longest piece in every pass — the chain kernel is three to four times faster per position)
__device__ inline bool zh_task_is_huge(const uint32_t *bnd, uint32_t np, uint32_t lane, uint32_t coop_min = ZH_COOP_MIN) {
   return zh_ballot(lane < np && bnd[lane + 1] - bnd[lane] > coop_min) != 0;
}

// ---- forward walk over the chosen parse of [t0, t1): histogram into LDS counters (blockdeflate.c:371-400) ----------
// Forward walks over a parse that issue NO vector-memory operation of their own (this one; the bit count of zh_post_tasks) go tile by tile with ZH_WALK_AHEAD tiles in
// flight, tracked by hand (zh_async_load_tile / zh_async_wait, zh_platform.h). A tile's work — a ballot, a few scalar hops, a handful of LDS updates — is a few hundred
// cycles, the round trip of its loads a microsecond, and ONE wave walks a whole task (a chain task: tens of thousands of positions). Rounds 2-5 requested "the next tile
// under this tile's work" with plain loads; the work contains a loop (zh_chain_mask), and around a loop the compiler waits for every load in flight, the youngest
// included, so the read-ahead was none (four tiles ahead with plain loads measured the same: profiles/r06_ab_results.txt). Measured, one 32 MiB run alone on the chip:
// zh_parse_chain 7.98 -> 7.42 ms over the four passes, zh_parse_lanes 9.76 -> 9.43, zh_post_tasks (with its byte ring) 1.02 -> 0.93; the step of the bench is within
// noise of where it was — the two parse kernels overlap. A lane whose position lies behind the range loads the range's last position and is masked.
#define ZH_WALK_AHEAD 4
#define ZH_WALK_BEGIN(best_, win_, prev_, t0_, t1_)                                                                                    \
   {                                                                                                                                   \
      zh_async_tile_t wq_[ZH_WALK_AHEAD];                                                                                              \
      zh_async_wait<0>();   /* (whatever the caller had in flight: the counts below are this walk's alone) */                         \
      _Pragma("unroll") for (uint32_t wu_ = 0; wu_ < ZH_WALK_AHEAD; wu_++) {                                                            \
         const uint32_t wp_ = min((t0_) + 64u * wu_ + zh_lane(), (t1_) - 1u);                                                          \
         zh_async_load_tile(wq_[wu_], (best_) + (wp_ - (prev_)), (win_) + wp_);                                                        \
      }                                                                                                                                \
      for (uint32_t wbase_ = (t0_); wbase_ < (t1_); wbase_ += 64u * ZH_WALK_AHEAD) {                                                   \
         _Pragma("unroll") for (uint32_t wu_ = 0; wu_ < ZH_WALK_AHEAD; wu_++) {                                                         \
            const uint32_t base = wbase_ + 64u * wu_;                                                                                  \
            if (base < (t1_)) {                                                                                                        \
               const uint32_t limit = min(64u, (t1_) - base);                                                                          \
               const uint32_t pos = base + zh_lane();                                                                                  \
               zh_async_wait<(ZH_WALK_AHEAD - 1) * ZH_ASYNC_TILE_LOADS>();   /* this tile's loads have landed: vector-memory operations complete in order */ \
               zh_async_landed(wq_[wu_]);                                                                                              \
               const uint32_t b = pos < (t1_) ? wq_[wu_].b : 0u, byte = pos < (t1_) ? (wq_[wu_].y & 0xffu) : 0u;                       \
               {                                                                                                                       \
                  const uint32_t wn_ = min(pos + 64u * ZH_WALK_AHEAD, (t1_) - 1u);                                                     \
                  zh_async_load_tile(wq_[wu_], (best_) + (wn_ - (prev_)), (win_) + wn_);                                               \
               }                                                                                                                       \
               (void)limit; (void)byte;
#define ZH_WALK_END            \
            }                  \
         }                     \
      }                        \
      zh_async_wait<0>();      \
   }

__device__ inline void zh_walk_histogram_wave(uint32_t *hist /* ZH_NSYM, zeroed */, const uint8_t *win, uint32_t prev, uint32_t t0, uint32_t t1,
                                              const uint32_t *best) {
   const uint32_t lane = zh_lane();
   uint32_t carry = 0;
   if (t0 < t1) ZH_WALK_BEGIN(best, win, prev, t0, t1)
      const uint32_t len = b & 0xffffu;
      const uint64_t mask = zh_chain_mask(len, carry, limit);
      if ((mask >> lane) & 1ull) {
         if (len >= ZH_MIN_MATCH) {
            atomicAdd(&hist[257 + zh_len_idx(len)], 1u);
            atomicAdd(&hist[ZH_NLIT + zh_dist_sym(b >> 16)], 1u);
         }
         else
            atomicAdd(&hist[byte], 1u);
      }
   ZH_WALK_END
   zh_wave_sync();
}

// ---- the parse kernel ---------------------------------------------------------------------------------------------
struct zh_parse_ws_t {
   union {
      uint16_t ring[4][ZH_RING + 16]; // per row (+32 B, likewise rec and tile +16 B: the four rows of a wave access their arrays at
                                     // the same offsets in the same instruction; with row strides that are multiples of 128 B
                                     // they would all hit the same LDS banks)
                                     // per row: cost[p mod ZH_RING] mod 2^16 of its current piece (the reference's cost[], blockdeflate.c:255)
      uint32_t hist[ZH_NSYM];        // after the parse: histogram of the task
   };
   uint4 rec[4][17];                 // per staged position: x = bitmap of short slot lengths 3..34, y/z = running minima
                                     // (distance price << 3 | slot) per short slot, w = see ZH_REC_* below
   uint16_t tile[4][17][ZH_NMATCH];  // per staged position and slot: len(9) | distance price(5) << 9 (the offset is re-read
                                     // from the match row when the winner is decoded: LDS is allocated in 2 KiB granules on
                                     // gfx950, and at <= 6 KiB 26 instead of 20 of these workgroups fit a CU)
   uint32_t bnd[ZH_MAXPIECES + 1];   // piece boundaries of the task
   uint8_t litprice[ZH_NLIT];        // code lengths with the 9-bit fill (blockdeflate.c:873-876)
   uint8_t lencost[256];             // price of length e+3 incl. extra bits (blockdeflate.c:216-219,263-264)
   uint8_t distcost[ZH_NDIST];       // price of a distance symbol incl. extra bits (blockdeflate.c:127-136)
};
// rec.w: bits 0..4 bitmap of lengths 35..39 | 5..7 their count | 8..11 number of long slots | 12..17 longest short
// length clamped to the sub-block end | 18..22 literal price
#define ZH_REC_NHI(w) (((w) >> 5) & 7u)
#define ZH_REC_NLONG(w) (((w) >> 8) & 15u)
#define ZH_REC_KMAX(w) (((w) >> 12) & 63u)
#define ZH_REC_LIT(w) (((w) >> 18) & 31u)

struct zh_tile_regs_t {
   uint4 a, b;      // the 8 match slots of this lane's position
   uint32_t byte;
};

// Digest of one position's 8 match slots into its tile entries and its 16-byte record (branch-free).
__device__ __forceinline__ void zh_stage_position(zh_parse_ws_t &ws, uint32_t row, uint32_t slot, const zh_tile_regs_t &regs, uint32_t room) {
   const uint32_t raw[ZH_NMATCH] = {regs.a.x, regs.a.y, regs.a.z, regs.a.w, regs.b.x, regs.b.y, regs.b.z, regs.b.w};
   uint32_t nlong = 0, nshort = 0, kmax = 0, run = 0xFFu;
   uint64_t pm = 0, lmask = 0;
#pragma unroll
   for (uint32_t m = 0; m < ZH_NMATCH; m++) {
      const uint32_t len = raw[m] & 0xffffu, off = raw[m] >> 16;
      const bool valid = len >= ZH_MIN_MATCH;
      const bool is_long = len >= ZH_LEAVE_ALONE;
      const bool is_short = valid && !is_long;
      const uint32_t oc = (uint32_t)ws.distcost[zh_dist_sym(valid ? off : 1u)];
      ws.tile[row][slot][m] = valid ? (uint16_t)(len | (oc << 9)) : (uint16_t)0;
      nlong += is_long ? 1u : 0u;
      kmax = max(kmax, is_short ? len : 0u);               // the first short slot is the longest
      lmask |= is_short ? (1ull << (len - ZH_MIN_MATCH)) : 0ull;
      run = is_short ? min(run, (oc << 3) | m) : run;
      pm |= is_short ? ((uint64_t)run << (8 * nshort)) : 0ull;
      nshort += is_short ? 1u : 0u;
   }
   const uint32_t mask_hi = (uint32_t)(lmask >> 32);
   uint4 r;
   r.x = (uint32_t)lmask;
   r.y = (uint32_t)pm;
   r.z = (uint32_t)(pm >> 32);
   // room = end clamp (blockdeflate.c:283-284); a no-op away from the sub-block end
   r.w = mask_hi | ((uint32_t)__popc(mask_hi) << 5) | (nlong << 8) | (min(kmax, room) << 12) | ((uint32_t)ws.litprice[regs.byte & 0xffu] << 18);
   ws.rec[row][slot] = r;
}

// This lane's best match candidate for the position at p with record R (tile entries at tile[trow][tslot]), costs in
// `ring`: lane s prices length 3+s, and in the rarely executed section (taken by the whole wave when any position needs
// it) lengths 19+s, 35+s and long slot s. Key = (cost - base) << 9 | slot << 6 | (39 - k).
template <bool SEG>
__device__ __forceinline__ uint32_t zh_lane_key(zh_parse_ws_t &ws, const uint16_t *ring, uint32_t trow, uint32_t tslot, const uint4 &R, uint32_t p, uint32_t pm,
                                                uint32_t s, uint32_t base, uint32_t lc0, uint32_t lc1, uint32_t lc2, uint32_t sb_end) {
   const uint32_t kmax = ZH_REC_KMAX(R.w), nlong = ZH_REC_NLONG(R.w), nhi = ZH_REC_NHI(R.w);
   uint32_t key = 0xFFFFFFFFu;
   if (3 + s <= kmax) {
      const uint32_t sel = (uint32_t)__popc(R.x >> s) + nhi - 1u;            // index of the last short slot reaching 3+s
      const uint32_t b = ((sel < 4 ? R.y : R.z) >> ((sel & 3u) * 8u)) & 0xffu;
      const uint32_t c = (lc0 + (b >> 3) + (uint32_t)ring[zh_ring_wrap(pm + 3 + s)] - base) & 0xffffu;
      key = (c << 9) | ((b & 7u) << 6) | (36u - s);                          // 39 - k
   }
   // rarely needed: lengths 19..39, and slots stored with length >= 40
   if (zh_ballot(kmax > 18u || nlong != 0)) {
      if (19 + s <= kmax) {
         const uint32_t sel = (uint32_t)__popc(R.x >> (16 + s)) + nhi - 1u;
         const uint32_t b = ((sel < 4 ? R.y : R.z) >> ((sel & 3u) * 8u)) & 0xffu;
         const uint32_t c = (lc1 + (b >> 3) + (uint32_t)ring[zh_ring_wrap(pm + 19 + s)] - base) & 0xffffu;
         key = min(key, (c << 9) | ((b & 7u) << 6) | (20u - s));
      }
      if (s < 5 && 35 + s <= kmax) {
         const uint32_t sel = (uint32_t)__popc((R.w & 31u) >> s) - 1u;
         const uint32_t b = ((sel < 4 ? R.y : R.z) >> ((sel & 3u) * 8u)) & 0xffu;
         const uint32_t c = (lc2 + (b >> 3) + (uint32_t)ring[zh_ring_wrap(pm + 35 + s)] - base) & 0xffffu;
         key = min(key, (c << 9) | ((b & 7u) << 6) | (4u - s));
      }
      if (s < nlong) {                                                        // long slot s: full (clamped) length only
         const uint32_t e = ws.tile[trow][tslot][s];
         const uint32_t mlen = min(e & 511u, (SEG ? ws.bnd[16 + trow] : sb_end) - p);   // (SEG: the row's own end, see zh_parse_one_task)
         uint32_t enc = mlen - ZH_MIN_MATCH;                                  // wraps below 3, then saturates (:289, :216-219)
         if (enc > 255) enc = 255;
         const uint32_t c = ((uint32_t)ws.lencost[enc] + ((e >> 9) & 31u) + (uint32_t)ring[zh_ring_wrap(pm + mlen)] - base) & 0xffffu;
         key = min(key, (c << 9) | (s << 6));
      }
   }
   return key;
}

// decode the winning (slot, length) of a staged position into the parse entry; `row8` = the position's match row (8 x
// len | offset << 16), which the same lane loaded one tile earlier (an L2 hit)
__device__ __forceinline__ uint32_t zh_decode_pick(zh_parse_ws_t &ws, uint32_t trow, uint32_t tslot, uint32_t kk, uint32_t room, const uint32_t *row_lo,
                                                   const uint32_t *row_hi) {
   if (kk == 0xFFFFFFFFu) return 0;
   const uint32_t m = (kk >> 6) & 7u;
   const uint32_t e = m < 4 ? row_lo[m] : row_hi[m - 4];
   const uint32_t nlong = ZH_REC_NLONG(ws.rec[trow][tslot].w);
   const uint32_t len = (m < nlong) ? min(e & 0xffffu, room) : (39u - (kk & 63u));
   return len | (e & 0xffff0000u);
}

// One task, by one wave (the calling workgroup); ws = its LDS workspace. SEG: the task is a cut one (see above) and the wave's
// pieces are its segments seg_k0 .. seg_k0 + 3 of seg_K, their cost vectors at slot seg_slot0 + k of `vecs`. The wave that
// finishes last among the task's waves (*seg_done counts them) checks the cuts from the right, parses the segments whose
// speculated costs do not match again — one at a time, on row 0, from the costs their right neighbour left — and takes the
// task's histogram.
template <bool SEG>
__device__ __forceinline__ void zh_parse_one_task(zh_parse_ws_t &ws, uint32_t gt, const uint8_t *__restrict__ data, const zh_block_t *__restrict__ blocks,
                                         const zh_match_t *__restrict__ match, uint64_t match_stride, const uint64_t *__restrict__ bars, uint64_t bar_stride,
                                         const zh_work_t *__restrict__ work, const uint2 *__restrict__ taskmap, const zh_sbstate_t *__restrict__ states,
                                         uint32_t *best_all, uint64_t best_stride, uint32_t *hist_part, int pass, uint32_t seg_KS, uint32_t seg_k0, uint32_t seg_slot0,
                                         int16_t *vecs, uint32_t *seg_done, uint32_t *cnt /* the run's counters */, uint32_t *demote_list, uint32_t demote_min) {
   uint32_t *seg_failed = SEG ? cnt + ZH_CNT_SEG_FAILED : (uint32_t *)NULL;
   uint32_t seg_nfail = 0;   // (SEG, checker) cuts of the task that failed in this pass
   const uint32_t seg_K = ZH_CUT_K(seg_KS), seg_S = ZH_CUT_S(seg_KS);
   const uint2 tm = taskmap[gt];
   const zh_work_t wk = work[tm.x];
   const zh_sbstate_t *st = states + tm.x;
   if (st->failed) return;
   if (!st->is_dynamic && pass > 0) return;   // static sub-blocks are parsed once (blockdeflate.c:836-858)
   if (st->settled) return;                   // same prices as in the last pass: same parse, same histogram (zh_sb_build_one)

   const zh_block_t blk = blocks[wk.block];
   const uint8_t *win = data + blk.win_off;
   const uint32_t prev = blk.prev;
   const uint4 *rows = (const uint4 *)(match + (uint64_t)wk.block * match_stride);   // row r = pos - prev: slots 0..3 (zh_common.h)
   const uint4 *rows_hi = rows + ZH_ROW_HI_OFF(match_stride);                        // ... and 4..7, present when slot 3 holds a match
   const uint64_t *bar = bars + (uint64_t)wk.block * bar_stride;
   uint32_t *best = best_all + (uint64_t)wk.block * best_stride;
   const uint32_t lane = zh_lane(), row = lane >> 4, s = lane & 15;
   const uint32_t sb_end = wk.start + wk.size;

   // ---- prices of the codes in force; unused symbols price at 9 / 6 bits (blockdeflate.c:873-881) ---------------
   for (uint32_t k = lane; k < ZH_NLIT; k += 64) {
      const uint32_t l = st->lit_len[k];
      ws.litprice[k] = (uint8_t)(l ? l : 9u);
   }
   if (lane < ZH_NDIST) {
      const uint32_t l = st->dist_len[lane];
      ws.distcost[lane] = (uint8_t)((l ? l : 6u) + (uint32_t)zh_dist_xbits((int)lane));
   }
   zh_sync();
   for (uint32_t e = lane; e < 256; e += 64) {
      const int idx = zh_len_idx(e + 3);
      ws.lencost[e] = (uint8_t)(ws.litprice[257 + idx] + zh_lenidx_xbits(idx));
   }
   // ---- task range and its pieces ------------------------------------------------------------------------------
   const uint32_t t0 = zh_task_boundary(bar, prev, wk.start, sb_end, tm.y, wk.ntasks);
   const uint32_t t1 = zh_task_boundary(bar, prev, wk.start, sb_end, tm.y + 1, wk.ntasks);
   uint32_t np;
   if (SEG) {
      // row j takes segment k = seg_k0 + j: bnd[j] = its low end, [8 + j] = where its recurrence starts, [16 + j] = where its candidates
      // end at the latest, [24 + j] = parse entries from here on are the right neighbour's, [32 + j] = vector slot | bit 31: has a
      // left neighbour
      np = min((uint32_t)ZH_CUT_ROWS, seg_K - seg_k0);
      if (lane < np) {
         const uint32_t k = seg_k0 + lane;
         const bool last = k + 1u == seg_K;
         const uint32_t b = t1 - (seg_K - 1u - k) * seg_S;
         ws.bnd[lane] = k ? b - seg_S : t0;
         ws.bnd[8 + lane] = last ? t1 : b + ZH_CUT_WARM;
         ws.bnd[16 + lane] = last ? sb_end : b + ZH_CUT_WARM;
         ws.bnd[24 + lane] = last ? 0xFFFFFFFFu : b;
         ws.bnd[32 + lane] = (seg_slot0 + k) | (k ? 0x80000000u : 0u);
      }
      zh_sync();
   }
   else {
      np = zh_task_pieces(ws.bnd, bar, prev, t0, t1, lane);
      zh_sync();
      // a barrier-free run of more than ZH_COOP_MIN positions: four independent recurrences are of no use when the task is
      // (mostly) one run. Such tasks are listed by zh_list_huge: cut into segments and back here (SEG), or parsed as one chain by
      // zh_parse_chain, next to this kernel.
      if (zh_task_is_huge(ws.bnd, np, lane)) return;
   }
   const uint32_t lc0 = ws.lencost[s], lc1 = ws.lencost[16 + s], lc2 = ws.lencost[32 + (s & 7)];

   bool seg_checker = false, seg_import = false;   // (SEG) this wave checks the task's cuts; the round parses a segment again
   uint32_t seg_exact = seg_K;                     // (SEG, checker) the segments from this one on are exact
   for (;;) {   // (SEG: the speculative round, then the checker's rounds; otherwise once)
   // ---- row scheduler state (uniform within a row) ---------------------------------------------------------------
   int32_t next_piece = (int32_t)np - 1;   // wave-uniform: pieces are handed out from the task's end
   uint32_t p_lo = 0, p_hi = 0;            // what is left of the row's piece: [p_lo, p_hi)
   uint32_t n_lo = 0, n_cnt = 0;           // the prefetched tile: positions n_lo + 0..n_cnt-1
   bool n_top = false;                     // the prefetched tile is the top of its piece (recurrence restarts there)
   uint32_t cost_next = 0;                 // cost[p+1] mod 2^16
   zh_tile_regs_t regs;

   // picks the row's next tile (popping a new piece if the current one is used up) and issues its loads
#define ZH_NEXT_TILE()                                                                                         \
   do {                                                                                                        \
      n_top = false;                                                                                           \
      for (uint32_t r_ = 0; r_ < 4; r_++) {                                                                    \
         const bool need_ = zh_readlane((uint32_t)(p_hi <= p_lo), (int)(r_ * 16)) != 0;                        \
         if (need_) {                                                                                          \
            uint32_t lo_ = 0, hi_ = 0;                                                                         \
            while (next_piece >= 0 && hi_ <= lo_) {                                                            \
               lo_ = ws.bnd[next_piece];                                                                       \
               hi_ = ws.bnd[next_piece + 1];                                                                   \
               next_piece--;                                                                                   \
            }                                                                                                  \
            if (row == r_ && hi_ > lo_) {                                                                      \
               p_lo = lo_;                                                                                     \
               p_hi = hi_;                                                                                     \
               n_top = true;                                                                                   \
            }                                                                                                  \
         }                                                                                                     \
      }                                                                                                        \
      n_cnt = min(16u, p_hi > p_lo ? p_hi - p_lo : 0u);                                                        \
      n_lo = p_hi - n_cnt;                                                                                     \
      p_hi = n_lo;                                                                                             \
      {                                                                                                        \
         const bool ok_ = s < n_cnt;                                                                           \
         const uint32_t pos_ = ok_ ? n_lo + s : wk.start;   /* clamped: the loads are always legal */          \
         const uint4 z_ = {0, 0, 0, 0};                                                                        \
         const uint4 a_ = rows[pos_ - prev];                                                                   \
         uint4 b_ = z_;                                                                                        \
         if (ok_ && (a_.w & 0xffffu) >= ZH_MIN_MATCH) b_ = rows_hi[pos_ - prev];                               \
         const uint32_t y_ = win[pos_];                                                                        \
         regs.a = ok_ ? a_ : z_;                                                                               \
         regs.b = ok_ ? b_ : z_;                                                                               \
         regs.byte = y_;                                                                                       \
      }                                                                                                        \
   } while (0)

   if (SEG) {
      // one segment per row, no scheduler
      next_piece = -1;
      if (row < np) {
         p_lo = ws.bnd[row];
         p_hi = ws.bnd[8 + row];
      }
   }
   if (np) ZH_NEXT_TILE();
   if (SEG) n_top = row < np;

   while (np) {
      const uint32_t c_lo = n_lo, c_cnt = n_cnt;
      const bool c_top = n_top;
      if (!zh_ballot(c_cnt != 0)) break;

      // ---- stage the tile: every lane digests the 8 slots of its own position (lanes beyond the tile's count digest
      //      zeros into their own, unused, record) ---------------------------------------------------------------------
      zh_stage_position(ws, row, s, regs, (SEG ? ws.bnd[16 + row] : sb_end) - (c_lo + s));   // (SEG: the row's candidates end at its own, made-up, end)
      const uint32_t c_lo_m = c_lo % ZH_RING;   // once per tile; the steps below keep indices reduced
      if (c_top && s == 0) ws.ring[row][zh_ring_wrap(c_lo_m + c_cnt)] = 0;   // cost[piece end] = 0
      if (c_top) cost_next = 0;
      if (SEG && seg_import && c_top) {
         // ... and the costs of the 258 positions above it, as the segment to the right computed them (relative to cost[piece end])
         const uint32_t *vl = (const uint32_t *)(vecs + (uint64_t)((ws.bnd[32 + row] & 0x7fffffffu) + 1u) * (2u * ZH_VEC) + ZH_VEC);
         const uint32_t e_m = zh_ring_wrap(c_lo_m + c_cnt);
         for (uint32_t i = s; i < (ZH_VEC_LIVE + 1u) / 2u; i += 16) {
            const uint32_t w2 = zh_load_agent_u32(vl + i);
            ws.ring[row][(e_m + 2u * i) % ZH_RING] = (uint16_t)w2;
            ws.ring[row][(e_m + 2u * i + 1u) % ZH_RING] = (uint16_t)(w2 >> 16);   // (entry 259 is padding: a ring slot nobody reads before it is rewritten)
         }
      }
      zh_sync();
      ZH_NEXT_TILE();   // the next tile's loads complete while this one is priced

      // ---- price the tile: one position per row per step ------------------------------------------------------------
      const uint32_t steps = max(max(zh_readlane(c_cnt, 0), zh_readlane(c_cnt, 16)), max(zh_readlane(c_cnt, 32), zh_readlane(c_cnt, 48)));
      for (uint32_t t = 0; t < steps; t++) {
         const bool act = t < c_cnt;
         const uint32_t a = act ? c_cnt - 1 - t : 0;
         const uint32_t p = c_lo + a;
         uint4 R = ws.rec[row][a];
         if (!act) R.w = 0;
         const uint32_t base = cost_next - ZH_KEY_BIAS;   // key cost = (candidate cost - base) mod 2^16, below 2^15
         const uint32_t pm = zh_ring_wrap(c_lo_m + a);
         const uint32_t key = zh_lane_key<SEG>(ws, ws.ring[row], row, a, R, p, pm, s, base, lc0, lc1, lc2, sb_end);
         const uint32_t rkey = zh_row_min(key);   // every lane of a row now holds that row's best match candidate
         // literal first; a match must be strictly cheaper (:292,:307). An absent candidate (all ones) prices at 2^23-1,
         // above any literal (5 bits + bias).
         const uint32_t lit = ZH_REC_LIT(R.w) + ZH_KEY_BIAS, mc = rkey >> 9;
         const bool take = mc < lit;
         const uint32_t c = (base + (take ? mc : lit)) & 0xffffu;
         if (act) {
            if (s == 0) {
               ws.ring[row][pm] = (uint16_t)c;
               ws.rec[row][a].x = take ? rkey : 0xFFFFFFFFu;   // the record's length bitmap is dead now: its place takes the winning key
            }
            cost_next = c;
         }
      }
      zh_sync();
      // ---- flush: decode the winning (slot, length) of each position and store the parse --------------------------
      if (s < c_cnt && (!SEG || c_lo < ws.bnd[24 + row]))   // (the boundary is a tile boundary)
         best[(c_lo + s) - prev] = zh_decode_pick(ws, row, s, ws.rec[row][s].x, sb_end - (c_lo + s), (const uint32_t *)(rows + (c_lo + s - prev)),
                                                  (const uint32_t *)(rows_hi + (c_lo + s - prev)));
      if (SEG) {
         // the relative costs of [c_lo, c_lo + 258] — all in the ring — when the row has just passed its cut (speculated: what the
         // right neighbour will be compared with) or finished its segment (what the left neighbour will be compared with)
         const uint32_t info = ws.bnd[32 + row];
         const bool spec = c_cnt != 0 && c_lo == ws.bnd[24 + row];
         const bool left = c_cnt != 0 && c_lo == ws.bnd[row] && (info >> 31) != 0;
         if (zh_ballot(spec || left)) {
            if (spec || left) {
               int16_t *v = vecs + (uint64_t)(info & 0x7fffffffu) * (2u * ZH_VEC) + (spec ? 0u : ZH_VEC);
               const uint32_t x = c_lo % ZH_RING;
               const uint32_t base0 = ws.ring[row][x];
               for (uint32_t i = s; i < ZH_VEC_LIVE + 1u; i += 16)   // (an even number of entries: the vectors are compared word-wise)
                  v[i] = i < ZH_VEC_LIVE ? (int16_t)(uint16_t)(ws.ring[row][(x + i) % ZH_RING] - base0) : (int16_t)0;
            }
         }
      }
      zh_sync();
   }
#undef ZH_NEXT_TILE
      if (!SEG) break;
      // ---- the task's waves are counted; the last one checks the cuts, right to left ----------------------------------------
      __threadfence();   // this round's parse entries and vectors are out
      if (!seg_checker) {
         uint32_t old = 0;
         if (lane == 0) old = atomicAdd(seg_done, 1u);
         old = zh_readfirstlane(old);
         if (((old & 0xffffu) + 1u) % ((seg_K + ZH_CUT_ROWS - 1u) / ZH_CUT_ROWS) != 0u) return;   // (the counter runs on over the passes; its high half counts the task's failed cuts)
         __threadfence();   // ... and the other waves' are in
         seg_checker = true;
         seg_exact = seg_K - 1u;   // the last segment started from the task's end: exact
      }
      bool again = false;
      while (seg_exact > 0) {
         const uint32_t k = seg_exact - 1u;
         const uint32_t *vs = (const uint32_t *)(vecs + (uint64_t)(seg_slot0 + k) * (2u * ZH_VEC));               // speculated at the cut
         const uint32_t *vl = (const uint32_t *)(vecs + (uint64_t)(seg_slot0 + k + 1u) * (2u * ZH_VEC) + ZH_VEC);   // computed by the (exact) right neighbour
         bool bad = false;
         for (uint32_t i = lane; i < (ZH_VEC_LIVE + 1u) / 2u; i += 64) bad |= zh_load_agent_u32(vs + i) != zh_load_agent_u32(vl + i);
         seg_exact = k;
         if (zh_ballot(bad)) {
            again = true;
            break;
         }
      }
      if (!again) break;
      zh_sync();
      if (lane == 0) {
         const uint32_t k = seg_exact;
         const uint32_t b = t1 - (seg_K - 1u - k) * seg_S;
         ws.bnd[0] = k ? b - seg_S : t0;
         ws.bnd[8] = b;
         ws.bnd[16] = sb_end;
         ws.bnd[24] = 0xFFFFFFFFu;
         ws.bnd[32] = (seg_slot0 + k) | (k ? 0x80000000u : 0u);
         atomicAdd(seg_failed, 1u);   // statistics
         atomicAdd(seg_done, 0x10000u);
      }
      seg_nfail++;
      np = 1;
      seg_import = true;
      zh_sync();
   }

   // A failed cut is parsed again by this one wave on one of its rows, cut after cut: a task with several of them holds its pass up
   // for milliseconds. Speculation that failed under one pass's prices mostly fails under the next: such a task goes
   // to zh_parse_chain as ONE chain for the passes left — a workgroup there steps several times faster than a row — by an entry in the
   // run's fourth chain list; the segment workgroups skip it from then on.
   if (SEG && demote_min && seg_nfail >= demote_min && pass < 3 && lane == 0) {
      atomicOr(seg_done - 2, ZH_CUT_DEMOTED);   // (segtasks[].y; seg_done points at .w, the counter)
      demote_list[atomicAdd(&cnt[ZH_CNT_DEMOTED], 1u)] = gt;
      atomicAdd(&cnt[ZH_CNT_DEMOTED_PASS + pass], 1u);
   }
   // ---- histogram of the task's parse; the per-sub-block sum is taken by zh_sb_build ------------------------------------
   if (st->is_dynamic) {
      __threadfence_block();
      zh_sync();
      for (uint32_t k = lane; k < ZH_NSYM; k += 64) ws.hist[k] = 0;
      zh_sync();
      zh_walk_histogram_wave(ws.hist, win, prev, t0, t1, best);
      uint32_t *hp = hist_part + (uint64_t)gt * ZH_NSYM;
      for (uint32_t k = lane; k < ZH_NSYM; k += 64) hp[k] = ws.hist[k];
   }
}

// Which of the two ways a run's cut tasks are parsed (see "speculative segments" above): many segments are throughput — four to a wave of
// the segment workgroups of zh_parse_lanes' launch, which also check them; a few are latency — each one a job of zh_parse_chain (five times faster per position). Decided on
// the device from the run's counters (zh_list_huge), the same way by both kernels (zh_parse_lanes takes the wide case in the first workgroups of its grid): the host launches both and never reads the counts.
__device__ __forceinline__ bool zh_segments_are_wide(const uint32_t *cnt, uint32_t seg_wide_min) {
   return cnt[ZH_CNT_SEGTASKS] != 0 && cnt[ZH_CNT_SEGITEMS] >= seg_wide_min;
}
