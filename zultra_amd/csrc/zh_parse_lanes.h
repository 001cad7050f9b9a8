// zh_parse_lanes.h — the backward optimal parse (reference src/blockdeflate.c:254-323, zultra_optimize_matches_lwd) with a
// QUAD OF LANES PER PIECE: a wave advances 16 independent recurrences per step, and a step is ~140 wave instructions.
//
// Why. Round 2's kernel (zh_parse_tasks; its row machinery lives on in zh_parse.h for the speculative segments) gave a piece to a 16-lane row, lane s pricing length 3+s: whatever the position offers, a
// step costs the same ~120 wave instructions for four positions (profiles/r02e_sq_counters.csv: 31.7 vector instructions per
// position and pass, the vector unit saturated). What a position needs is much less once the work is ordered differently.
//
// One step for the position p of a piece (a barrier-to-barrier run, zh_parse.h), by the four lanes q = 0..3 of its quad:
//   * the costs a step looks at — cost[p + 3 .. p + 39] — live in REGISTERS: lane q keeps cost[p + 3 + 9 q .. p + 12 + 9 q] (<< 9, ready
//     to take a price), and moving to p - 1 shifts the window by one: the new element of lane q is the one that leaves lane q - 1
//     (one DPP move), lane 0's is the cost decided three steps earlier. No LDS ring: costs farther than 39 positions ahead are
//     only ever needed by slots stored with length >= 40, and those read them back from a uint16 array in global memory that
//     every step appends to (fetched with the rows, two batches ahead). A piece is at most ZH_COOP_MIN positions and a position
//     costs at most 15 bits: absolute costs from the piece's end fit 16 bits.
//   * prefix minima over the lengths: G(k) = cost[p+k] + price of length k, P(L) = min over k <= L of (G(k), larger k first),
//     L = 3..39 — none of it depends on the position's matches. Lane q scans the ten lengths 3+9q .. 12+9q, the lanes' totals are
//     combined with two DPP steps, P goes to an LDS scratch [L][piece].
//   * the reference tries every pair (slot m, k <= len_m) in the order m ascending, k descending, and takes strict improvements
//     only: the winner is the minimum of (cost, m, -k). For slot m all of its k's at once: P(min(len_m, room)) + distance price.
//     So each slot costs one gather from the scratch and one add (lane q takes slots q and 4+q); slots stored with length >= 40
//     are tried at their clamped length only (blockdeflate.c:286-297): one cost from the array in global memory.
//   * a quad minimum picks the position's best match; literal first, a match must be strictly cheaper (:292,:307).
// Keys: cost << 9 | slot << 6 | (39 - k), as in zh_parse.h.
// (A lane per piece — 64 recurrences per wave — needs 4 x the LDS per wave, leaves one wave per SIMD and, measured, 10 to 20 of
// the 64 lanes with a position: barrier-free runs of a few hundred positions are common enough that one of 64 pieces always is one.)
//
// Memory: the quad fetches four positions per request — lane q the row, the second plane of the row and the byte of position
// p-1-q — two batches ahead of their use, stages them in LDS, and stores the four parse entries of a batch with one instruction.
//
// Work: a wave takes ZH_LP_TASKS consecutive tasks of the run's task list (zh_parse.h: a task = ~2048 positions between two
// barriers), groups them by sub-block (prices differ), cuts each task into pieces (zh_task_pieces) and hands the
// pieces to its quads, the long ones first, a new one whenever a quad runs out. Tasks with a barrier-free run of more than
// ZH_COOP_MIN positions are left to zh_parse_chain / the segment workgroups of this launch (zh_list_huge lists them with the same
// test). The histogram of the chosen parse (blockdeflate.c:371-400) is taken by the 64 lanes, each walking pieces forward (a
// piece starts on a token boundary), and stored in the slot of the group's first task (zh_sb_build sums a sub-block's slots).
#pragma once
#include <zh_platform.h>
#include "zh_common.h"
#include "zh_parse.h"

#ifndef ZH_LP_TASKS
#define ZH_LP_TASKS 8u             // tasks per wave
#endif
#define ZH_LP_C 16u                // pieces in flight per wave (one per quad)
#define ZH_LP_KPL 10u              // lengths per lane: lane q scans 3 + 9 q .. 12 + 9 q (neighbours share one: a stride of 9 ring rows
                                   // puts the four lanes of a quad on different LDS banks, a stride of 10 puts lanes 0 and 2 on the same)
#define ZH_LP_QSTRIDE 9u
#define ZH_LP_NL 37u               // rows of the prefix scratch: lengths 3 .. 39 (ZH_LEAVE_ALONE - 1)
#define ZH_LP_PPT ((ZH_TASK + ZH_COOP_MIN + ZH_PIECE - 1) / ZH_PIECE <= 32 ? 32u : 64u)   // pieces a task can have: fewer than 32 with the default piece size, at most ZH_MAXPIECES
#define ZH_LP_MAXP (ZH_LP_TASKS * ZH_LP_PPT) // pieces per group: a task without a run of more than ZH_COOP_MIN positions has fewer than ZH_LP_PPT
#define ZH_LP_LONG 192u            // pieces of at least this many positions are handed out first
#define ZH_LP_NOKEY 0xFFFFFFFFu
// a task that is parsed here has no barrier-free run of more than ZH_COOP_MIN positions, so it has fewer than 32 pieces of >= ZH_PIECE positions
// (zh_task_pieces); a group whose pieces outgrew ZH_LP_MAXP would silently lose a task — variant builds (-D...) must keep these
static_assert((ZH_TASK + ZH_COOP_MIN + ZH_PIECE - 1) / ZH_PIECE <= ZH_LP_PPT && ZH_LP_PPT <= ZH_MAXPIECES, "pieces per task must stay within ZH_LP_PPT (ZH_LP_MAXP = ZH_LP_PPT per task)");
static_assert(ZH_LP_TASKS >= 1 && ZH_LP_TASKS <= 64, "the parsed-task mask of a group is 64 bits");
// The prefix scratch [row][column]: row L - 3, and piece i in column i ^ 8 for the rows of lanes 2 and 3 — the four lanes of a quad
// write rows 9 q + k, whose LDS banks (16 per row) would coincide for lanes 0 / 2 and for lanes 1 / 3
#define ZH_LP_PCOL(row_, piece_) ((piece_) ^ ((row_) >= 2u * ZH_LP_QSTRIDE ? 8u : 0u))

#ifdef ZH_LP_PROFILE
// probe builds only (tools/lp_profile.py): 0 steps, 1 quad-steps with a position, 2 cycles of the step loops, 3 groups, 4 cycles of the
// group setup, 5 cycles of the histogram walks, 6 pieces, 7 batches with a second plane
__device__ unsigned long long zh_lp_prof[16];   // 8..11: cycles of the batch staging, stage C, stage B, stage A of the step loop; 12..14: of the histogram phase — waiting for the
                                                // group's own stores and clearing the counters, the walks, storing the counters
#define ZH_LP_COUNT(slot_, n_) do { if (zh_lane() == 0) atomicAdd(&zh_lp_prof[slot_], (unsigned long long)(n_)); } while (0)
#define ZH_LP_CLOCK() zh_clock()
#else
#define ZH_LP_COUNT(slot_, n_)
#define ZH_LP_CLOCK() 0
#endif

#ifdef ZH_LP_TRACE
// probe builds only (tools/lp_trace.py): per pass and ticket of the LAST launch that ran {wall clock when the wave started, when it took the ticket, when it was done with
// it, blockIdx << 32 | HW_ID}
#define ZH_LP_TRACE_SLOTS 8192u
__device__ uint64_t zh_lp_trace[4u * ZH_LP_TRACE_SLOTS * 4u];
#endif

struct alignas(16) zh_lp_ws_t {
   union {
      struct {
         uint32_t pmin[ZH_LP_NL][ZH_LP_C];      // [L - 3][piece, see ZH_LP_PCOL]: prefix minimum P(L) of this step
         uint32_t stage0[ZH_LP_C][4][4];        // the batch: [piece][entry][slot 0..3] of the entries' rows,
         uint32_t stage1[ZH_LP_C][4][4];        //            slots 4..7,
         union {
            uint16_t stagef[ZH_LP_C][4][8];     //            per slot stored with length >= 40 the cost behind it
            uint4 stagef4[ZH_LP_C][4];          //            (written 16 bytes at a time),
         };
         uint32_t stageb[ZH_LP_C][4];           //            the entries' bytes
         uint32_t outp[2][ZH_LP_C][4];          // the parse entries of two batches (by batch parity) ...
         uint32_t outc[2][ZH_LP_C][4];          // ... and their costs
      };
      uint32_t hist[ZH_NSYM];                   // after the parse: histogram of the group
   };
   uint32_t plo[ZH_LP_MAXP], phi[ZH_LP_MAXP];   // the group's pieces: the long ones from the front, the others from the back
   uint32_t bnd[ZH_MAXPIECES + 1];
   uint8_t litprice[ZH_NLIT];             // code lengths with the 9-bit fill (blockdeflate.c:873-876)
   uint8_t lencost[256];                  // price of length e+3 incl. extra bits (blockdeflate.c:216-219,263-264)
   uint8_t distprice[512];                // price of a distance incl. extra bits, by the reference's table index (blockdeflate.c:45-58,127-136)
};

// index of distance d = v + 1 into the 512-entry distance tables of the reference (blockdeflate.c:45-58): v below 256 -> v, else
// 256 + ((v - 256) >> 7)  [= 254 + (v >> 7), which is >= v exactly up to v = 255]
__device__ __forceinline__ uint32_t zh_lp_dist_index(uint32_t v) { return min(v, 254u + (v >> 7)); }
__device__ __forceinline__ uint32_t zh_lp_index_dist(uint32_t idx) { return idx < 256u ? idx + 1u : ((idx - 254u) << 7) + 1u; }   // smallest distance of the index

// a batch of up to four positions of one piece: its entries off .. off + n - 1 are the positions p0, p0 - 1, ...; fresh = p0 is the
// piece's last position (the recurrence starts there). A piece's first batch has at most three positions, in its LAST entries: the
// step before a piece is always an empty one, which zeroes its ring slot — cost[piece end] — and nothing interrupts a piece.
struct zh_lp_batch_t {
   uint32_t p0, n, off, end;   // end = the piece's end: the cost there is 0
   bool fresh;
};

// Parses the tasks [g0, g1) of one sub-block (all of sub-block tm.x). All 64 lanes of ONE wave call; the workspace is the wave's own, and the
// syncs are the wave's (zh_wave_sync): the wave may be one of several of a workgroup.
__device__ __forceinline__ void zh_lp_group(zh_lp_ws_t &ws, uint32_t g0, uint32_t g1, const uint8_t *__restrict__ data, const zh_block_t *__restrict__ blocks,
                                            const zh_match_t *__restrict__ match, uint64_t match_stride, const uint64_t *__restrict__ bars, uint64_t bar_stride,
                                            const zh_work_t *__restrict__ work, const uint2 *__restrict__ taskmap, const zh_sbstate_t *__restrict__ states,
                                            uint32_t *best_all, uint64_t best_stride, uint16_t *cost_all, uint32_t *hist_part, int pass, const uint2 *__restrict__ taskinfo) {
   const uint2 tm0 = taskmap[g0];
   const zh_work_t wk = work[tm0.x];
   const zh_sbstate_t *st = states + tm0.x;
   if (st->failed) return;
   const bool sb_dynamic = st->is_dynamic != 0;
   if (!sb_dynamic && pass > 0) return;   // static sub-blocks are parsed once (blockdeflate.c:836-858)
   if (st->settled) return;               // same prices as in the last pass: same parse, same histogram (zh_sb_build_one)
   const zh_block_t blk = blocks[wk.block];
   const uint8_t *win = data + blk.win_off;
   const uint32_t prev = blk.prev;
   const uint4 *rows = (const uint4 *)(match + (uint64_t)wk.block * match_stride);   // row r = pos - prev: slots 0..3 (zh_common.h)
   const uint4 *rows_hi = rows + ZH_ROW_HI_OFF(match_stride);                        // ... and 4..7, present when slot 3 holds a match
   const uint64_t *bar = bars + (uint64_t)wk.block * bar_stride;
   uint32_t *best = best_all + (uint64_t)wk.block * best_stride;
   uint16_t *costs = cost_all + (uint64_t)wk.block * best_stride;   // cost of every position from its piece's end, for the slots of length >= 40
   const uint32_t lane = zh_lane(), q = lane & 3u, piece = lane >> 2;
   const uint32_t sb_end = wk.start + wk.size;
   const uint64_t lt_mask = (1ull << lane) - 1ull, lt_quad = (1ull << (lane & ~3u)) - 1ull;

   const uint64_t tic0 = ZH_LP_CLOCK();
   zh_wave_sync();   // the previous group is done with the workspace
   // ---- prices of the codes in force; unused symbols price at 9 / 6 bits (blockdeflate.c:873-881) ---------------
   for (uint32_t k = lane; k < ZH_NLIT; k += 64) {
      const uint32_t l = st->lit_len[k];
      ws.litprice[k] = (uint8_t)(l ? l : 9u);
   }
   for (uint32_t k = lane; k < 512u; k += 64) {
      const int sym = zh_dist_sym(zh_lp_index_dist(k));
      const uint32_t l = st->dist_len[sym];
      ws.distprice[k] = (uint8_t)((l ? l : 6u) + (uint32_t)zh_dist_xbits(sym));
   }
   zh_wave_sync();
   for (uint32_t e = lane; e < 256; e += 64) {
      const int idx = zh_len_idx(e + 3);
      ws.lencost[e] = (uint8_t)(ws.litprice[257 + idx] + zh_lenidx_xbits(idx));
   }
   zh_wave_sync();
   // price << 9 | (39 - k) of the lane's ten lengths k = 3 + 9 q + j
   uint32_t kc[ZH_LP_KPL];
#pragma unroll
   for (uint32_t j = 0; j < ZH_LP_KPL; j++) {
      const uint32_t e = ZH_LP_QSTRIDE * q + j;   // k - 3 <= 36
      kc[j] = ((uint32_t)ws.lencost[e] << 9) | (36u - e);
   }

   // ---- the group's pieces ---------------------------------------------------------------------------------------------
   uint32_t nlongp = 0, nshortp = 0;
   uint64_t parsed = 0;   // bit j: task g0 + j is parsed here
   for (uint32_t gt = g0; gt < g1; gt++) {
      // the task's range, and whether it has a barrier-free run of more than ZH_COOP_MIN positions: zh_list_huge has listed such a
      // task for zh_parse_chain / the segment workgroups — and left both here
      const uint2 ti = taskinfo[gt];
      if (ti.y >> 31) continue;
      const uint32_t t0 = ti.x, t1 = ti.y;
      const uint32_t np = zh_task_pieces(ws.bnd, bar, prev, t0, t1, lane);
      zh_wave_sync();
      if (nlongp + nshortp + np <= ZH_LP_MAXP) {
         const uint32_t lo = lane < np ? ws.bnd[lane] : 0u, hi = lane < np ? ws.bnd[lane + 1] : 0u;
         const bool is_l = hi - lo >= ZH_LP_LONG, is_s = hi > lo && !is_l;   // (empty pieces are dropped)
         const uint64_t ml = zh_ballot(is_l), ms = zh_ballot(is_s);
         if (is_l) {
            const uint32_t at = nlongp + (uint32_t)zh_popc64(ml & lt_mask);
            ws.plo[at] = lo;
            ws.phi[at] = hi;
         }
         if (is_s) {
            const uint32_t at = ZH_LP_MAXP - 1u - nshortp - (uint32_t)zh_popc64(ms & lt_mask);
            ws.plo[at] = lo;
            ws.phi[at] = hi;
         }
         nlongp += (uint32_t)zh_popc64(ml);
         nshortp += (uint32_t)zh_popc64(ms);
         parsed |= 1ull << (gt - g0);
      }
      zh_wave_sync();
   }
   const uint32_t npieces = nlongp + nshortp;
   ZH_LP_COUNT(3, 1);
   ZH_LP_COUNT(6, npieces);
   ZH_LP_COUNT(4, ZH_LP_CLOCK() - tic0);
   const uint64_t tic1 = ZH_LP_CLOCK();
#ifdef ZH_LP_PROFILE
   uint32_t prof_steps = 0, prof_quads = 0, prof_hi = 0;
   uint64_t prof_lap[4] = {0, 0, 0, 0}, prof_t = ZH_LP_CLOCK();
#define ZH_LP_LAP(k_) do { const uint64_t n_ = ZH_LP_CLOCK(); prof_lap[k_] += n_ - prof_t; prof_t = n_; } while (0)
   (void)prof_hi;
#endif

   // ---- the recurrences ----------------------------------------------------------------------------------------------------
   // The only thing a step needs from the step before it is cost[p + 1], for the literal: everything about the matches of p looks
   // at cost[p + 3] and beyond. So the steps are software-pipelined over three consecutive entries of the wave's stream — while
   // entry i takes its decision (stage C), entry i+1 scans its prefix minima and has its gathers from the scratch in flight (stage
   // B) and entry i+2 has its slots digested (stage A) — and no LDS round trip is waited for with nothing else to issue.
   uint32_t next = 0;              // wave-uniform: pieces handed out so far
   uint32_t sp = 0, sleft = 0, send = 0;   // the quad's fetch cursor: the next position to request is sp - 1, sleft are left of its piece, which ends at send
   uint32_t c1 = 0;                // cost[p + 1] of the position deciding next: the literal's continuation
   uint32_t cin1 = 0, cin2 = 0, cin3 = 0;   // cost << 9 decided one, two and three steps ago: the last one is what enters lane 0's window
   zh_lp_batch_t cur = {0, 0, 0, 0, false}, old = {0, 0, 0, 0, false}, b1 = {0, 0, 0, 0, false}, b2 = {0, 0, 0, 0, false};   // the batch entering stage A, the one before it, the next two
   uint4 a1 = make_uint4(0, 0, 0, 0), a2 = make_uint4(0, 0, 0, 0), h1 = make_uint4(0, 0, 0, 0);   // this lane's entry of batch b1 / b2: first plane; of b1: second plane
   uint32_t y1 = 0, y2 = 0;        // ... and its byte
   uint32_t f1[4] = {0, 0, 0, 0};  // ... of b1: the words that hold the costs behind its slots 0..3 where they are stored with length >= 40
   bool b1far = false;             // wave-uniform: batch b1 has such slots
   // the cost window of the entry in stage B, << 9: lane q's ten costs are w[wb .. wb + 9], wb = 3, 2, 1, 0 over the four entries of a
   // batch, then everything moves up by four
   uint32_t w[14];
#pragma unroll
   for (uint32_t k = 0; k < 14; k++) w[k] = 0;
   uint32_t *const pw = &ws.pmin[ZH_LP_QSTRIDE * q][ZH_LP_PCOL(ZH_LP_QSTRIDE * q, piece)];   // the rows of the prefix scratch this lane writes

   // the quad's next batch: a new piece when the current one is used up; requests its rows and bytes
#define ZH_LP_FETCH(bt_, a_, y_)                                                                                    \
   do {                                                                                                            \
      const uint64_t idle_ = zh_ballot(sleft == 0 && q == 0);                                                      \
      bool fresh_ = false;                                                                                         \
      if (idle_ && next < npieces) {                                                                               \
         const uint32_t i_ = next + (uint32_t)zh_popc64(idle_ & lt_quad);                                          \
         if (sleft == 0 && i_ < npieces) {                                                                         \
            const uint32_t at_ = i_ < nlongp ? i_ : ZH_LP_MAXP - nshortp + (i_ - nlongp);                          \
            sp = send = ws.phi[at_];                                                                               \
            sleft = sp - ws.plo[at_];                                                                              \
            fresh_ = true;                                                                                         \
         }                                                                                                         \
         next += (uint32_t)zh_popc64(idle_);                                                                       \
      }                                                                                                            \
      bt_.n = min(fresh_ ? 3u : 4u, sleft);                                                                        \
      bt_.off = fresh_ ? 4u - bt_.n : 0u;                                                                          \
      bt_.p0 = sp - 1u;                                                                                            \
      bt_.end = send;                                                                                              \
      bt_.fresh = fresh_;                                                                                          \
      a_ = make_uint4(0, 0, 0, 0);                                                                                 \
      y_ = 0;                                                                                                      \
      if (q - bt_.off < bt_.n) {                                                                                   \
         a_ = rows[bt_.p0 - (q - bt_.off) - prev];                                                                 \
         y_ = win[bt_.p0 - (q - bt_.off)];                                                                         \
      }                                                                                                            \
      sp -= bt_.n;                                                                                                 \
      sleft -= bt_.n;                                                                                              \
   } while (0)
   // The cost behind a slot stored with length >= 40 (e_ = length | offset << 16) of the position pos_ of a piece that ends at end_: 0 at
   // the piece's end (and wherever the sub-block's end clamps the length: the last piece ends there), else what the step of that
   // position left in global memory — at least 40 steps ago; read past this CU's vector cache, which may hold the line from before.
   // The load is of the aligned word (nothing has to look at its result before it is used, a batch later), ZH_LP_FAR_PICK takes the half.
#define ZH_LP_FAR_ON(e_, pos_, end_) (((e_) & 0xffffu) >= ZH_LEAVE_ALONE && (pos_) + ((e_) & 0xffffu) < (end_))
#define ZH_LP_FAR_AT(e_, pos_, end_) (ZH_LP_FAR_ON(e_, pos_, end_) ? (pos_) + ((e_) & 0xffffu) - prev : (pos_) - prev)
#define ZH_LP_FAR_PICK(word_, e_, pos_, end_) (ZH_LP_FAR_ON(e_, pos_, end_) ? ((word_) >> (16u * (((pos_) + ((e_) & 0xffffu) - prev) & 1u))) & 0xffffu : 0u)
   const uint32_t *const costs32 = (const uint32_t *)costs;
   // second plane of this lane's entry of batch b1 (slots 4..7 exist only behind a full first plane, zh_common.h), and the far costs
   // of its first plane
#define ZH_LP_FETCH_HI()                                                                                           \
   do {                                                                                                            \
      h1 = make_uint4(0, 0, 0, 0);                                                                                 \
      const bool on_ = q - b1.off < b1.n;                                                                          \
      const uint32_t pos_ = on_ ? b1.p0 - (q - b1.off) : wk.start;                                                 \
      if (on_ && (a1.w & 0xffffu) >= ZH_MIN_MATCH) h1 = rows_hi[pos_ - prev];                                      \
      b1far = zh_ballot(on_ && (a1.x & 0xffffu) >= ZH_LEAVE_ALONE) != 0;   /* rows are longest first */             \
      if (b1far) {                                                                                                 \
         f1[0] = zh_load_agent_u32(costs32 + (ZH_LP_FAR_AT(a1.x, pos_, b1.end) >> 1));                             \
         f1[1] = zh_load_agent_u32(costs32 + (ZH_LP_FAR_AT(a1.y, pos_, b1.end) >> 1));                             \
         f1[2] = zh_load_agent_u32(costs32 + (ZH_LP_FAR_AT(a1.z, pos_, b1.end) >> 1));                             \
         f1[3] = zh_load_agent_u32(costs32 + (ZH_LP_FAR_AT(a1.w, pos_, b1.end) >> 1));                             \
      }                                                                                                            \
   } while (0)

   // what an entry carries from stage A to stage B, and from stage B to stage C
   struct zh_lp_a_t {
      uint32_t e0, e1, mlen0, mlen1, dp0, dp1, lit, lr0, lr1, j;   // (lr: for a slot stored with length >= 40, the price of its clamped length + the cost behind it; j: entry | batch parity << 2)
      bool act, fresh;
   };
   struct zh_lp_b_t {
      uint32_t e0, e1, mlen0, mlen1, key0, key1, lit, j;
      bool act, fresh;
   };
   zh_lp_a_t ea;
   zh_lp_b_t eb;
   ea.e0 = ea.e1 = ea.mlen0 = ea.mlen1 = ea.dp0 = ea.dp1 = ea.lit = ea.lr0 = ea.lr1 = ea.j = 0;
   ea.act = ea.fresh = false;
   eb.e0 = eb.e1 = eb.mlen0 = eb.mlen1 = eb.lit = eb.j = 0;
   eb.key0 = eb.key1 = ZH_LP_NOKEY;
   eb.act = eb.fresh = false;

   ZH_LP_FETCH(b1, a1, y1);
   ZH_LP_FETCH(b2, a2, y2);
   ZH_LP_FETCH_HI();
   uint32_t drain = 0;   // the pipeline runs two entries behind the stream
   uint32_t it = 0;      // batches so far; the parse entries of a batch are buffered under its parity
   zh_lp_batch_t pend = {0, 0, 0, 0, false};   // the batch whose entries wait for the next one's: two batches' stores go out back to back
   bool cur_hi = false, old_hi = false;   // wave-uniform: some entry of the batch cur / old has a second plane
   for (;;) {
      // ---- batch b1 enters stage A: into LDS, where every lane of the quad finds its slot of every entry ---------------------------
      old = cur;
      cur = b1;
      if (!zh_ballot(cur.n != 0)) {   // (a quad without a batch has none later either: pieces are handed out in order)
         if (drain) break;
         drain = 1;   // one more round of empty entries lets the last real ones through stages B and C
      }
      {
         const uint32_t pos_ = q - cur.off < cur.n ? cur.p0 - (q - cur.off) : wk.start;
         uint4 fv = make_uint4(0, 0, 0, 0);
         if (b1far) {
            fv.x = ZH_LP_FAR_PICK(f1[0], a1.x, pos_, cur.end) | (ZH_LP_FAR_PICK(f1[1], a1.y, pos_, cur.end) << 16);
            fv.y = ZH_LP_FAR_PICK(f1[2], a1.z, pos_, cur.end) | (ZH_LP_FAR_PICK(f1[3], a1.w, pos_, cur.end) << 16);
            // (five and more slots of length >= 40 at one position: their costs are fetched here, with nothing to hide the round trip)
            if (zh_ballot((h1.x & 0xffffu) >= ZH_LEAVE_ALONE)) {
               const uint32_t g0 = zh_load_agent_u32(costs32 + (ZH_LP_FAR_AT(h1.x, pos_, cur.end) >> 1)), g1 = zh_load_agent_u32(costs32 + (ZH_LP_FAR_AT(h1.y, pos_, cur.end) >> 1));
               const uint32_t g2 = zh_load_agent_u32(costs32 + (ZH_LP_FAR_AT(h1.z, pos_, cur.end) >> 1)), g3 = zh_load_agent_u32(costs32 + (ZH_LP_FAR_AT(h1.w, pos_, cur.end) >> 1));
               fv.z = ZH_LP_FAR_PICK(g0, h1.x, pos_, cur.end) | (ZH_LP_FAR_PICK(g1, h1.y, pos_, cur.end) << 16);
               fv.w = ZH_LP_FAR_PICK(g2, h1.z, pos_, cur.end) | (ZH_LP_FAR_PICK(g3, h1.w, pos_, cur.end) << 16);
            }
            ws.stagef4[piece][q] = fv;
         }
         old_hi = cur_hi;
         cur_hi = zh_ballot((h1.x & 0xffffu) >= ZH_MIN_MATCH) != 0;
         *(uint4 *)&ws.stage0[piece][q][0] = a1;
         if (cur_hi) *(uint4 *)&ws.stage1[piece][q][0] = h1;
         ws.stageb[piece][q] = y1;
      }
      zh_lockstep_sync();
      b1 = b2;
      a1 = a2;
      y1 = y2;
      ZH_LP_FETCH_HI();
      ZH_LP_FETCH(b2, a2, y2);

#ifdef ZH_LP_PROFILE
      ZH_LP_LAP(0);
#endif
#pragma unroll
      for (uint32_t j = 0; j < 4; j++) {
         // ======== stage A, first half, of entry (cur, j): its slots and its byte ==========================================================
         zh_lp_a_t na;
         na.act = j - cur.off < cur.n;
         na.fresh = cur.fresh && j == cur.off;
         na.j = j | ((it & 1u) << 2);
         const uint32_t apos = na.act ? cur.p0 - (j - cur.off) : wk.start;
         const uint32_t aroom = sb_end - apos;   // end clamp (blockdeflate.c:283-284); a no-op away from the sub-block end
         na.e0 = ws.stage0[piece][j][q];
         na.e1 = cur_hi ? ws.stage1[piece][j][q] : 0u;
         const uint32_t abyte = ws.stageb[piece][j];
#ifdef ZH_LP_PROFILE
         prof_steps++;
         prof_quads += (uint32_t)zh_popc64(zh_ballot(na.act && q == 0));
#endif

#ifdef ZH_LP_PROFILE
         ZH_LP_LAP(3);
#endif
         // ======== stage C of the entry two before: literal first; a match must be strictly cheaper (:292,:307) =====================
         {
            const uint32_t bestkey = zh_quad_min(min(eb.key0, eb.key1));
            if (eb.fresh) c1 = 0;
            const uint32_t litc = c1 + eb.lit;
            const uint32_t mc = bestkey >> 9;
            const bool take = mc < litc;
            const uint32_t c = eb.act ? (take ? mc : litc) : 0u;   // (an empty step leaves a zero: the end of the piece that may follow)
            // the lane whose slot won writes the parse entry, lane 0 a literal and the cost
            const bool own0 = take && eb.key0 == bestkey, own1 = take && eb.key1 == bestkey;
            const uint32_t we = own1 ? eb.e1 : eb.e0, wm = own1 ? eb.mlen1 : eb.mlen0;
            const uint32_t pick = take ? ((we & 0xffffu) >= ZH_LEAVE_ALONE ? wm : 39u - (bestkey & 63u)) | (we & 0xffff0000u) : 0u;
            if (own0 || own1 || (!take && q == 0)) {
               ws.outp[eb.j >> 2][piece][eb.j & 3u] = pick;
               ws.outc[eb.j >> 2][piece][eb.j & 3u] = c;
            }
            c1 = c;
            cin3 = cin2;
            cin2 = cin1;
            cin1 = c << 9;
            zh_lockstep_sync();
            if (j == 1) {
               // The batch before this one is through. Its four parse entries are 16 bytes, its costs 8: stored batch by batch they left
               // the L2 as half-written 32-byte sectors, 19 bytes of HBM writes per position where 6 are stored (rocprofv3 WRITE_SIZE).
               // Every second batch, the entries of two batches go out back to back instead: neighbours in memory, merged in the L2.
               if (it & 1u) {
                  if (q - pend.off < pend.n) {
                     best[pend.p0 - (q - pend.off) - prev] = ws.outp[1][piece][q];
                     costs[pend.p0 - (q - pend.off) - prev] = (uint16_t)ws.outc[1][piece][q];
                  }
                  if (q - old.off < old.n) {
                     best[old.p0 - (q - old.off) - prev] = ws.outp[0][piece][q];
                     costs[old.p0 - (q - old.off) - prev] = (uint16_t)ws.outc[0][piece][q];
                  }
               }
               else
                  pend = old;
            }
         }

#ifdef ZH_LP_PROFILE
         ZH_LP_LAP(1);
#endif
         // ======== stage B of the entry before: its window, one position down; then the prefix minima over the lengths, P(L) = min over
         //          3 <= k <= L of (cost[pos + k] + price(k)) << 9 | (39 - k): the lane's own ten, then the minimum of the lanes below it ====
         zh_lp_b_t nb;
         {
            const uint32_t wb = 3u - j;
            {
               const uint32_t up = zh_quad_shr1(w[wb + 9u]);   // (the window before this step's was w[wb + 1 .. wb + 10])
               w[wb] = q == 0 ? cin3 : up;   // cost[pos + 3]: decided three entries before this one
            }
            uint32_t pl[ZH_LP_KPL];
            uint32_t P = ZH_LP_NOKEY;
#pragma unroll
            for (uint32_t k = 0; k < ZH_LP_KPL; k++) {
               P = min(P, w[wb + k] + kc[k]);
               pl[k] = P;
            }
            uint32_t x = zh_quad_shr1(P);
            x = q == 0 ? ZH_LP_NOKEY : x;
            x = min(x, zh_quad_shr1(x));
            x = min(x, zh_quad_lo2(x));
#pragma unroll
            for (uint32_t k = 0; k < ZH_LP_KPL; k++)
               if (k < ZH_LP_QSTRIDE || q == 3) pw[k * ZH_LP_C] = min(pl[k], x);   // (a lane's tenth length is its neighbour's first)
            zh_lockstep_sync();
            // a slot's best pair (slot, k <= its clamped length) is P(that length) + its distance price; a slot stored with length
            // >= 40 is tried at its full (clamped) length only (blockdeflate.c:286-297)
            const uint32_t len0 = ea.e0 & 0xffffu, len1 = ea.e1 & 0xffffu;
            const bool long0 = len0 >= ZH_LEAVE_ALONE, long1 = len1 >= ZH_LEAVE_ALONE;
            const bool short0 = len0 >= ZH_MIN_MATCH && !long0 && ea.mlen0 >= ZH_MIN_MATCH, short1 = len1 >= ZH_MIN_MATCH && !long1 && ea.mlen1 >= ZH_MIN_MATCH;
            const uint32_t r0 = short0 ? ea.mlen0 - ZH_MIN_MATCH : 0u, r1 = short1 ? ea.mlen1 - ZH_MIN_MATCH : 0u;
            const uint32_t pk0 = ws.pmin[r0][ZH_LP_PCOL(r0, piece)];
            const uint32_t lk0 = long0 ? ((ea.lr0 + ea.dp0) << 9) | (q << 6) : ZH_LP_NOKEY;
            nb.key0 = short0 ? pk0 + ((ea.dp0 << 9) | (q << 6)) : lk0;
            nb.key1 = ZH_LP_NOKEY;
            if (j == 0 ? old_hi : cur_hi) {   // (the entry in this stage is the last one of the batch before when j is 0)
               const uint32_t pk1 = ws.pmin[r1][ZH_LP_PCOL(r1, piece)];
               const uint32_t lk1 = long1 ? ((ea.lr1 + ea.dp1) << 9) | ((4u + q) << 6) : ZH_LP_NOKEY;
               nb.key1 = short1 ? pk1 + ((ea.dp1 << 9) | ((4u + q) << 6)) : lk1;
            }
            nb.e0 = ea.e0;
            nb.e1 = ea.e1;
            nb.mlen0 = ea.mlen0;
            nb.mlen1 = ea.mlen1;
            nb.lit = ea.lit;
            nb.j = ea.j;
            nb.act = ea.act;
            nb.fresh = ea.fresh;
         }

#ifdef ZH_LP_PROFILE
         ZH_LP_LAP(2);
#endif
         // ======== stage A, second half: the slots' distance prices, the literal's price; for a slot stored with length >= 40 the
         //          price of its clamped length and the cost behind it ========================================================================
         {
            // (an entry without a position holds zeros in the stage: its lane fetched nothing, ZH_LP_FETCH / ZH_LP_FETCH_HI)
            const uint32_t len0 = na.e0 & 0xffffu, len1 = na.e1 & 0xffffu;
            na.mlen0 = min(len0, aroom);
            na.mlen1 = min(len1, aroom);
            na.dp0 = ws.distprice[len0 >= ZH_MIN_MATCH ? zh_lp_dist_index((na.e0 >> 16) - 1u) : 0u];   // rows are zero padded: an empty slot ends the row
            na.dp1 = 0;
            if (cur_hi) na.dp1 = ws.distprice[len1 >= ZH_MIN_MATCH ? zh_lp_dist_index((na.e1 >> 16) - 1u) : 0u];
            na.lit = ws.litprice[abyte & 0xffu];
            const bool long0 = len0 >= ZH_LEAVE_ALONE;
            na.lr0 = na.lr1 = 0;
            if (zh_ballot(long0)) {
               // (rows are longest first: a long slot 4..7 sits behind four long slots 0..3)
               uint32_t enc0 = na.mlen0 - ZH_MIN_MATCH, enc1 = na.mlen1 - ZH_MIN_MATCH;   // wraps below 3, then saturates (:289, :216-219)
               if (enc0 > 255u) enc0 = 255u;
               if (enc1 > 255u) enc1 = 255u;
               na.lr0 = (uint32_t)ws.lencost[enc0] + (uint32_t)ws.stagef[piece][j][q];
               na.lr1 = (uint32_t)ws.lencost[enc1] + (uint32_t)ws.stagef[piece][j][4u + q];
            }
         }
         eb = nb;
         ea = na;
      }
      // the window moves up by four: the next batch's four entries find it at w[4 .. 13] again
#pragma unroll
      for (uint32_t k = 13; k >= 4; k--) w[k] = w[k - 4];
      it++;
   }
   if (it & 1u) {   // (the last round was an even one: its `old` batch is still waiting)
      zh_lockstep_sync();
      if (q - pend.off < pend.n) {
         best[pend.p0 - (q - pend.off) - prev] = ws.outp[1][piece][q];
         costs[pend.p0 - (q - pend.off) - prev] = (uint16_t)ws.outc[1][piece][q];
      }
   }
#undef ZH_LP_FETCH
#undef ZH_LP_FETCH_HI
#undef ZH_LP_FAR_ON
#undef ZH_LP_FAR_AT
#undef ZH_LP_FAR_PICK
#ifdef ZH_LP_PROFILE
   ZH_LP_COUNT(0, prof_steps);
   ZH_LP_COUNT(1, prof_quads);
   ZH_LP_COUNT(7, prof_hi);
   for (int k = 0; k < 4; k++) ZH_LP_COUNT(8 + k, prof_lap[k]);
#undef ZH_LP_LAP
#endif
   ZH_LP_COUNT(2, ZH_LP_CLOCK() - tic1);
   const uint64_t tic2 = ZH_LP_CLOCK();

   // ---- histogram of the group's parse; the per-sub-block sum is taken by zh_sb_build ------------------------------------
   if (sb_dynamic) {
      __threadfence_block();
      zh_wave_sync();
      for (uint32_t k = lane; k < ZH_NSYM; k += 64) ws.hist[k] = 0;
      zh_wave_sync();
      ZH_LP_COUNT(12, ZH_LP_CLOCK() - tic2);
      const uint64_t tic3 = ZH_LP_CLOCK();
      for (uint32_t gt = g0; gt < g1; gt++) {
         if (!((parsed >> (gt - g0)) & 1ull)) continue;
         const uint2 ti = taskinfo[gt];
         zh_walk_histogram_wave(ws.hist, win, prev, ti.x, ti.y, best);
      }
      zh_wave_sync();
      ZH_LP_COUNT(13, ZH_LP_CLOCK() - tic3);
      (void)tic3;
      bool first = true;
      for (uint32_t gt = g0; gt < g1; gt++) {
         if (!((parsed >> (gt - g0)) & 1ull)) continue;
         uint32_t *hp = hist_part + (uint64_t)gt * ZH_NSYM;
         for (uint32_t k = lane; k < ZH_NSYM; k += 64) hp[k] = first ? ws.hist[k] : 0u;
         first = false;
      }
   }
   ZH_LP_COUNT(5, ZH_LP_CLOCK() - tic2);
}

// The cut tasks' segments, when a run has many (zh_parse.h, "speculative segments"; zh_segments_are_wide), are parsed by the FIRST seg_grid workgroups of
// this kernel's grid — one wave per entry of segwaves, four segments each, the wave that finishes a task checks it — next to the workgroups behind them,
// which parse the task list on the quads: two kinds of work in one launch. (Rounds 2-4 launched zh_parse_segments as a kernel of its own on a third
// stream per run. Since the host no longer knows whether a run has such segments — round 5: the counts stay on the device — that launch would be made
// for every run and pass, and a third active stream per run turned out to be one too many next to other contexts on the device: three jobs in flight fell
// from 2.7 GB/s to 0.36, the runtime's eight hardware queues stalling on each other's event waits.) A run whose segments go to zh_parse_chain, or that
// has none, sees those workgroups leave at once; entries beyond the grid are taken in strides.
struct zh_seg_args_t {
   uint4 *segtasks;
   const uint2 *segwaves;
   int16_t *vecs;
   uint32_t *demote_list;   // the run's fourth chain list
   uint32_t demote_min;     // a task with this many failed cuts in a pass is a whole chain from the next pass on; 0: never
   uint32_t seg_wide_min;   // fewer segments in the run: zh_parse_chain takes them
   uint32_t seg_grid;       // workgroups of this launch that take segment entries (0: none — files mode)
};

// Behind them, single-wave workgroups take groups of tasks_per_wave consecutive tasks from *ticket until the run's task list — whose length only the device
// knows (cnt[ZH_CNT_TASKS]) — is used up. The host gives a wave up to ZH_LP_TASKS tasks — a pool of pieces large enough to keep its sixteen quads
// busy — but no more than it takes to give every wave slot of the chip a wave: a small batch (one 40 KB input: 20 tasks) is a matter of latency,
// not of lane utilisation. Next to chains (zh_parse_chain, the segment workgroups: the run's counters say whether it has any) only the first
// `bounded` workgroups stay: a grid that keeps every wave slot, register and LDS granule of the chip taken would make the four-wave workgroup that
// carries the longest chain of the batch wait for room until the grid has drained (measured, tools/probes/chain2_probe.hip: a 3.6 ms chain next to
// such a grid ended after 25 ms).
// (118 registers. Capped at 96 — a wave would then fit on a SIMD next to the four 104-register waves of another run's matchfinder
// workgroup — the compiler spills 17 of them and the kernel takes 1.5 times as long: measured, not kept.)
__global__ void __launch_bounds__(64)
zh_parse_lanes(const uint8_t *__restrict__ data, const zh_block_t *__restrict__ blocks, const zh_match_t *__restrict__ match, uint64_t match_stride,
               const uint64_t *__restrict__ bars, uint64_t bar_stride, const zh_work_t *__restrict__ work, const uint2 *__restrict__ taskmap,
               uint32_t *cnt, const zh_sbstate_t *__restrict__ states, uint32_t *best_all, uint64_t best_stride, uint16_t *cost_all,
               uint32_t *hist_part, int pass, uint32_t *ticket, const uint2 *__restrict__ taskinfo, uint32_t tasks_per_wave /* 1 .. ZH_LP_TASKS */,
               uint32_t bounded /* quad workgroups that stay when the run has chains */, zh_seg_args_t sg) {
   __shared__ union {
      zh_lp_ws_t ws;
      zh_parse_ws_t seg_ws;
   } sh;
   if (zh_run_is_void(cnt)) return;   // (no chain kernels were launched, and the run has chains: the host runs the batch again)
   if (blockIdx.x < sg.seg_grid) {
      // ---- a workgroup of segment entries
      if (!zh_segments_are_wide(cnt, sg.seg_wide_min)) return;
      const uint32_t nwaves = cnt[ZH_CNT_SEGWAVES];
      for (uint32_t w = blockIdx.x; w < nwaves; w += sg.seg_grid) {
         zh_sync();   // the entry before this one is done with the workspace
         const uint2 sw = sg.segwaves[w];
         const uint4 stask = sg.segtasks[sw.x];
         if (stask.y & ZH_CUT_DEMOTED) continue;   // (set by the task's checker at the end of an earlier pass: every wave of the task sees it or none)
         zh_parse_one_task<true>(sh.seg_ws, stask.x, data, blocks, match, match_stride, bars, bar_stride, work, taskmap, states, best_all, best_stride, hist_part, pass, stask.y, sw.y,
                                 stask.z, sg.vecs, &sg.segtasks[sw.x].w, (uint32_t *)cnt, sg.demote_list, sg.demote_min);
      }
      return;
   }
   zh_lp_ws_t &ws = sh.ws;
#ifdef ZH_LP_TRACE
   const uint64_t trace_born = zh_wall_clock();
#endif
   const uint32_t ntasks = cnt[ZH_CNT_TASKS];
   if (blockIdx.x - sg.seg_grid >= bounded && (cnt[ZH_CNT_VLONG] | cnt[ZH_CNT_LONG] | cnt[ZH_CNT_SHORT] | cnt[ZH_CNT_SEGTASKS]) != 0u) return;
   for (;;) {
      uint32_t w = blockIdx.x - sg.seg_grid;
      if (ticket) {
         if (zh_lane() == 0) w = atomicAdd(ticket, 1u);
         w = zh_readfirstlane(w);
      }
      const uint32_t g0 = w * tasks_per_wave;
      if (g0 >= ntasks) return;
      const uint32_t g1 = min(ntasks, g0 + tasks_per_wave);
#ifdef ZH_LP_TRACE
      const uint64_t trace_t0 = zh_wall_clock();
#endif
      // the wave's tasks, sub-block by sub-block
      for (uint32_t g = g0; g < g1;) {
         const uint32_t sb = taskmap[g].x;
         uint32_t ge = g + 1;
         while (ge < g1 && taskmap[ge].x == sb) ge++;
         zh_lp_group(ws, g, ge, data, blocks, match, match_stride, bars, bar_stride, work, taskmap, states, best_all, best_stride, cost_all, hist_part, pass, taskinfo);
         g = ge;
      }
#ifdef ZH_LP_TRACE
      if (zh_lane() == 0 && w < ZH_LP_TRACE_SLOTS) {
         uint64_t *tr = zh_lp_trace + ((uint64_t)pass * ZH_LP_TRACE_SLOTS + w) * 4u;
         tr[0] = trace_born;
         tr[1] = trace_t0;
         tr[2] = zh_wall_clock();
         tr[3] = ((uint64_t)blockIdx.x << 32) | (uint64_t)__builtin_amdgcn_s_getreg((4 /* HW_REG_HW_ID */) | (0 << 6) | (31 << 11));
      }
#endif
      if (!ticket) return;
   }
}
