// zh_matchfinder.h — stage 1 of the hot path on MI355X: the match rows of a max-block.
//
// Replaces zultra_build_suffix_array + zultra_skip_matches + zultra_find_all_matches
// (reference src/matchfinder.c:49-286, called at src/libzultra.c:287-293).
//
// The reference's rows have a closed form: for block position i, scan the earlier window
// positions p from nearest to farthest; whenever the match length L(p) = min(LCP(i,p), 258, windowEnd-i)
// is >= 3 and strictly longer than everything seen so far, p is a row entry; entries farther than 32768 are
// dropped; the 8 longest survive, longest first. No suffix array is needed for that. Three facts bound the work:
//   * the nearest entry is the previous occurrence of the trigram at i; it has length exactly 3 unless its 4th byte
//     matches too. Likewise the previous occurrence of the 4-gram gives the entry of length 4 (unless its 5th byte
//     matches), the previous occurrence of the 5-gram the entry of length 5;
//   * every other entry is 6 or longer, i.e. an earlier occurrence of the 6-gram at i: only those need a walk, nearest
//     first, over the 6-gram class (measured against walking 4-gram classes: 2.3x fewer steps on text, 1.5x on code);
//   * a position with six or more bytes of a byte run ahead (indentation, padding) has its frontier in closed form from a
//     table of the window's runs (see "byte runs" below): the class "cccccc" is never walked.
// The orders are built with exact stable counting passes, digits = bytes:
//   zh_mf_group    persistent workgroups, one segment (zh_common.h) at a time.
//                  trigram order (3 passes)      -> prev.x = previous occurrence of the same trigram;
//                  + one pass on byte 3          -> 4-gram classes, contiguous and ascending in position -> prev4
//                  + one pass on byte 4          -> 5-gram classes -> prev5
//                  + one pass on byte 5          -> 6-gram order with class heads marked: candidate lists are contiguous
//                                                   memory, no pointer chasing (hash chains would serialise on latency).
//                  Every pass gathers its digit from the window copy in LDS.
//   zh_mf_frontier waves pull 64-entry chunks of the 6-gram order from a counter in HBM (any number of workgroups can
//                  serve one segment: finished workgroups help unfinished segments). Neighbouring entries belong to the
//                  same class and have near-equal candidate counts, so the lanes of a wave stay busy together. The scan is
//                  wave-synchronous: lane l's k-th candidate is lane l-1's (k-1)-th, so the candidate stream is passed up
//                  the lanes with one DPP wave shift per step and refilled from one coalesced 64-entry load per 64
//                  steps; the window sits in LDS.
//
// HBM traffic per segment: window read twice linearly, 6 x 2 x 4 B per window position for the sort ping-pong,
// 8 B per position for the prev records, 32 B per block position for the rows.
#pragma once
#include <zh_platform.h>
#include "zh_common.h"

#ifndef ZH_MF_THREADS
#define ZH_MF_THREADS 1024
#endif
#define ZH_MF_WAVES (ZH_MF_THREADS / 64)
#ifndef ZH_MF_UNROLL
#define ZH_MF_UNROLL 4u              // 64-element steps of a sort pass whose loads are issued together
#endif
#ifndef ZH_MF_STEP
#define ZH_MF_STEP 4u                // candidates a lane looks at per step of the class walk (a divisor of 64)
#endif
#define ZH_MF_HEAD 0x80000000u       // sorted entry: first position of its 6-gram class
#define ZH_MF_POS_MASK 0x7fffffffu
#define ZH_MF_SENTINEL 0xffffffffu   // "no entry": marked, and farther than any legal distance

#define ZH_MF_NONE 0xffffffffu       // prev.x: no earlier occurrence of the trigram
#define ZH_MF_GROUP_LDS 163840u        // dynamic LDS bytes of zh_mf_group: all of the CU's. Window, two counter tables, small variables, 256 cursors for the
                                      // passes through HBM ((ZH_MF_LDS_WINDOW / 4 + 4 + 2 * ZH_MF_WAVES * 256 + ZH_MF_WAVES + 1 + 256) * 4 bytes); the rest for the chunks of zh_mf_group_lds.h
#define ZH_MF_FRONTIER_LDS ((ZH_MF_LDS_WINDOW / 4 + 4 + 8 * ZH_MF_THREADS + 1) * 4)              // ... of zh_mf_frontier
#ifndef ZH_MF_EXT_TO
#define ZH_MF_EXT_TO 80u             // zh_mf_length_past16: the lanes extend their own matches up to here (32 + a multiple of 16), the whole wave beyond
#endif
#define ZH_MF_HELP_WINDOW 4096u       // zh_mf_frontier: a workgroup out of tickets looks at the last 4096 segments for one to help
#ifndef ZH_MF_HELP_MIN
#define ZH_MF_HELP_MIN 32u           // ... and joins only for at least this many 64-entry chunks per workgroup
#endif

// unaligned little-endian loads from global memory
__device__ __forceinline__ uint32_t zh_ld24(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16); }
__device__ __forceinline__ uint32_t zh_ld32(const uint8_t *p) { return zh_ld24(p) | ((uint32_t)p[3] << 24); }

// LDS_WIN: the whole window (<= ZH_MF_LDS_WINDOW bytes: 64 KiB max-blocks + 32 KiB history) is staged in LDS once
// per workgroup with coalesced dword loads, so the scattered byte reads hit the 160 KiB LDS instead of L1/L2.
#define ZH_MF_LDS_WINDOW 98304
__device__ inline void zh_stage_window(uint32_t *lwin32, const uint8_t *gwin, uint32_t W) {
   uint8_t *lw = (uint8_t *)lwin32;
   if ((((uintptr_t)gwin) & 3u) == 0) {
      const uint32_t *g32 = (const uint32_t *)gwin;
      const uint32_t nw = W >> 2;
      for (uint32_t k = threadIdx.x; k < nw; k += ZH_MF_THREADS) lwin32[k] = g32[k];
      for (uint32_t k = (nw << 2) + threadIdx.x; k < W; k += ZH_MF_THREADS) lw[k] = gwin[k];
   }
   else {
      for (uint32_t k = threadIdx.x; k < W; k += ZH_MF_THREADS) lw[k] = gwin[k];
   }
   __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------
// zh_mf_group
// ---------------------------------------------------------------------------------------------------------
// One stable counting pass over M elements: wave w owns the contiguous slice [w*seg, (w+1)*seg), so element order
// within a digit is preserved. hist = ZH_MF_WAVES x 256 counters in LDS.
// MODE 0/1/2: the trigram's bytes 2/1/0 are the digits (exact order, three passes, elements are positions);
// MODE 5/6/7: byte 3/4/5 of the string at the position: a stable pass over the k-gram order gives the (k+1)-gram classes,
// contiguous and ascending in position (the classes come out ordered by their last byte first, which nobody minds).
// Mode 0 reads the identity permutation and the window linearly (win); the others gather through gwin.
// HAVE: hist already holds this pass's per-(wave, digit) counts — the previous pass counted them while it scattered
// (its outputs are this pass's inputs), which saves this pass one read of its input and one digit gather per element.
// NEXT >= 0: count the digits of pass mode NEXT into hist_next the same way (M_next = number of elements this pass writes).
__device__ __forceinline__ uint32_t zh_mf_slice(uint32_t M) { return (((M + ZH_MF_WAVES - 1) / ZH_MF_WAVES) + 63) & ~63u; }

// MODE 4: byte 2 likewise (a class of the bigram order that is refined through HBM, zh_mf_group_lds.h). MODE 10/11: the bigram order —
// 10 reads the identity permutation, digit = byte 1, and writes position | byte 0 << 24; 11 takes its digit from there and writes the position.
// MODE 8/9: the elements are indices into the run table (aux_rs = run starts, aux_rl = run lengths): digit = the byte that follows the
// run (0 where the run reaches the window end) / the run's byte. MODE 8 with src == nullptr reads the identity permutation.
// PAY: the elements carry a payload through the pass — 1: one word (psrc -> pdst), 2: two words (psrc, qsrc -> the pairs pdst2).
// PREV = K (3, 4, 5): the pass also takes, for every element of its INPUT order — the K-gram order, classes contiguous and ascending in
// position — the distance to the nearest earlier position sharing K bytes: the element before it, when that one is in the same
// class. Recorded as distance - 1 in 16 bits (0xffff: none within ZH_MAX_DIST) and sent along as payload: K = 3: d3 | none << 16 ->
// pdst; K = 4: d3 (read from psrc) | d4 << 16 -> pdst; K = 5: (psrc value, d5) -> pdst2. So the three distances of a position arrive
// next to its entry of the 6-gram order, where zh_mf_frontier reads them with the entries, coalesced. (Round 1 scattered them into
// a table indexed by position: 8.1 GB of HBM writes per 50 MB. Round 2 took them in a pass of their own over each order, zh_mf_prev_level:
// one more read of the order and one more write of the payload per level — 0.7 of this kernel's 4 ms per 50 MB and a quarter of its
// HBM traffic.) The last five window positions drop out of the orders before they reach the 6-gram order: theirs go to tail[pos]
// (the frontier's position-indexed table: entries from W - 5 on are free, the 6-gram order has at most W - 5 entries).
// Run-interior positions: the byte before and the six from the position on are all the same. Such a position never enters the orders (round 4,
// zh_mf_group_lds.h): the nearest earlier occurrence of its 3-, 4- and 5-gram is the position before it, its 6-gram class "cccccc" is
// never walked (zh_mf_frontier takes the frontier of such positions from the run table) — and in indented source code or zero-padded
// binaries they are a fifth of all positions, one bigram class that fits no chunk. g = the window (LDS), readable up to p + 5 < W.
__device__ __forceinline__ bool zh_mf_run_interior(const uint8_t *g, uint32_t p, uint32_t W) {
   if (p == 0 || p + 5u >= W) return false;
   // (the four bytes from p on first, with two aligned words: on text hardly any position gets past them)
   const uint32_t *g32 = (const uint32_t *)g;
   const uint32_t w = zh_funnel(g32[(p >> 2) + 1], g32[p >> 2], p * 8u);
   const uint32_t c = w & 0xffu;
   if (w != c * 0x01010101u) return false;
   return g[p - 1] == c && g[p + 4] == c && g[p + 5] == c;
}
// ... and the other side of it: a position whose K-gram is K times the byte before it has that position as its nearest earlier occurrence,
// in the orders or not (distance 1, recorded as 0)
template <int K>
__device__ __forceinline__ bool zh_mf_prev_is_neighbour(const uint8_t *g, uint32_t p) {
   if (p == 0) return false;
   const uint32_t c = g[p - 1];
   if (g[p] != c) return false;   // (nearly always)
   bool same = g[p + 1] == c && g[p + 2] == c;
   if (K >= 4) same = same && g[p + 3] == c;
   if (K >= 5) same = same && g[p + 4] == c;
   return same;
}

template <int MODE, bool HAVE = false, int NEXT = -1, int PAY = 0, int PREV = 0>
__device__ inline void zh_mf_sort_pass(const uint8_t *win, const uint8_t *gwin, uint32_t M, const uint32_t *src, uint32_t *dst, uint32_t *hist,
                                       uint32_t *wave_tot, uint32_t W = 0, uint32_t *hist_next = nullptr, uint32_t M_next = 0,
                                       const uint32_t *aux_rs = nullptr, const uint32_t *aux_rl = nullptr, const uint32_t *psrc = nullptr,
                                       uint32_t *pdst = nullptr, const uint32_t *qsrc = nullptr, uint2 *pdst2 = nullptr, uint2 *tail = nullptr) {
   const uint32_t tid = threadIdx.x;
   const uint32_t lane = tid & 63, wave = tid >> 6;
   const uint32_t seg = zh_mf_slice(M);
   const uint32_t lo = wave * seg;
   const uint32_t hi = min(M, lo + seg);
   const uint64_t lt_mask = (1ull << lane) - 1;
   (void)M_next;

   if (!HAVE)
      for (uint32_t k = tid; k < ZH_MF_WAVES * 256; k += ZH_MF_THREADS) hist[k] = 0;
   if (NEXT >= 0)
      for (uint32_t k = tid; k < ZH_MF_WAVES * 256; k += ZH_MF_THREADS) hist_next[k] = 0;
   __syncthreads();

   // element and digit of slice index idx
#define ZH_MF_FETCH(idx, e, d)                                                                      \
   do {                                                                                             \
      if (MODE == 0) {                                                                              \
         e = (idx);                                                                                 \
         d = win[(idx) + 2];                                                                        \
      }                                                                                             \
      else if (MODE == 1 || MODE == 2) {                                                            \
         e = src[idx];                                                                              \
         d = gwin[e + (MODE == 1 ? 1 : 0)];                                                         \
      }                                                                                             \
      else if (MODE == 8) {                                                                         \
         e = src ? src[idx] : (idx);                                                                \
         const uint32_t after_ = aux_rs[e] + aux_rl[e];                                             \
         d = after_ < W ? (uint32_t)gwin[after_] : 0u;                                              \
      }                                                                                             \
      else if (MODE == 9) {                                                                         \
         e = src[idx];                                                                              \
         d = gwin[aux_rs[e]];                                                                       \
      }                                                                                             \
      else if (MODE == 10) {                                                                        \
         e = (idx) | ((uint32_t)gwin[idx] << 24);                                                   \
         d = zh_mf_run_interior(gwin, (idx), W) ? 0xffffffffu : (uint32_t)gwin[(idx) + 1];          \
      }                                                                                             \
      else if (MODE == 11) {                                                                        \
         e = src[idx];                                                                              \
         d = e >> 24;                                                                               \
      }                                                                                             \
      else { /* MODE 4/5/6/7: byte 3/4/5 of the string at e; strings that end before it are dropped */ \
         e = src[idx];                                                                              \
         d = (e + (MODE - 2) < W) ? (uint32_t)gwin[e + (MODE - 2)] : 0xffffffffu;                   \
      }                                                                                             \
   } while (0)

   // the same in two steps: the element (a load from HBM/L2 — issued a tile ahead), then its digit (LDS, or small tables)
#define ZH_MF_FETCH_E(idx, e)                                             \
   do {                                                                   \
      if (MODE == 0)                                                      \
         e = (idx);                                                       \
      else if (MODE == 10)                                                \
         e = (idx) | ((uint32_t)gwin[idx] << 24);                         \
      else if (MODE == 8)                                                 \
         e = src ? src[idx] : (idx);                                      \
      else                                                                \
         e = src[idx];                                                    \
   } while (0)
#define ZH_MF_FETCH_D(idx, e, d)                                                                    \
   do {                                                                                             \
      if (MODE == 0)                                                                                \
         d = win[(idx) + 2];                                                                        \
      else if (MODE == 1 || MODE == 2)                                                              \
         d = gwin[e + (MODE == 1 ? 1 : 0)];                                                         \
      else if (MODE == 8) {                                                                         \
         const uint32_t after_ = aux_rs[e] + aux_rl[e];                                             \
         d = after_ < W ? (uint32_t)gwin[after_] : 0u;                                              \
      }                                                                                             \
      else if (MODE == 9)                                                                           \
         d = gwin[aux_rs[e]];                                                                       \
      else if (MODE == 10)                                                                          \
         d = zh_mf_run_interior(gwin, (idx), W) ? 0xffffffffu : (uint32_t)gwin[(idx) + 1];          \
      else if (MODE == 11)                                                                          \
         d = e >> 24;                                                                               \
      else                                                                                          \
         d = (e + (MODE - 2) < W) ? (uint32_t)gwin[e + (MODE - 2)] : 0xffffffffu;                   \
   } while (0)

   // ---- digit totals: counted by the pass before (HAVE: per producing wave, summed here) or by a pass over the input -----------
   for (uint32_t base = lo; base < hi && !HAVE; base += 64 * ZH_MF_UNROLL) {
      uint32_t e[ZH_MF_UNROLL], d[ZH_MF_UNROLL];
#pragma unroll
      for (uint32_t u = 0; u < ZH_MF_UNROLL; u++) {
         const uint32_t idx = base + u * 64 + lane;
         e[u] = 0;
         d[u] = 0xffffffffu;
         if (idx < hi) ZH_MF_FETCH(idx, e[u], d[u]);
      }
#pragma unroll
      for (uint32_t u = 0; u < ZH_MF_UNROLL; u++)
         if (d[u] != 0xffffffffu) atomicAdd(&hist[wave * 256 + d[u]], 1u);
   }
   __syncthreads();
   // cursor[d] = where digit d's run of the output starts (exclusive scan of the totals; 256 values: one wave, four per lane)
   uint32_t *cursor = wave_tot + ZH_MF_WAVES + 1 + ZH_MF_WAVES * 256;   // behind the second counter table (ZH_MF_GROUP_LDS)
   if (tid < 256) {
      uint32_t t = 0;
      for (uint32_t w2 = 0; w2 < ZH_MF_WAVES; w2++) t += hist[w2 * 256 + tid];
      cursor[tid] = t;
   }
   __syncthreads();
   if (wave == 0) {
      uint32_t v[4], t = 0;
      for (int q = 0; q < 4; q++) {
         v[q] = cursor[lane * 4 + (uint32_t)q];
         t += v[q];
      }
      uint32_t ex = zh_wave_excl_sum(t);
      for (int q = 0; q < 4; q++) {
         cursor[lane * 4 + (uint32_t)q] = ex;
         ex += v[q];
      }
   }
   __syncthreads();

   // ---- stable scatter, a tile of ZH_MF_WAVES x 64 x ZH_MF_UNROLL consecutive elements at a time: wave w takes the w-th stretch of the
   // tile, counts its digits, the counts of the waves before it (and the digit's cursor) give where its elements go. One cursor per
   // digit and workgroup: 256 lines of the output are open at a time, and every tile appends to them. (Round 2 gave every wave a
   // contiguous sixteenth of the input and cursors of its own: 4096 open lines per workgroup, 32 workgroups to an XCD's 4 MB of L2 —
   // lines left the L2 with a few 4-byte entries in them and the kernel wrote 6 GB per launch for 3.3 GB of stores.)
   const uint32_t tile = ZH_MF_WAVES * 64u * ZH_MF_UNROLL;
   // the elements (and payloads) of the next tile are requested before this tile's barriers: their latency hides behind them
   uint32_t en[ZH_MF_UNROLL], pn[ZH_MF_UNROLL], qn[ZH_MF_UNROLL], ebn = ZH_MF_NONE;
#define ZH_MF_REQUEST(tile0_)                                                                           \
   do {                                                                                                 \
      const uint32_t b4_ = (tile0_) + wave * 64u * ZH_MF_UNROLL;                                        \
      _Pragma("unroll") for (uint32_t u = 0; u < ZH_MF_UNROLL; u++) {                                   \
         const uint32_t idx = b4_ + u * 64 + lane;                                                      \
         en[u] = pn[u] = qn[u] = 0;                                                                     \
         if (idx < M) {                                                                                 \
            ZH_MF_FETCH_E(idx, en[u]);                                                                  \
            if (PAY >= 1 && PREV != 3) pn[u] = psrc[idx];                                               \
            if (PAY >= 2 && !PREV) qn[u] = qsrc[idx];                                                   \
         }                                                                                              \
      }                                                                                                 \
      if (PREV) ebn = (b4_ > 0 && b4_ < M) ? src[b4_ - 1] : ZH_MF_NONE;                                  \
   } while (0)
   ZH_MF_REQUEST(0u);
   for (uint32_t tile0 = 0; tile0 < M; tile0 += tile) {
      const uint32_t base4 = tile0 + wave * 64u * ZH_MF_UNROLL;
      uint32_t e4[ZH_MF_UNROLL], d4[ZH_MF_UNROLL], p4[ZH_MF_UNROLL], q4[ZH_MF_UNROLL];
      for (uint32_t k = lane; k < 256; k += 64) hist[wave * 256 + k] = 0;
      zh_lockstep_point();
#pragma unroll
      for (uint32_t u = 0; u < ZH_MF_UNROLL; u++) {
         const uint32_t idx = base4 + u * 64 + lane;
         e4[u] = en[u];
         p4[u] = pn[u];
         q4[u] = qn[u];
         d4[u] = 0xffffffffu;
         if (idx < M) ZH_MF_FETCH_D(idx, e4[u], d4[u]);
      }
      if (PREV) {
         uint32_t e_before = ebn;   // wave-uniform: the element before this step's first one
#pragma unroll
         for (uint32_t u = 0; u < ZH_MF_UNROLL; u++) {
            const uint32_t idx = base4 + u * 64 + lane;
            const uint32_t pos = e4[u];
            const uint32_t q = zh_wave_shr1(pos, e_before);   // the element before this one in the order
            e_before = zh_readlane(pos, 63);
            if (idx < M) {
               bool same = q != ZH_MF_NONE;
               if (same) {
                  if (PREV == 3)
                     same = zh_ld24(gwin + q) == zh_ld24(gwin + pos);
                  else
                     same = zh_ld32(gwin + q) == zh_ld32(gwin + pos) && (PREV == 4 || gwin[q + 4] == gwin[pos + 4]);
               }
               const uint32_t dist = pos - q;
               uint32_t dd = (same && dist <= ZH_MAX_DIST) ? dist - 1u : 0xffffu;
               if (zh_mf_prev_is_neighbour<PREV>(gwin, pos)) dd = 0;   // (the position before it: in the order or not, zh_mf_run_interior)
               if (PREV == 3) p4[u] = dd | 0xffff0000u;
               if (PREV == 4) p4[u] = (p4[u] & 0xffffu) | (dd << 16);
               if (PREV == 5) q4[u] = dd;
               if (pos + 5u >= W) {
                  if (PREV == 3)
                     tail[pos] = make_uint2(p4[u], 0xffffu);
                  else if (PREV == 4)
                     tail[pos].x = p4[u];
                  else
                     tail[pos].y = dd;
               }
            }
         }
      }
      // a lane's place among the wave's elements of its digit in this tile: the lanes of a step with the same digit find each other with
      // eight ballots (zh_peers8); everybody reads the wave's counter of the digit — what the steps before put there — and the first of them
      // moves it. (A returning LDS atomic per lane would do it in one instruction, and costs a wave a cycle per lane: zh_mf_group_lds.h.)
      uint32_t o4[ZH_MF_UNROLL];
#pragma unroll
      for (uint32_t u = 0; u < ZH_MF_UNROLL; u++) {
         const uint32_t d = d4[u];
         const bool valid = d != 0xffffffffu;
         const uint64_t peers = zh_peers8(d, valid);
         const uint32_t rank = zh_rank_below(peers);
         const uint32_t before = valid ? hist[wave * 256 + d] : 0u;
         zh_lockstep_point();   // (every lane has read the counter before the first of its peers moves it)
         if (valid && rank == 0) hist[wave * 256 + d] = before + (uint32_t)zh_popc64(peers);
         zh_lockstep_point();
         o4[u] = before + rank;
      }
      if (tile0 + tile < M) ZH_MF_REQUEST(tile0 + tile);
      zh_sync_lds();   // (LDS only: the requests above stay in flight)
      if (tid < 256) {
         uint32_t run = cursor[tid];
         for (uint32_t w2 = 0; w2 < ZH_MF_WAVES; w2++) {
            const uint32_t cnt = hist[w2 * 256 + tid];
            hist[w2 * 256 + tid] = run;
            run += cnt;
         }
         cursor[tid] = run;
      }
      zh_sync_lds();   // (LDS only: the requests above stay in flight)
#pragma unroll
      for (uint32_t u = 0; u < ZH_MF_UNROLL; u++) {
         const uint32_t e = e4[u], d = d4[u];
         const bool valid = d != 0xffffffffu;
         if (valid) {
            const uint32_t out = hist[wave * 256 + d] + o4[u];
            dst[out] = MODE == 11 ? (e & 0xffffffu) : e;
            if (PAY == 1) pdst[out] = p4[u];
            if (PAY == 2) pdst2[out] = make_uint2(p4[u], q4[u]);
            if (NEXT >= 0) {
               // the next pass's digit of this element (which wave counts it does not matter: the next pass adds them up)
               const bool has = NEXT < 5 || NEXT == 11 || e + (uint32_t)(NEXT - 2) < W;
               if (has) {
                  const uint32_t d2 = NEXT == 11 ? e >> 24 : (uint32_t)gwin[(e & 0xffffffu) + (NEXT == 1 ? 1u : (NEXT == 2 ? 0u : (uint32_t)(NEXT - 2)))];
                  atomicAdd(&hist_next[wave * 256 + d2], 1u);
               }
            }
         }
      }
      zh_lockstep_sync();   // (the next tile zeroes this wave's counters: after these reads)
      // (the next tile's zeroing of this wave's counters comes after these reads in program order; the other waves' are not touched)
   }
#undef ZH_MF_FETCH
#undef ZH_MF_FETCH_E
#undef ZH_MF_FETCH_D
#undef ZH_MF_REQUEST
   __threadfence_block();
   __syncthreads();
}

// ---- byte runs ------------------------------------------------------------------------------------------------
// Positions inside a run of one repeated byte (indentation, zero padding) all share the 4-gram "cccc"; scanning that
// class position by position is where real text and binaries spend their time (measured: 93 % of all candidates on
// Python sources). But for a position with r bytes of its run left, a candidate with r_p bytes left matches exactly
// min(r, r_p) bytes unless r_p == r — so per earlier run only ONE position can beat the current length. The frontier
// of such positions is therefore computed from a table of the window's runs (zh_mf_frontier), not from the class scan.
// Table layout per max-block (u32, Q = W/4 + 1): start[Q] grouped by byte value and ascending within a byte, length[Q]
// in the same order, then first[256] / end[256] (slice of each byte value) and the run count; then the same runs in a second
// order (zh_mf_build_runs): start | following byte << 24 [Q], length [Q]. 4 Q + 513 words <= the window size + 576.
#define ZH_RUN_MIN 4
__device__ __forceinline__ uint32_t zh_runs_q(uint32_t W) { return W / 4 + 1; }

// remaining length of the run of byte g[q] starting at q, capped at `cap` (g readable a few bytes past W)
__device__ __forceinline__ uint32_t zh_run_length(const uint8_t *g, uint32_t q, uint32_t W, uint32_t cap) {
   const uint32_t c4 = (uint32_t)g[q] * 0x01010101u;
   uint32_t l = 1;
   const uint32_t lim = min(cap, W - q);
   while (l + 4 <= lim && zh_ld32(g + q + l) == c4) l += 4;
   while (l < lim && g[q + l] == g[q]) l++;
   return l;
}

#ifdef ZH_MFG_PROFILE
// probe builds only (tools/mfg_profile.py): cycles of thread 0 of every workgroup by phase — 0 window staging and ticket, 1 the two passes through
// HBM, 2 chunk load and boundary, 3 the four passes of a chunk, 4 its sweep, 5 oversized classes, 6 run starts, 7 first run order,
// 8 run lengths, 9 second run order, 10 its table; counts — 12 chunks, 13 oversized classes, 14 their entries, 15 segments
__device__ unsigned long long zh_mfg_prof[32];   // 16..23: inside a pass of a chunk — counting, wait, totals, wait, scan and places, wait, scatter, wait
#define ZH_MFG_LAP(slot_) do { if (threadIdx.x == 0) { const uint64_t n_ = zh_clock(); atomicAdd(&zh_mfg_prof[slot_], (unsigned long long)(n_ - mfg_t_)); mfg_t_ = n_; } } while (0)
#define ZH_MFG_COUNT(slot_, n_) do { if (threadIdx.x == 0) atomicAdd(&zh_mfg_prof[slot_], (unsigned long long)(n_)); } while (0)
#define ZH_MFG_TIC() mfg_t_ = zh_clock()
#else
#define ZH_MFG_LAP(slot_)
#define ZH_MFG_COUNT(slot_, n_)
#define ZH_MFG_TIC()
#endif

__device__ inline void zh_mf_build_runs(const uint8_t *win, const uint8_t *gwin, uint32_t W, uint32_t Qn, uint32_t *T, uint32_t *runs, uint32_t *hist, uint32_t *wave_tot, uint64_t &mfg_t_) {
   const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
   const uint32_t Q = zh_runs_q(W);
   uint32_t *RS = runs, *RL = runs + Q, *first = runs + 2 * Q, *end = first + 256, *count = end + 256;
   const uint32_t M = min(Qn, W >= ZH_RUN_MIN ? W - (ZH_RUN_MIN - 1) : 0u);   // look-ahead positions start no runs of interest
   const uint32_t seg = (((M + ZH_MF_WAVES - 1) / ZH_MF_WAVES) + 63) & ~63u;
   const uint32_t lo = min(M, wave * seg), hi = min(M, lo + seg);
   const uint64_t lt_mask = (1ull << lane) - 1;
   if (tid < 256) {
      first[tid] = 0;
      end[tid] = 0;
   }
#define ZH_IS_RUN_START(q) ((q) < hi && gwin[q] == gwin[(q) + 1] && gwin[q] == gwin[(q) + 2] && gwin[q] == gwin[(q) + 3] && ((q) == 0 || gwin[(q) - 1] != gwin[q]))
   uint32_t cnt = 0;
   for (uint32_t base = lo; base < hi; base += 64) cnt += (uint32_t)zh_popc64(zh_ballot(ZH_IS_RUN_START(base + lane)));
   if (lane == 0) wave_tot[wave] = cnt;
   __syncthreads();
   uint32_t off = 0, total = 0;
   for (uint32_t w2 = 0; w2 < ZH_MF_WAVES; w2++) {
      if (w2 < wave) off += wave_tot[w2];
      total += wave_tot[w2];
   }
   for (uint32_t base = lo; base < hi; base += 64) {   // ordered compaction: T = run starts in position order
      const bool f = ZH_IS_RUN_START(base + lane);
      const uint64_t m = zh_ballot(f);
      if (f) T[off + (uint32_t)zh_popc64(m & lt_mask)] = base + lane;
      off += (uint32_t)zh_popc64(m);
   }
#undef ZH_IS_RUN_START
   __threadfence_block();
   __syncthreads();
   ZH_MFG_LAP(6);
   zh_mf_sort_pass<2>(win, gwin, total, T, RS, hist, wave_tot);   // stable by byte value: digit = gwin[start]
   ZH_MFG_LAP(7);
   for (uint32_t idx = tid; idx < total; idx += ZH_MF_THREADS) {
      const uint32_t q = RS[idx];
      const uint32_t c = gwin[q];
      RL[idx] = zh_run_length(gwin, q, W, 0xffffffffu);
      if (idx == 0 || gwin[RS[idx - 1]] != c) first[c] = idx;
      if (idx + 1 == total || gwin[RS[idx + 1]] != c) end[c] = idx + 1;
   }
   if (tid == 0) *count = total;
   // The same runs once more, ordered by (byte, byte that follows the run, position): a position with r run bytes left can only
   // find a match longer than r at the one position of an earlier run that has exactly r bytes left too, and only if that run is
   // followed by the same byte — the walk over "earlier runs of this byte" then skips every run that ends differently
   // (indentation in source code: a thousand runs of spaces within reach, a few dozen of them followed by the same letter).
   // Entry j of this order: RS2[j] = start | following byte << 24, RL2[j] = length.
   uint32_t *RS2 = count + 1, *RL2 = RS2 + Q;
   uint32_t *Pa = T + Q, *Pb = T + 2 * Q;
   __threadfence_block();
   __syncthreads();
   ZH_MFG_LAP(8);
   zh_mf_sort_pass<8>(win, gwin, total, nullptr, Pa, hist, wave_tot, W, nullptr, 0, RS, RL);
   zh_mf_sort_pass<9>(win, gwin, total, Pa, Pb, hist, wave_tot, W, nullptr, 0, RS, RL);
   ZH_MFG_LAP(9);
   for (uint32_t idx = tid; idx < total; idx += ZH_MF_THREADS) {
      const uint32_t e = Pb[idx];
      const uint32_t st = RS[e], ln = RL[e];
      const uint32_t x = st + ln < W ? (uint32_t)gwin[st + ln] : 0u;
      RS2[idx] = st | (x << 24);
      RL2[idx] = ln;
   }
   ZH_MFG_LAP(10);
}

// The nearest earlier position sharing K bytes (K = 3, 4, 5) with the position at entry idx of the K-gram order X (MK entries) is
// the entry before it, when that one is in the same class (classes ascend in position). Recorded as distance - 1 in 16 bits
// (0xffff: none within ZH_MAX_DIST) IN THE ORDER OF X, and carried along by the passes that refine the order (zh_mf_sort_pass,
// PAY): in the end the three distances of a position sit next to its entry of the 6-gram order, where zh_mf_frontier reads them
// with the entries, coalesced. (Round 1 scattered them into a table indexed by position: three partial writes per position,
// to lines that had long left the L2 — 8.1 GB of HBM writes per 50 MB of input, most of this kernel's time.)
//   K = 3: pay[idx] = d3 | none << 16        K = 4: pay[idx] = d3 (as carried) | d4 << 16        K = 5: pay[idx] = d5
// The last five window positions drop out of the orders before they reach the 6-gram order: theirs go to tail[pos] (the
// frontier's position-indexed table: entries from W - 5 on are free, the 6-gram order has at most W - 5 entries).
template <int K>
__device__ inline void zh_mf_prev_level(const uint32_t *__restrict__ X, uint32_t MK, const uint8_t *gwin, uint32_t W, uint32_t *pay, uint2 *tail) {
   for (uint32_t idx0 = threadIdx.x; idx0 < MK; idx0 += 4 * ZH_MF_THREADS) {
      uint32_t pos[4], q[4], old[4] = {0, 0, 0, 0};
#pragma unroll
      for (uint32_t u = 0; u < 4; u++) {
         const uint32_t idx = idx0 + u * ZH_MF_THREADS;
         pos[u] = idx < MK ? X[idx] : 0u;
         q[u] = (idx < MK && idx > 0) ? X[idx - 1] : ZH_MF_NONE;
         if (K == 4 && idx < MK) old[u] = pay[idx];
      }
#pragma unroll
      for (uint32_t u = 0; u < 4; u++) {
         const uint32_t idx = idx0 + u * ZH_MF_THREADS;
         if (idx < MK) {
            bool same = q[u] != ZH_MF_NONE;
            if (same) {
               if (K == 3)
                  same = zh_ld24(gwin + q[u]) == zh_ld24(gwin + pos[u]);
               else
                  same = zh_ld32(gwin + q[u]) == zh_ld32(gwin + pos[u]) && (K == 4 || gwin[q[u] + 4] == gwin[pos[u] + 4]);
            }
            const uint32_t dist = pos[u] - q[u];
            const uint32_t d = (same && dist <= ZH_MAX_DIST) ? dist - 1u : 0xffffu;
            const uint32_t v = K == 3 ? (d | 0xffff0000u) : (K == 4 ? ((old[u] & 0xffffu) | (d << 16)) : d);
            pay[idx] = v;
            if (pos[u] + 5u >= W) {
               if (K == 3)
                  tail[pos[u]] = make_uint2(v, 0xffffu);
               else if (K == 4)
                  tail[pos[u]].x = v;
               else
                  tail[pos[u]].y = v;
            }
         }
      }
   }
   __syncthreads();
}

// `win` is read linearly (global memory); `gwin` is the copy used for scattered reads (LDS when the window fits)
__device__ inline void zh_mf_group_body(const uint8_t *win, const uint8_t *gwin, uint32_t W, uint32_t Qn, uint32_t first_needed, uint32_t *A, uint32_t *B,
                                        uint2 *prev, uint32_t *pay, uint64_t pay_stride, uint32_t *runs, uint32_t *hist, uint32_t *wave_tot, int stop) {
   uint32_t *Pa = pay, *Pb = pay + pay_stride;   // payload ping, pong
   const uint32_t tid = threadIdx.x;
   // W = window bytes, Qn = positions that are candidates or get rows (the rest of the window is look-ahead)
   const uint32_t M3 = min(Qn, W >= 3 ? W - 2 : 0u);   // positions that start a trigram
   const uint32_t M4 = min(Qn, W >= 4 ? W - 3 : 0u);   // positions that start a 4-gram

   // ---- exact trigram order -> previous occurrence of every trigram -------------------------------------------
   if (stop == 1) return;
   uint32_t *hist2 = hist + ZH_MF_WAVES * 256 + ZH_MF_WAVES + 1;   // second counter table, behind the kernel's small variables
   zh_mf_sort_pass<0, false, 1>(win, gwin, M3, nullptr, A, hist, wave_tot, W, hist2, M3);
   if (stop == 2) return;
   zh_mf_sort_pass<1, true, 2>(win, gwin, M3, A, B, hist2, wave_tot, W, hist, M3);
   zh_mf_sort_pass<2, true, 5>(win, gwin, M3, B, A, hist, wave_tot, W, hist2, M3);
   if (stop == 3) return;
   if (stop == 4) return;

   // ---- 4-, 5- and 6-gram classes: one more stable pass each over the previous order ------------------------------
   // prev4 / prev5 (nearest earlier position sharing 4 / 5 bytes) give the records of length 4 and 5 directly; only
   // matches of 6 and more are found by walking a class, and 6-gram classes are several times smaller than 4-gram
   // classes (on text the walk shrinks 2.3x, on source code 1.5x). Each pass takes the distances of its input order as it reads it.
   const uint32_t M5 = min(Qn, W >= 5 ? W - 4 : 0u), M6 = min(Qn, W >= 6 ? W - 5 : 0u);
   zh_mf_sort_pass<5, true, 6, 1, 3>(win, gwin, M3, A, B, hist2, wave_tot, W, hist, M4, nullptr, nullptr, nullptr, Pb, nullptr, nullptr, prev);   // B: 4-gram order, M4 entries; Pb: d3
   zh_mf_sort_pass<6, true, 7, 1, 4>(win, gwin, M4, B, A, hist, wave_tot, W, hist2, M5, nullptr, nullptr, Pb, Pa, nullptr, nullptr, prev);      // A: 5-gram order, M5 entries; Pa: d3 | d4 << 16
   // B: 6-gram order, M6 entries; prev[idx] = the distances of the position at B[idx]
   zh_mf_sort_pass<7, true, -1, 2, 5>(win, gwin, M5, A, B, hist2, wave_tot, W, nullptr, 0, nullptr, nullptr, Pa, nullptr, nullptr, prev, prev);
   if (stop == 5) return;
   {
      const uint32_t *__restrict__ Br = B;
      uint32_t *__restrict__ Aw = A;
      for (uint32_t idx0 = tid; idx0 < M6; idx0 += 4 * ZH_MF_THREADS) {
         uint32_t e[4], eq[4];
#pragma unroll
         for (uint32_t u = 0; u < 4; u++) {
            const uint32_t idx = idx0 + u * ZH_MF_THREADS;
            e[u] = idx < M6 ? Br[idx] : 0u;
            eq[u] = (idx < M6 && idx > 0) ? Br[idx - 1] : 0u;
         }
#pragma unroll
         for (uint32_t u = 0; u < 4; u++) {
            const uint32_t idx = idx0 + u * ZH_MF_THREADS;
            if (idx < M6) {
               const bool head = idx == 0 || zh_ld32(gwin + eq[u]) != zh_ld32(gwin + e[u]) ||
                                 (((uint32_t)gwin[eq[u] + 4] | ((uint32_t)gwin[eq[u] + 5] << 8)) != ((uint32_t)gwin[e[u] + 4] | ((uint32_t)gwin[e[u] + 5] << 8)));
               Aw[idx] = e[u] | (head ? ZH_MF_HEAD : 0u);   // the scan stops after consuming a marked entry
            }
         }
      }
   }
   __syncthreads();
   uint64_t mfg_t_ = 0;
   zh_mf_build_runs(win, gwin, W, Qn, B, runs, hist, wave_tot, mfg_t_);   // B is free again: scratch for the run starts
}

// 4 bytes at an arbitrary byte offset x of a dword-aligned buffer: two aligned dword reads + funnel shift
// (the buffer must be readable up to 7 bytes past x). ZH_MF_UNALIGNED32 / 128 = 1 (A/B builds, tools/build_variant.sh): the hardware's
// reads at any byte address instead (zh_platform.h) — measured on gfx950: the 4-byte one is slower than this (frontier 6.6 -> 10.3 ms
// per 50 MB), the 16-byte one the same as five words.
#ifndef ZH_MF_UNALIGNED32
#define ZH_MF_UNALIGNED32 0
#endif
#ifndef ZH_MF_UNALIGNED128
#define ZH_MF_UNALIGNED128 0
#endif
__device__ __forceinline__ uint32_t zh_load32_at(const uint32_t *w32, uint32_t x) {
#if ZH_MF_UNALIGNED32
   return zh_load32_any((const uint8_t *)w32 + x);
#else
   const uint32_t lo = w32[x >> 2], hi = w32[(x >> 2) + 1];
   return zh_funnel(hi, lo, x * 8u);   // (the shift counts modulo 32)
#endif
}
#include "zh_mf_group_lds.h"

// lds_cap: elements per chunk of the in-LDS refinement (zh_mf_group_lds.h; the layout may allow fewer); 0: rounds 1-3's six passes through HBM
template <bool LDS_WIN>
__global__ void __launch_bounds__(ZH_MF_THREADS)
zh_mf_group(const uint8_t *__restrict__ data, const zh_seg_t *__restrict__ segs, uint32_t *sort_a,
            uint32_t *sort_b, uint2 *prev_all, uint32_t *runs_all, uint64_t sort_stride, uint64_t run_stride, int stop, uint32_t nsegs,
            uint32_t *ticket, uint32_t *pay_all /* 3 x sort_stride words per workgroup: the payload of the refining passes */, uint32_t lds_cap) {
   // All of this kernel's LDS is dynamic (ZH_MF_GROUP_LDS bytes at launch). Measured on gfx950: next to a workgroup with
   // 128 KiB of STATIC LDS, workgroups of another stream's kernel that use static LDS are not scheduled at all although
   // they would fit (a 0.3 ms kernel took 6.4 ms); with the allocation made dynamic on either side they run side by side
   // (tools/probes/corun_probe.hip).
   ZH_DYN_LDS(dyn_lds);
   uint32_t *lwin32 = dyn_lds;                                   // ZH_MF_LDS_WINDOW / 4 + 4 words
   uint32_t *hist = dyn_lds + ZH_MF_LDS_WINDOW / 4 + 4;          // ZH_MF_WAVES x 256
   uint32_t *wave_tot = hist + ZH_MF_WAVES * 256;                // ZH_MF_WAVES
   uint32_t &cur_seg = wave_tot[ZH_MF_WAVES];
   // Persistent workgroups (the host launches one per CU) take segments from a ticket counter. A grid of one workgroup
   // per segment would keep workgroups of 114 KiB LDS waiting for a CU throughout the kernel, and while such a workgroup
   // waits the dispatcher holds back the small kernels of the other run's stream (measured: a 0.3 ms kernel took 6.4 ms).
   uint64_t mfg_t_ = 0;
   ZH_MFG_TIC();
   for (;;) {
   __syncthreads();   // the previous segment is done with LDS (and with cur_seg)
   if (threadIdx.x == 0) cur_seg = atomicAdd(ticket, 1u);
   __syncthreads();
   const uint32_t seg = cur_seg;
   if (seg >= nsegs) return;
   const zh_seg_t blk = segs[seg];
   const uint8_t *win = data + blk.win_off;
   const uint32_t W = blk.prev + blk.n + blk.tail;
   uint32_t *A = sort_a + (uint64_t)seg * sort_stride;
   uint32_t *B = sort_b + (uint64_t)seg * sort_stride;
   uint2 *prev3 = prev_all + (uint64_t)seg * sort_stride;
   uint32_t *runs = runs_all + (uint64_t)seg * run_stride;
   const uint8_t *gwin = win;
   if (LDS_WIN) {
      zh_stage_window(lwin32, win, W);
      gwin = (const uint8_t *)lwin32;
   }
#ifdef ZH_MF_GROUP_HBM   // A/B builds (tools/build_variant.sh): rounds 1-3's six passes through HBM when the host passes lds_cap = 0 (ZULTRA_HIP_MF_CAP=0)
   if (!LDS_WIN || !lds_cap)
   {
      zh_mf_group_body(win, gwin, W, blk.prev + blk.n, blk.prev, A, B, prev3, pay_all + (uint64_t)blockIdx.x * 3u * sort_stride, sort_stride, runs, hist, wave_tot,
                       stop);
      if (threadIdx.x == 0) runs[run_stride - ZH_MFL_NOTE_WORDS] = 0;   // (nothing for zh_mf_group_big)
   }
   else
#endif
      zh_mf_group_body_lds(win, dyn_lds, W, blk.prev + blk.n, A, B, prev3, pay_all + (uint64_t)blockIdx.x * 3u * sort_stride, sort_stride, runs, runs + run_stride - ZH_MFL_NOTE_WORDS, hist, wave_tot, stop,
                           lds_cap, mfg_t_);
   }
}

// The bigram classes that fit no chunk of zh_mf_group (zh_mf_group_lds.h: byte runs of a few bytes each, alphabets of two or three symbols):
// noted there, refined here by the generic passes through HBM — a kernel of its own so that zh_mf_group keeps its registers. Persistent
// workgroups take the segments from a ticket; a segment without notes costs a load.
__global__ void __launch_bounds__(ZH_MF_THREADS)
zh_mf_group_big(const uint8_t *__restrict__ data, const zh_seg_t *__restrict__ segs, uint32_t *sort_a, uint32_t *sort_b, uint2 *prev_all, const uint32_t *__restrict__ runs_all,
                uint64_t sort_stride, uint64_t run_stride, uint32_t nsegs, uint32_t *ticket, uint32_t *pay_all) {
   ZH_DYN_LDS(dyn_lds);
   uint32_t *lwin32 = dyn_lds;
   uint32_t *hist = dyn_lds + ZH_MF_LDS_WINDOW / 4 + 4;
   uint32_t *wave_tot = hist + ZH_MF_WAVES * 256;
   uint32_t &cur_seg = wave_tot[ZH_MF_WAVES];
   for (;;) {
      __syncthreads();
      if (threadIdx.x == 0) {
         // (the next segment that has notes: thread 0 skips the others)
         uint32_t sg;
         do {
            sg = atomicAdd(ticket, 1u);
         } while (sg < nsegs && (runs_all + (uint64_t)sg * run_stride + run_stride - ZH_MFL_NOTE_WORDS)[0] == 0);
         cur_seg = sg;
      }
      __syncthreads();
      const uint32_t seg = cur_seg;
      if (seg >= nsegs) return;
      const zh_seg_t blk = segs[seg];
      const uint32_t W = blk.prev + blk.n + blk.tail;
      const uint32_t *notes = runs_all + (uint64_t)seg * run_stride + run_stride - ZH_MFL_NOTE_WORDS;
      zh_stage_window(lwin32, data + blk.win_off, W);
      const uint32_t nn = notes[0];
      for (uint32_t k = 0; k < nn; k++) {
         const uint32_t a = notes[1u + 2u * k], b = notes[2u + 2u * k];
         const uint32_t cs = a & 0x1ffffu, cob = b & 0x1ffffu, cn = (a >> 17) | ((b >> 17) << 15);
         (void)zh_mfl_oversized((const uint8_t *)lwin32, W, cn, sort_b + (uint64_t)seg * sort_stride + cs, sort_a + (uint64_t)seg * sort_stride + cob,
                                prev_all + (uint64_t)seg * sort_stride + cob, prev_all + (uint64_t)seg * sort_stride, pay_all + (uint64_t)blockIdx.x * 3u * sort_stride, hist, wave_tot);
      }
   }
}

// ---------------------------------------------------------------------------------------------------------
// zh_mf_frontier
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t zh_trigram(const uint8_t *p) {
   return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
}

// the 16 bytes at byte offset x: five aligned words, all in flight together
__device__ __forceinline__ void zh_load128_at(const uint32_t *w32, uint32_t x, uint32_t out[4]) {
#if ZH_MF_UNALIGNED128
   const zh_u128_any_t v = zh_load128_any((const uint8_t *)w32 + x);
   out[0] = v.x;
   out[1] = v.y;
   out[2] = v.z;
   out[3] = v.w;
#else
   const uint32_t *p = w32 + (x >> 2);
   const uint32_t a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3], a4 = p[4];
   const uint32_t sh = x * 8u;
   out[0] = zh_funnel(a1, a0, sh);
   out[1] = zh_funnel(a2, a1, sh);
   out[2] = zh_funnel(a3, a2, sh);
   out[3] = zh_funnel(a4, a3, sh);
#endif
}

// match length of a candidate's first 16 bytes (c16) and the lane's own (own16): 0..16. Without branches: the loads of a walk step
// (two candidates, five aligned words each) are then all in flight together — with early exits the compiler fetched a candidate's
// words one round trip at a time
__device__ __forceinline__ uint32_t zh_mf_len16(const uint32_t (&c16)[4], const uint32_t (&own16)[4]) {
   const uint32_t x0 = c16[0] ^ own16[0], x1 = c16[1] ^ own16[1], x2 = c16[2] ^ own16[2], x3 = c16[3] ^ own16[3];
   const uint32_t xs = x0 ? x0 : x1 ? x1 : x2 ? x2 : x3;
   const uint32_t base = x0 ? 0u : x1 ? 4u : x2 ? 8u : 12u;
   return xs ? base + ((uint32_t)__builtin_ctz(xs) >> 3) : 16u;
}

#ifdef ZH_MF_PROFILE
// probe builds only (tools/mf_profile.py): wave-cycles of zh_mf_frontier by phase — 0 window staging, 1 chunk head (entries, prev
// records, first records), 2 byte-run path, 3 class walk, 4 row store; counts — 6 chunks, 8 walk steps, 9 lanes alive in them, 10 cycles
// up to the probes' answer, 11 cycles of the verification, 12 steps with a verification, 13 rounds of zh_mf_extend_wave. A wave keeps
// them in registers and adds them up once per segment: an atomic per lap would sit in front of the next wait for a load.
__device__ unsigned long long zh_mf_prof[16];
#define ZH_MF_PROF_VARS() uint64_t pf_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pf_t_ = 0, pf_w_ = 0, pf_s_ = 0
#define ZH_MF_PROF_FLUSH() do { if (lane == 0) { for (int k_ = 0; k_ < 16; k_++) { if (pf_[k_]) atomicAdd(&zh_mf_prof[k_], (unsigned long long)pf_[k_]); pf_[k_] = 0; } } } while (0)
#define ZH_MF_TIC() pf_t_ = zh_clock()
#define ZH_MF_LAP(slot_) do { const uint64_t now_ = zh_clock(); pf_[slot_] += now_ - pf_t_; pf_t_ = now_; } while (0)
#define ZH_MF_COUNT(slot_, n_) pf_[slot_] += (n_)
#define ZH_MF_SUBTIC() pf_s_ = zh_clock()
#define ZH_MF_SUBLAP(slot_) do { const uint64_t now_ = zh_clock(); pf_[slot_] += now_ - pf_s_; pf_s_ = now_; } while (0)
#define ZH_MF_WALK_TOP(mask_) do { pf_w_ = zh_clock(); pf_[8]++; pf_[9] += (uint32_t)zh_popc64(mask_); } while (0)
#define ZH_MF_WALK_PROBED() do { const uint64_t n_ = zh_clock(); pf_[10] += n_ - pf_w_; pf_w_ = n_; } while (0)
#define ZH_MF_WALK_VERIFIED() do { pf_[11] += zh_clock() - pf_w_; pf_[12]++; } while (0)
#define ZH_MF_WALK_EXT(need_) pf_[13] += (uint32_t)zh_popc64(zh_ballot(need_))
#else
#define ZH_MF_PROF_VARS()
#define ZH_MF_PROF_FLUSH()
#define ZH_MF_TIC()
#define ZH_MF_LAP(slot_)
#define ZH_MF_COUNT(slot_, n_)
#define ZH_MF_SUBTIC()
#define ZH_MF_SUBLAP(slot_)
#define ZH_MF_WALK_TOP(mask_) (void)(mask_)
#define ZH_MF_WALK_PROBED()
#define ZH_MF_WALK_VERIFIED()
#define ZH_MF_WALK_EXT(need_)
#endif

// Matches that run past 16 bytes are finished by the whole wave: for every lane with `need` in turn, lane k compares the four
// bytes at offset 16 + 4k of the two strings (64 lanes cover 16..271: every length up to 258 in one step), a ballot finds the
// first difference. A lane stepping through its own long match four bytes at a time kept the other 63 waiting for up to 60
// rounds — on source code and other repetitive data that was most of the walk. All lanes call; returns l, or the lane's full
// match length (not yet clamped to maxlen beyond the comparison range) where it had `need`.
template <uint32_t BASE = 16u>
__device__ __forceinline__ uint32_t zh_mf_extend_wave(const uint32_t *lwin32, bool need, uint32_t q, uint32_t i, uint32_t maxlen, uint32_t l) {
   uint64_t todo = zh_ballot(need);
   const uint32_t lane = zh_lane();
   const uint32_t off = BASE + 4u * lane;
   while (todo) {
      const int src = zh_ctz64(todo);
      todo &= todo - 1;
      const uint32_t qq = zh_readlane(q, src), ii = zh_readlane(i, src), ml = zh_readlane(maxlen, src);
      const bool beyond = off >= ml;   // (also keeps every read inside the window)
      uint32_t x = 0;
      if (!beyond) x = zh_load32_at(lwin32, qq + off) ^ zh_load32_at(lwin32, ii + off);
      const uint64_t stop = zh_ballot(beyond || x != 0);   // lane 63 compares offset 268 or more: always beyond
      const int fl = zh_ctz64(stop);
      const uint32_t xf = zh_readlane(x, fl);
      const uint32_t len = BASE + 4u * (uint32_t)fl + (xf ? ((uint32_t)(__ffs((int)xf) - 1) >> 3) : 0u);
      if ((int)lane == src) l = min(len, ml);
   }
   return l;
}


// A lane on its own (the byte-run path, where the lanes of a chunk are at different stages): the strings at p and i agree on their first
// l bytes; 16 more per round trip. (Round 2 took four: a 40-byte match of an indented line was ten dependent LDS round trips.)
__device__ __forceinline__ uint32_t zh_mf_extend_lane(const uint32_t *lwin32, uint32_t p, uint32_t i, uint32_t l, uint32_t maxlen) {
   while (l < maxlen) {
      uint32_t a[4], b[4];
      zh_load128_at(lwin32, p + l, a);
      zh_load128_at(lwin32, i + l, b);
      const uint32_t m = zh_mf_len16(a, b);
      l += m;
      if (m < 16u) break;
   }
   return min(l, maxlen);
}

// ... and FOUR lanes at a time (round 6): the wave as four groups of sixteen lanes, group g serving the g-th lane in need — lane k of a group compares the SIXTEEN bytes at
// offset BASE + 16 k of its group's two strings (five aligned words each, as everywhere in the walk), so one round covers BASE .. BASE + 255: every length up to 258
// for BASE >= 18. A wave instruction of the whole-wave form above compares 4 bytes per lane for ONE match; here it compares 16 bytes per lane for four: ~75
// instructions per round of four against ~30 per match, and the per-lane loop over bytes 32..79 that ran in front of the whole-wave form — three rounds of ~35
// instructions for the two or three lanes of a step that get past 32 bytes, the other sixty idle — is gone. (tools/mf_profile.py, Python sources: 2.15 whole-wave rounds
// per walk step, a third of the walk's cycles.) All lanes call; returns l, or the lane's match length (clamped to maxlen) where it had `need`.
template <uint32_t BASE>
__device__ __forceinline__ uint32_t zh_mf_extend_groups(const uint32_t *lwin32, bool need, uint32_t q, uint32_t i, uint32_t maxlen, uint32_t l) {
   static_assert(BASE + 240u >= ZH_MAX_MATCH, "a group's last lane lies beyond every match: there is always a lane that stops");
   uint64_t todo = zh_ballot(need);
   const uint32_t lane = zh_lane(), grp = lane >> 4, sub = lane & 15u;
   const uint32_t off = BASE + 16u * sub;
   while (todo) {
      // the next four lanes in need (fewer left: the last one again — the same answer twice)
      const int s0 = zh_ctz64(todo);
      todo &= todo - 1;
      const int s1 = todo ? zh_ctz64(todo) : s0;
      todo &= todo - 1;
      const int s2 = todo ? zh_ctz64(todo) : s1;
      todo &= todo - 1;
      const int s3 = todo ? zh_ctz64(todo) : s2;
      todo &= todo - 1;
      const int src = grp == 0 ? s0 : (grp == 1 ? s1 : (grp == 2 ? s2 : s3));
      const uint32_t qq = zh_shfl(q, src), ii = zh_shfl(i, src), ml = zh_shfl(maxlen, src);
      const bool beyond = off >= ml;   // (also keeps every read within sixteen bytes of the window's end: inside this kernel's LDS)
      uint32_t m = 0;                  // how many of the lane's sixteen bytes agree
      if (!beyond) {
         uint32_t a[4], b[4];
         zh_load128_at(lwin32, qq + off, a);
         zh_load128_at(lwin32, ii + off, b);
         m = zh_mf_len16(a, b);
      }
      const uint64_t stop = zh_ballot(beyond || m < 16u);
      const uint32_t g16 = (uint32_t)(stop >> (16u * grp)) & 0xffffu;   // the group's lanes that stop: never empty (its last lane is beyond)
      const uint32_t fl = (uint32_t)__builtin_ctz(g16 | 0x8000u);
      const uint32_t mf = zh_shfl(m, (int)((grp << 4) + fl));            // (0 where that lane is beyond: the length is clamped to ml below)
      const uint32_t len = min(BASE + 16u * fl + mf, ml);
      // a lane in need takes the answer of the group that served it
      const uint32_t mine = (int)lane == s0 ? 0u : ((int)lane == s1 ? 1u : ((int)lane == s2 ? 2u : 3u));
      const uint32_t got = zh_shfl(len, (int)(mine << 4));
      if ((int)lane == s0 || (int)lane == s1 || (int)lane == s2 || (int)lane == s3) l = got;
   }
   return l;
}

#ifndef ZH_MF_EXT_GROUPS
#define ZH_MF_EXT_GROUPS 1   // 0 (A/B builds): rounds 4-5's form — the lanes extend their own matches up to ZH_MF_EXT_TO, the whole wave beyond, one lane at a time
#endif

// The length of a match that agrees on its first 16 bytes (lanes with `need`; all lanes call): bytes 16..31 are compared by the lane itself
// against its own bytes 16..31 in registers — every lane that needs it at once, and half of the long matches of source code end there —
// and only what still agrees at 32 goes to the whole wave, one lane at a time.
__device__ __forceinline__ uint32_t zh_mf_length_past16(const uint32_t *lwin32, bool need, uint32_t q, uint32_t i, uint32_t maxlen, const uint32_t (&own32)[4], uint32_t l) {
   if (!zh_ballot(need)) return l;   // (wave-uniform)
   uint32_t g[4];
   zh_load128_at(lwin32, need ? q + 16u : 0u, g);
   const uint32_t m = zh_mf_len16(g, own32);
   if (need) l = 16u + m;
   bool more = need && m == 16u && maxlen > 32u;
#if ZH_MF_EXT_GROUPS
   return zh_mf_extend_groups<32u>(lwin32, more, q, i, maxlen, l);
#else
   // Bytes 32..79 by the lanes themselves, sixteen at a time and all that need it at once (round 4): on source code two lanes of a step
   // go past 32 on average and most of them end before 80 — handing each of them to the whole wave in turn (zh_mf_extend_wave: ~350 cycles
   // per lane) was a fifth of this kernel there. What still agrees at 80 goes to the whole wave.
#pragma unroll
   for (uint32_t off = 32u; off < ZH_MF_EXT_TO; off += 16u) {
      if (!zh_ballot(more)) return l;   // (wave-uniform)
      uint32_t a[4], b[4];
      zh_load128_at(lwin32, more ? q + off : 0u, a);
      zh_load128_at(lwin32, more ? i + off : 0u, b);
      const uint32_t mm = zh_mf_len16(a, b);
      if (more) {
         l = off + mm;
         more = mm == 16u && l < maxlen;
      }
   }
   return zh_mf_extend_wave<ZH_MF_EXT_TO>(lwin32, more, q, i, maxlen, l);
#endif
}

template <bool LDS_WIN>
__global__ void __launch_bounds__(ZH_MF_THREADS)
zh_mf_frontier(const uint8_t *__restrict__ data, const zh_seg_t *__restrict__ segs,
               const uint32_t *__restrict__ sorted, const uint2 *__restrict__ prev_all, const uint32_t *__restrict__ runs_all,
               uint64_t sort_stride, uint64_t run_stride, zh_match_t *match, uint64_t match_stride, uint32_t *longest_all, uint64_t longest_stride,
               uint32_t *chunk_ctr, uint32_t nsegs, uint32_t steal) {
   // Persistent workgroups (one per CU, see zh_mf_group) take segments from a ticket counter, then help: on real data a
   // few segments carry several times the average scan work (measured: 7x on source code), and a workgroup that only did
   // its own would leave the chip waiting for them. Every segment hands out its 64-entry chunks from a counter in HBM, so
   // any number of workgroups can serve one segment.
   ZH_DYN_LDS(dyn_lds);                                            // all dynamic (ZH_MF_FRONTIER_LDS bytes), see zh_mf_group
   uint32_t *lwin32 = dyn_lds;                                     // ZH_MF_LDS_WINDOW / 4 + 4 words
   uint32_t *mring = dyn_lds + ZH_MF_LDS_WINDOW / 4 + 4;           // per thread: ring of the last 8 accepted matches, [slot][thread]
   uint32_t &help_key = mring[8 * ZH_MF_THREADS];
   const uint32_t lane = threadIdx.x & 63;
   uint32_t seg_id = 0;
   bool owner = true, tickets_left = true;
   ZH_MF_PROF_VARS();

   for (;;) {
   if (tickets_left) {
      if (threadIdx.x == 0) help_key = atomicAdd(chunk_ctr + 2 * nsegs, 1u);
      __syncthreads();
      const uint32_t t = help_key;
      __syncthreads();
      if (t < nsegs)
         seg_id = t;
      else
         tickets_left = false;
   }
   if (!tickets_left) {
      // ---- help: every segment has been handed out; join the unfinished one with the most chunks left per workgroup already
      //      on it — if that is worth staging its window for. The unfinished ones are among the last handed out.
      if (!steal) return;
      if (threadIdx.x == 0) help_key = 0;
      __syncthreads();
      const uint32_t lo = nsegs > ZH_MF_HELP_WINDOW ? nsegs - ZH_MF_HELP_WINDOW : 0u;
      for (uint32_t sc = lo + threadIdx.x; sc < nsegs; sc += ZH_MF_THREADS) {
         const zh_seg_t o = segs[sc];
         const uint32_t oQn = o.prev + o.n, oW = oQn + o.tail;
         const uint32_t oM = min(oQn, oW >= 6 ? oW - 5 : 0u);
         const uint32_t ochunks = (oM + 63) >> 6;
         uint32_t opow = 1;
         while (opow < ochunks) opow <<= 1;   // tickets run to the next power of two (see below)
         const uint32_t next = zh_load_relaxed(chunk_ctr + 2 * sc), workers = zh_load_relaxed(chunk_ctr + 2 * sc + 1);
         const uint32_t left = next < opow * 64u ? (((opow * 64u - next) >> 6) * ochunks) / opow : 0u;
         const uint32_t score = min(left / (workers + 1), 0x3fffu);
         if (score >= ZH_MF_HELP_MIN) atomicMax(&help_key, (score << 18) | (((sc * 0x9e3779b1u + blockIdx.x * 0x85ebca6bu) >> 26) << 12) | (sc - lo));
      }
      __syncthreads();
      const uint32_t key = help_key;
      __syncthreads();
      if (!key) return;
      seg_id = lo + (key & 4095u);
      owner = false;
   }
   const zh_seg_t blk = segs[seg_id];
   const uint8_t *gwin = data + blk.win_off;
   const uint32_t prev = blk.prev;
   const uint32_t Qn = blk.prev + blk.n;                             // positions below Qn are candidates / get rows
   const uint32_t W = Qn + blk.tail;                                 // window incl. look-ahead: match lengths clamp here
   const uint32_t M = min(Qn, W >= 6 ? W - 5 : 0u);                  // entries of the 6-gram order
   const uint32_t *S = sorted + (uint64_t)seg_id * sort_stride;
   const uint2 *prevs = prev_all + (uint64_t)seg_id * sort_stride;   // per entry of the 6-gram order: x = distance - 1 to the previous occurrence of the trigram | of the 4-gram << 16, y = of the 5-gram (0xffff: none)
   const uint32_t *runs = runs_all + (uint64_t)seg_id * run_stride;
   // row r = segment position prev + r: slots 0..3 in rows_lo[r], slots 4..7 in rows_hi[r] (zh_common.h). (Round 2 stored the longest
   // match once more in a 4-byte array for the kernels that only follow the greedy chain; a scattered 4-byte store leaves the L2 as a
   // 64-byte sector, 16 x the bytes: those kernels now read slot 0 of the rows, zh_split.h)
   uint4 *rows_lo = (uint4 *)(match + (uint64_t)blk.block * match_stride) + blk.row_off;
   uint4 *rows_hi = rows_lo + ZH_ROW_HI_OFF(match_stride);
   (void)longest_all;
   (void)longest_stride;
   const uint8_t *win = gwin;

   if (LDS_WIN) {
      ZH_MF_TIC();
      zh_stage_window(lwin32, gwin, W);
      win = (const uint8_t *)lwin32;
      ZH_MF_LAP(0);
   }
   if (threadIdx.x == 0) {
      help_key = 0;
      atomicAdd(chunk_ctr + 2 * seg_id + 1, 1u);   // workgroups serving this segment
   }
   __syncthreads();

   // The last five window positions are not in the 6-gram order: their matches cannot be longer than the bytes left, so
   // the nearest earlier occurrences of their 3-, 4- and 5-grams are their whole frontier (matchfinder.c:71: LCP bounded
   // by the window end; W-1 and W-2 cannot start a match).
   if (owner && threadIdx.x < 5 && W >= 1 + threadIdx.x) {
      const uint32_t i = W - 1 - threadIdx.x, maxlen = threadIdx.x + 1;
      if (i >= prev && i < Qn) {
         uint32_t e3 = 0, e4 = 0, e5 = 0;   // the entries of length 3, 4 and 5 (0: none)
         if (maxlen >= 3) {
            const uint2 pv = prevs[i];   // (the last positions' records are indexed by position, zh_mf_prev_level)
            const uint32_t d3 = pv.x & 0xffffu;
            if (d3 != 0xffffu) {
               const uint32_t p3 = i - 1u - d3;
               if (maxlen == 3 || gwin[p3 + 3] != gwin[i + 3]) e3 = 3u | ((d3 + 1u) << 16);
               const uint32_t d4 = pv.x >> 16, d5 = pv.y & 0xffffu;
               if (maxlen >= 4 && d4 != 0xffffu) {
                  if (maxlen == 4 || gwin[i - 1 - d4 + 4] != gwin[i + 4]) e4 = 4u | ((d4 + 1) << 16);
                  if (maxlen >= 5 && d5 != 0xffffu) e5 = 5u | ((d5 + 1) << 16);
               }
            }
         }
         uint4 a;   // rows are longest first (selects, not an array indexed at run time: that would live in scratch memory)
         a.x = e5 ? e5 : (e4 ? e4 : e3);
         a.y = e5 ? (e4 ? e4 : e3) : (e4 ? e3 : 0u);
         a.z = (e5 && e4) ? e3 : 0u;
         a.w = 0;
         rows_lo[i - prev] = a;
      }
   }

   // Wave-synchronous sliding window over the sorted array: the wave owns sorted entries [c, c+64). At step k lane l
   // looks at entry c+l-1-k, which is what lane l-1 looked at one step earlier: the candidates travel up the lanes with
   // one DPP wave shift per step and enter at lane 0 from a 64-entry vector fetched with one coalesced load per 64 steps.
   // No per-candidate memory access except the LDS probes.
   // The counter is a device-scope atomic (it is served memory-side, microseconds away): each wave asks for its next
   // chunk before it starts on the current one, so the round trip hides behind the scan.
   uint32_t *ctr = chunk_ctr + 2 * seg_id;
   // Chunks are handed out in a scattered order (ticket -> chunk is a bijection on the next power of two): the 6-gram
   // order is sorted by bytes, so the expensive classes (common words) sit next to each other, and in ticket order
   // they would all be met at the same time — or all at the end.
   const uint32_t nchunks = (M + 63) >> 6;
   uint32_t cmask = 1;
   while (cmask < nchunks) cmask <<= 1;
   cmask -= 1;
   uint32_t c_next = 0;
   if (lane == 0) c_next = atomicAdd(ctr, 64u);
   for (;;) {
      const uint32_t ticket = zh_readfirstlane(c_next);
      if (ticket >= (cmask + 1) * 64u) break;
      if (lane == 0) c_next = atomicAdd(ctr, 64u);
      const uint32_t c = ((((ticket >> 6) * 40503u + 12345u) & cmask)) << 6;
      if (c >= M) continue;

      ZH_MF_COUNT(6, 1);
      ZH_MF_TIC();
      const uint32_t t = c + lane;
      const uint32_t own = t < M ? S[t] : ZH_MF_SENTINEL;
      const uint32_t i = own & ZH_MF_POS_MASK;
      const bool mine = t < M && i >= prev;
      const uint32_t maxlen = t < M ? min((uint32_t)ZH_MAX_MATCH, W - i) : 0;   // >= 4
      uint32_t nm = 0;                       // matches accepted so far; the last 8 sit in mring (length | offset<<16)
      uint32_t *myring = mring + threadIdx.x;
      uint32_t cur = ZH_MIN_MATCH - 1;
      bool alive = false;
      // the lane's own first 16 bytes stay in registers: a candidate's match length up to 16 (most are shorter) then costs
      // one round of five LDS reads instead of a dependent pair of reads per four bytes — the walk is where the kernel's
      // time goes (80 % of its wave-cycles on text), and in the walk it is these length computations, serialised over the
      // lanes that have one, that take it (measured: 2000 cycles per walk step)
      uint32_t own16[4] = {0, 0, 0, 0};
      if (mine) {
         if (LDS_WIN)
            zh_load128_at(lwin32, i, own16);
         else
            own16[0] = zh_ld32(win + i);
      }
      uint32_t own32[4] = {0, 0, 0, 0};   // bytes 16..31 likewise (past the window end: garbage, and never decisive — lengths are clamped to maxlen)
      if (mine && LDS_WIN) zh_load128_at(lwin32, i + 16u, own32);
      const uint32_t first4 = own16[0];
      bool has4 = false;
      uint32_t d4 = 0xffffu, d5 = 0xffffu;
      if (mine) {
         // nearest occurrence of the trigram: without it there is no match at all; if its 4th byte differs it is the
         // (only) length-3 entry. Likewise the nearest occurrences of the 4-gram and the 5-gram.
         const uint2 pv = prevs[t];   // the record travels with the entry of the 6-gram order (zh_mf_prev_level)
         const uint32_t d3 = pv.x & 0xffffu, p3 = i - 1u - d3;
         if (d3 != 0xffffu) {
            d4 = pv.x >> 16;
            d5 = pv.y & 0xffffu;
            has4 = d4 != 0xffffu;
            const uint32_t q4 = LDS_WIN ? zh_load32_at(lwin32, p3) : zh_ld32(win + p3);
            if (q4 != first4) {
               myring[0] = ZH_MIN_MATCH | ((i - p3) << 16);
               nm = 1;
               cur = ZH_MIN_MATCH;
            }
         }
      }
      // six equal bytes: the class "cccccc" is not walked, the run table below gives the frontier. (A run that ends within
      // five bytes is an ordinary string: its 6-gram class is walked like any other; taking the run path for it would
      // put a few slow lanes into every chunk of indented text.)
      const bool isrun = first4 == (first4 & 0xffu) * 0x01010101u && mine && win[i + 4] == (first4 & 0xffu) && win[i + 5] == (first4 & 0xffu);
      if (mine && has4 && !isrun) {
         // (every position of the 6-gram order has at least 6 bytes left)
         if (win[i - 1 - d4 + 4] != win[i + 4]) {
            myring[(nm & 7u) * ZH_MF_THREADS] = 4u | ((d4 + 1) << 16);
            nm++;
            cur = 4;
         }
         if (d5 != 0xffffu) {
            if (win[i - 1 - d5 + 5] != win[i + 5]) {
               myring[(nm & 7u) * ZH_MF_THREADS] = 5u | ((d5 + 1) << 16);
               nm++;
               cur = 5;
            }
            alive = !(own & ZH_MF_HEAD);   // a class head has no earlier occurrence of its 6-gram
         }
      }
      ZH_MF_LAP(1);
      // ---- positions with six or more bytes of a byte run ahead: the frontier comes from the run table -----------------
      if (mine && has4 && isrun) {
         alive = false;
         ZH_MF_SUBTIC();
         const uint32_t c = first4 & 0xffu;
         const uint32_t r = LDS_WIN ? zh_run_length(win, i, W, maxlen) : zh_run_length(gwin, i, W, maxlen);   // run bytes left, clamped to maxlen
         // a candidate with r_p run bytes left matches min(r, r_p) bytes unless r_p == r. Nearest first:
         // (1) the own run: every earlier position of it has r_p > r, so they all match r and only the nearest counts
         if (i > 0 && win[i - 1] == c) {
            myring[(nm & 7u) * ZH_MF_THREADS] = r | (1u << 16);
            nm++;
            cur = r;
         }
         if (cur < maxlen) {
            const uint32_t Q = zh_runs_q(W);
            const uint32_t *RS = runs, *RL = runs + Q;
            const uint32_t g0 = runs[2 * Q + c], g1 = runs[2 * Q + 256 + c];
            // own run = last run of this byte value that starts at or before i
            uint32_t a = g0, b = g1;
            while (a < b) {
               const uint32_t mid = (a + b) >> 1;
               if (RS[mid] <= i)
                  a = mid + 1;
               else
                  b = mid;
            }
            // (2) earlier runs of the same byte, nearest first; inside a run the nearer positions have fewer bytes left.
            //     Four table entries are fetched per round trip.
            //     Only until the record reaches r: from then on the second order takes over (below).
            uint32_t j = a > g0 ? a - 1 : g0;
            ZH_MF_SUBLAP(5);
            while (j > g0 && cur < r) {
               const uint32_t nf = min(4u, j - g0);
               uint32_t es[4], ls[4];
#pragma unroll
               for (uint32_t u = 0; u < 4; u++) {
                  const uint32_t jj = j - 1 - min(u, nf - 1);
                  ls[u] = RL[jj];
                  es[u] = RS[jj] + ls[u];   // run [start, e)
               }
               j -= nf;
#pragma unroll
               for (uint32_t u = 0; u < 4; u++) {
                  if (u >= nf || cur >= maxlen) break;
                  const uint32_t e = es[u], len = ls[u];
                  if (i - (e - ZH_RUN_MIN) > ZH_MAX_DIST) {   // its nearest class member is out of reach: so is everything farther
                     j = g0;
                     break;
                  }
                  // positions with cur < r_p < r bytes left: each matches r_p bytes, a new record every time
                  for (uint32_t k = max(cur + 1, (uint32_t)ZH_RUN_MIN); k < r && k <= len; k++) {
                     if (i - (e - k) > ZH_MAX_DIST) break;
                     myring[(nm & 7u) * ZH_MF_THREADS] = k | ((i - (e - k)) << 16);
                     nm++;
                     cur = k;
                  }
                  // the one position with exactly r bytes left: the match continues past the runs
                  if (len >= r) {
                     const uint32_t p = e - r;
                     if (i - p <= ZH_MAX_DIST && (cur < r || win[p + cur] == win[i + cur])) {   // cheap reject: it must agree at byte `cur`
                        uint32_t l = r;
                        if (LDS_WIN) {
                           l = zh_mf_extend_lane(lwin32, p, i, l, maxlen);
                        }
                        else {
                           while (l < maxlen && win[p + l] == win[i + l]) l++;
                        }
                        if (l > cur) {
                           myring[(nm & 7u) * ZH_MF_THREADS] = l | ((i - p) << 16);
                           nm++;
                           cur = l;
                        }
                     }
                  }
               }
            }
            // (3) a match longer than r: only the position with exactly r run bytes left of an earlier run (of at least r bytes)
            //     that is followed by the same byte as this run. Those runs are contiguous in the second order of the table.
            ZH_MF_SUBLAP(7);
            if (cur >= r && cur < maxlen && a > g0) {
               const uint32_t *RS2 = runs + 2 * Q + 513, *RL2 = RS2 + Q;
               const uint32_t X = win[i + r];                              // cur < maxlen: i + r is inside the window
               const uint32_t key = (X << 24) | RS[a - 1];                 // this position's own run
               uint32_t lo = g0, hi = g1;
               while (lo < hi) {
                  const uint32_t mid = (lo + hi) >> 1;
                  if (RS2[mid] < key)
                     lo = mid + 1;
                  else
                     hi = mid;
               }
               ZH_MF_SUBLAP(14);
               uint32_t j2 = lo;                                           // entries below j2: runs that start earlier (or end in a smaller byte)
               // (the table is in HBM/L2: the entries of the round after this one are requested before this round is looked at — on
               // indented source code this loop is two thirds of the byte-run path, hundreds of runs of spaces followed by the same letter)
               uint32_t nss[4], nls[4];
#define ZH_MF_RUNS2_REQUEST()                                    \
   do {                                                          \
      const uint32_t nf_ = min(4u, j2 - g0);                     \
      _Pragma("unroll") for (uint32_t u = 0; u < 4; u++) {       \
         const uint32_t jj = j2 - 1 - min(u, nf_ - 1);           \
         nss[u] = RS2[jj];                                       \
         nls[u] = RL2[jj];                                       \
      }                                                          \
   } while (0)
               if (j2 > g0) ZH_MF_RUNS2_REQUEST();
               while (j2 > g0 && cur < maxlen) {
                  const uint32_t nf = min(4u, j2 - g0);
                  uint32_t ss[4], ls[4];
#pragma unroll
                  for (uint32_t u = 0; u < 4; u++) {
                     ss[u] = nss[u];
                     ls[u] = nls[u];
                  }
                  j2 -= nf;
                  if (j2 > g0) ZH_MF_RUNS2_REQUEST();
#pragma unroll
                  for (uint32_t u = 0; u < 4; u++) {
                     if (u >= nf || cur >= maxlen) break;
                     const uint32_t len = ls[u], e = (ss[u] & 0xffffffu) + len;
                     if ((ss[u] >> 24) != X || i - (e - ZH_RUN_MIN) > ZH_MAX_DIST) {   // the runs ending in X are used up, or out of reach
                        j2 = g0;
                        break;
                     }
                     if (len >= r) {
                        const uint32_t p = e - r;
                        if (i - p <= ZH_MAX_DIST && win[p + cur] == win[i + cur]) {   // cheap reject: it must agree at byte `cur`
                           uint32_t l = r;
                           if (LDS_WIN) {
                              l = zh_mf_extend_lane(lwin32, p, i, l, maxlen);
                           }
                           else {
                              while (l < maxlen && win[p + l] == win[i + l]) l++;
                           }
                           if (l > cur) {
                              myring[(nm & 7u) * ZH_MF_THREADS] = l | ((i - p) << 16);
                              nm++;
                              cur = l;
                           }
                        }
                     }
                  }
               }
#undef ZH_MF_RUNS2_REQUEST
               ZH_MF_SUBLAP(15);
            }
         }
      }
      ZH_MF_LAP(2);
      // what a candidate must match to beat `cur`: the four bytes ending at position max(cur, 3) — for cur <= 3 that is
      // a window inside the first six bytes, which every member of the class shares
      uint32_t fo = max(cur, 3u) - 3u;
      uint32_t ci = (mine && fo) ? (LDS_WIN ? zh_load32_at(lwin32, i + fo) : zh_ld32(win + i + fo)) : first4;

      // The feed's "no entry": marked like a class head, farther than any legal distance from every window position, and — unlike ZH_MF_SENTINEL — a
      // position whose probe is a harmless read inside this kernel's LDS: the probes of a walk step are issued for every lane as it stands
#define ZH_MF_FEED_NONE (ZH_MF_HEAD | (ZH_MF_LDS_WINDOW + 256u))
      static_assert(ZH_MF_LDS_WINDOW + 256u + ZH_MAX_MATCH + 16u < ZH_MF_FRONTIER_LDS, "the feed's no-entry position lies inside the kernel's LDS");
      uint32_t cand = t < M ? own : ZH_MF_FEED_NONE;
      int64_t vbase = (int64_t)c - 64;                            // sorted index of lane 0 of the feed vector
      uint32_t vec = (vbase + lane >= 0) ? S[vbase + lane] : ZH_MF_FEED_NONE;
      int vi = 63;

      // A candidate can only beat `cur` if its bytes fo..fo+3 equal ci: the 4-byte probe weeds out nearly all; the survivors get
      // their true match length from byte 0 (an entry of a neighbouring class met after the class head fails there) — the
      // first 16 bytes per lane against the lane's own 16 in registers, anything longer by the whole wave at once
      // (zh_mf_extend_wave). The kernel is bound by the instructions it issues (round 5: fifteen more per step made it 8 % slower; its probes'
      // bank conflicts in LDS do not matter), so a step is written for few of them:
      //  * the probes are issued for every lane as it stands — a dead lane's, or one whose candidate is out of reach, is a harmless read —
      //    instead of steering the lanes that should not probe to address 0;
      //  * the lanes verify their survivors by RANK, not by candidate: in a round every lane with a survivor left takes its nearest one.
      //    A step has a survivor in two or three of its four candidates (some lane's), but a lane rarely has more than one: rounds 1-4 ran the
      //    verification once per candidate with a survivor in any lane, each time for a lane or two.
      // A probe taken before `cur` grew stays a valid pre-filter.
      while (const uint64_t alive_mask_ = zh_ballot(alive)) {
         ZH_MF_WALK_TOP(alive_mask_);
         // advance ZH_MF_STEP times: entries c+l-1-Nk .. c+l-N-Nk arrive at lane l, nearest first; lane 0 takes the next entries below the chunk
         uint32_t cs[ZH_MF_STEP], qs[ZH_MF_STEP], pbs[ZH_MF_STEP];
#pragma unroll
         for (int k = 0; k < (int)ZH_MF_STEP; k++) {
            cand = zh_wave_shr1(cand, zh_readlane(vec, vi - k));
            cs[k] = cand;
         }
         vi -= (int)ZH_MF_STEP;
         if (vi < 0) {
            vbase -= 64;
            vec = (vbase + lane >= 0) ? S[vbase + lane] : ZH_MF_FEED_NONE;
            vi = 63;
         }
         // all the 4-byte probes of the step are issued before any is used
#pragma unroll
         for (int k = 0; k < (int)ZH_MF_STEP; k++) {
            qs[k] = cs[k] & ZH_MF_POS_MASK;
            pbs[k] = LDS_WIN ? zh_load32_at(lwin32, qs[k] + fo) : zh_ld32(win + ((alive && i - qs[k] <= ZH_MAX_DIST) ? qs[k] + fo : 0u));
         }
         // the lane's survivors, nearest first: bit k of pend
         uint32_t pend = 0, heads = 0;
#pragma unroll
         for (int k = 0; k < (int)ZH_MF_STEP; k++) {
            pend |= (alive && i - qs[k] <= ZH_MAX_DIST && pbs[k] == ci) ? (1u << k) : 0u;   // (out of reach: false for the feed's no-entry too)
            heads |= cs[k];
         }
         // a class is ascending in position and walked downwards: while no head has been met, the step's last candidate is its farthest
         const bool inreach = i - qs[ZH_MF_STEP - 1u] <= ZH_MAX_DIST;
         ZH_MF_WALK_PROBED();
         if (zh_ballot(pend != 0u)) {
            bool moved = false;   // a record of this step moved `cur`: the probes above were taken at the old one (still a valid pre-filter)
            do {
               const bool v = pend != 0u;
               const uint32_t kk = v ? (uint32_t)__builtin_ctz(pend) : 0u;
               pend &= pend - 1u;
               uint32_t q = qs[0];
#pragma unroll
               for (int k = 1; k < (int)ZH_MF_STEP; k++) q = kk == (uint32_t)k ? qs[k] : q;
               uint32_t f[4];
               if (LDS_WIN)
                  zh_load128_at(lwin32, v ? q : 0u, f);
               else {
#pragma unroll
                  for (uint32_t u = 0; u < 4; u++) f[u] = zh_ld32(win + (v ? q : 0u) + 4u * u);
               }
               uint32_t l = zh_mf_len16(f, own16);   // (bytes past the window end are garbage, but the lengths are clamped to maxlen)
               // a candidate has to beat the record as it stands now. Sixteen bytes or fewer: its length is known. More: where the
               // record has moved since the probe, four bytes ending at the new record are probed first
               bool deep = v && l == 16 && maxlen > 16 && cur < maxlen;
               if (zh_ballot(deep && moved)) {
                  if (deep && moved) {
                     const uint32_t fn = cur - 3u;
                     deep = LDS_WIN ? zh_load32_at(lwin32, q + fn) == zh_load32_at(lwin32, i + fn) : zh_ld32(win + q + fn) == zh_ld32(win + i + fn);
                  }
               }
               ZH_MF_WALK_EXT(deep);
               if (LDS_WIN)
                  l = zh_mf_length_past16(lwin32, deep, q, i, maxlen, own32, l);
               else if (deep)
                  while (l < maxlen && win[q + l] == win[i + l]) l++;
               l = min(l, maxlen);
               const bool rec = v && l > cur && (l < 16u || deep || maxlen <= 16u);
               if (rec) {
                  myring[(nm & 7u) * ZH_MF_THREADS] = l | ((i - q) << 16);   // offset 32768 needs all 16 bits
                  nm++;
                  cur = l;
               }
               moved = moved || rec;
            } while (zh_ballot(pend != 0u));
            if (moved && cur < maxlen) {
               fo = cur - 3;   // (a record of the class walk is at least 6)
               ci = LDS_WIN ? zh_load32_at(lwin32, i + fo) : zh_ld32(win + i + fo);
            }
            ZH_MF_WALK_VERIFIED();
         }
         // the class ends at its head; beyond 32 KiB everything else is farther still; 258 (or the window end) cannot be beaten
         alive = alive && inreach && !(heads & ZH_MF_HEAD) && cur < maxlen;
      }
#undef ZH_MF_FEED_NONE
      ZH_MF_LAP(3);
      if (mine) {
         // rows are longest first: the ring read backwards from the last accepted match
         uint32_t m[8];
#pragma unroll
         for (uint32_t k = 0; k < 8; k++) m[k] = (k < nm) ? myring[((nm - 1 - k) & 7u) * ZH_MF_THREADS] : 0u;
         uint4 a, b2;
         a.x = m[0]; a.y = m[1]; a.z = m[2]; a.w = m[3];
         b2.x = m[4]; b2.y = m[5]; b2.z = m[6]; b2.w = m[7];
         rows_lo[i - prev] = a;
         if (nm >= 4) rows_hi[i - prev] = b2;   // readers fetch it whenever slot 3 holds a match
      }
      ZH_MF_LAP(4);
   }

   ZH_MF_PROF_FLUSH();
   __syncthreads();   // every wave is done with the window in LDS
   }
}
