// zh_parse_chain.h — the optimal parse (zh_parse.h) of a barrier-free run as ONE chain: repeated boilerplate, near-copies of
// earlier data, constant padding — a 64 KiB max-block of such data can be a single run, and then the recurrence
// cost[p] <- cost[p+1 .. p+258] (reference src/blockdeflate.c:254-323) is the whole run time of the batch. One wave alone on a
// SIMD issues an instruction every 4-5 cycles whatever it is (measured, tools/probes), so what decides the speed of the chain
// is the instruction count of one position's step on the wave that carries the recurrence — everything that does not depend
// on the costs is done by other waves of the workgroup.
//
// Workgroup = 4 waves, one chain at a time:
//   producers (waves 1, 2)  digest the match rows of the NEXT tile of 32 positions (loads issued three tile periods earlier) and write
//                           one 32-bit descriptor per (position, candidate): price of the candidate in the bits the costs are
//                           kept in, below it the (slot, length) bits that break ties the way the reference's evaluation order
//                           does, and — for the slots stored with length >= 40, tried at full length only
//                           (blockdeflate.c:286-297) — above it the cost-ring entry that holds cost[p + length];
//   flusher   (wave 3)      turns the winning keys of the PREVIOUS tile into parse entries and stores them (the rows it needs for
//                           that were left in LDS by the producers);
//   consumer  (wave 0)      one position per step, candidate = lane: lane j < 37 prices length 3+j, lanes 40..47 the long slots.
//                           The costs live in REGISTERS across the lanes: lane j holds cost[p+3+j], and moving to p-1 is one DPP
//                           wave shift with cost[p+2] — known two steps earlier — entering at lane 0. So a step is: one LDS read
//                           (descriptor), one LDS gather for the long slots, select, add, a wave-wide minimum (six DPP steps and a
//                           readlane), and the scalar literal-or-match decision: ~22 instructions, none of which touches the cost
//                           ring on the way from cost[p+1] to cost[p]. (The previous kernel priced three positions per step with the
//                           costs in an LDS ring: 185 cycles per position, 40 % of them LDS traffic for the ring.)
// Costs are kept as cost << 9 in 32 bits, absolute from the chain's end; every tile the consumer dumps its 32 new costs into
// a 512-entry LDS ring for the long slots (whose targets are >= 40 positions away, i.e. always in an earlier tile) and, when the
// costs approach 2^30, subtracts a common base from registers and ring. Ring entry i stores cost - (i << 23): a long
// descriptor carries its ring index in bits 23..31, where it doubles as the address of the gather (descriptor >> 21) and
// cancels out when descriptor and gathered value are added.
#pragma once
#include "zh_parse.h"

#define ZH_CHAIN_THREADS 256
#define ZH_CHAIN_TILE 32u
#define ZH_CHAIN_RING 512u
// price field of a candidate that does not exist: above every real difference between two costs at most 258 positions apart
// (258 x 15 bits) plus any price, so it never beats the literal — and its bits 21..22 are clear (the gather address stays aligned)
#define ZH_CHAIN_NOPRICE (4095u << 9)
#define ZH_CHAIN_LONG0 40u           // consumer lane of long slot 0
#define ZH_CHAIN_REBASE (1u << 30)
#define ZH_CHAIN_VLONG_TASK 24576u  // positions: listed tasks longer than this get the first tickets ...
#define ZH_CHAIN_LONG_TASK 6144u    // ... those longer than this the next ones

// ---- segments of cut tasks (see "speculative segments", zh_parse.h) as chain jobs ---------------------------------------------
// When a run has few cut tasks their segments are latency, not throughput: a four-wave chain workgroup prices a position in
// 0.054 us, a 16-lane row of the segment workgroups (zh_parse.h) in about 0.3 us. The host then hands the segments to zh_parse_chain (a segment = a job
// with a made-up end, a limit for its parse entries and vectors to record); the workgroup that finishes a task's last
// segment checks the task (zh_chain_check_task).
#ifndef ZH_TRACE_SLOTS
#define ZH_TRACE_SLOTS 4096u
#endif

struct zh_chain_job_t {
   uint32_t t0, t1;        // positions to price: [t0, t1), from t1 - 1 down
   uint32_t clamp;         // candidates end here at the latest (the sub-block end, or the made-up end of a speculative start)
   uint32_t store_hi;      // parse entries of positions >= store_hi are not stored (a tile boundary: t1 - store_hi is a multiple of the tile)
   const int16_t *import;  // relative costs of [t1, t1 + 258], or NULL: cost[t1] = 0 and nothing reaches beyond
   int16_t *export_spec;   // relative costs of [store_hi, store_hi + 258], written when the recurrence passes store_hi, or NULL
   int16_t *export_left;   // relative costs of [t0, t0 + 258] at the end (t1 - t0 a multiple of the tile), or NULL
};

#ifdef ZH_CHAIN_PROFILE
__device__ uint64_t zh_chain_profile[4];   // probe builds only (tools/probes/chain2_probe.hip): busy cycles per role of the last chain
#endif

struct zh_chain_ws_t {
   union {
      struct {
         uint32_t ring[ZH_CHAIN_RING];                    // entry i: (cost << 9) - (i << 23) of the position p with p % 512 == i
         uint32_t desc[2][ZH_CHAIN_TILE][65];             // [tile parity][position, top first][consumer lane]; the odd row stride spreads
                                                          // the producers' stores (16 positions per store instruction) over the LDS banks
         uint32_t lit[2][ZH_CHAIN_TILE];                  // literal price << 9 (0 for positions below the range)
         uint32_t bt[2][ZH_CHAIN_TILE];                   // winning key per position; low 9 bits zero = literal
         uint32_t raw[4][ZH_CHAIN_TILE][9];               // [tile % 4]: the positions' match rows as staged (8 slots; the odd stride spreads
                                                          // the banks), for the flusher two tile periods later
      } p;
      uint32_t hist[ZH_NSYM];                             // after the parse: histogram of the range
   };
   uint8_t litprice[ZH_NLIT];
   uint8_t lencost[256];
   uint8_t distcost[ZH_NDIST];
};

// producer thread pl = 4 j + part: requests the row of position thi-1-j of the tile [thi - cnt, thi) — both planes
// unconditionally (the second one holds stale bytes unless slot 3 is a match: zh_chain_stage masks it) and the byte at the
// position. The loads stay in flight for three tile periods (zh_async_load_row: completion tracked by the caller).
__device__ __forceinline__ void zh_chain_fetch(zh_async_row_t &f, const uint4 *rows, const uint4 *rows_hi, const uint8_t *win, uint32_t prev, uint32_t thi,
                                               uint32_t cnt, uint32_t pl) {
   const uint32_t j = pl >> 2;
   const uint32_t pos = j < cnt ? thi - 1 - j : thi - 1;   // clamped: the loads are always legal
   zh_async_load_row(f, rows + (pos - prev), rows_hi + (pos - prev), win + pos);
}

// producer: stage tile [thi - cnt, thi) into parity `buf`. Thread pl = 4 j + part serves position thi-1-j:
//   part 0: consumer lanes 0..15 (lengths 3..18)      part 1: lanes 16..31 (lengths 19..34)
//   part 2: lanes 32..39 (lengths 35..39, three unused) and 40..47 (the slots stored with length >= 40)
//   part 3: the literal's price and the raw row for the flusher
// lcw = the 16 length prices of the thread's part (lencost[16 part ..], four per word), loaded once per chain: a byte read
// from LDS per descriptor, each waited for, was most of this function's time.
__device__ __forceinline__ void zh_chain_stage(zh_chain_ws_t &ws, uint32_t buf, uint32_t rbuf, const zh_async_row_t &f, uint32_t thi, uint32_t cnt, uint32_t sb_end,
                                               uint32_t pl, const uint32_t (&lcw)[4]) {
   const uint32_t j = pl >> 2, part = pl & 3u;
   uint32_t *d = &ws.p.desc[buf][j][16u * part];
   const bool ok = j < cnt;
   const uint32_t pos = ok ? thi - 1 - j : thi - 1;
   const uint32_t room = sb_end - pos;
   const bool more = ok && (f.a.w & 0xffffu) >= ZH_MIN_MATCH;   // slots 4..7 exist only behind a full first plane (zh_common.h)
   const uint32_t raw[ZH_NMATCH] = {ok ? f.a.x : 0u, ok ? f.a.y : 0u, ok ? f.a.z : 0u, ok ? f.a.w : 0u, more ? f.b.x : 0u, more ? f.b.y : 0u, more ? f.b.z : 0u, more ? f.b.w : 0u};
   if (part == 3) {
#pragma unroll
      for (uint32_t m = 0; m < ZH_NMATCH; m++) ws.p.raw[rbuf][j][m] = raw[m];
      ws.p.lit[buf][j] = ok ? (uint32_t)ws.litprice[f.byte & 0xffu] << 9 : 0u;
      return;
   }
   // the same digest as zh_stage_position: bitmap of short slot lengths, running minima of (distance price, slot)
   uint32_t nlong = 0, nshort = 0, kmax = 0, run = 0xFFu;
   uint64_t pm = 0, lmask = 0;
   uint32_t oc[ZH_NMATCH];
#pragma unroll
   for (uint32_t m = 0; m < ZH_NMATCH; m++) oc[m] = (uint32_t)ws.distcost[zh_dist_sym((raw[m] & 0xffffu) >= ZH_MIN_MATCH ? raw[m] >> 16 : 1u)];
#pragma unroll
   for (uint32_t m = 0; m < ZH_NMATCH; m++) {
      const uint32_t len = raw[m] & 0xffffu;
      const bool valid = len >= ZH_MIN_MATCH;
      const bool is_long = len >= ZH_LEAVE_ALONE;
      const bool is_short = valid && !is_long;
      nlong += is_long ? 1u : 0u;
      kmax = max(kmax, is_short ? len : 0u);               // the first short slot is the longest
      lmask |= is_short ? (1ull << (len - ZH_MIN_MATCH)) : 0ull;
      run = is_short ? min(run, (oc[m] << 3) | m) : run;
      pm |= is_short ? ((uint64_t)run << (8 * nshort)) : 0ull;
      nshort += is_short ? 1u : 0u;
   }
   kmax = min(kmax, room);                                  // end clamp (blockdeflate.c:283-284)
   // lengths k = 3 + 16 part + s: the last short slot reaching k has the cheapest distance among those that can provide it
   const uint32_t e0 = 16u * part;
   const uint32_t w = (uint32_t)(lmask >> e0);                              // lengths e0+3 .. e0+34
   const uint32_t above = (uint32_t)__popcll(lmask >> (e0 + 16u) >> 16u);   // slots longer than that window
   const uint32_t pm_lo = (uint32_t)pm, pm_hi = (uint32_t)(pm >> 32);
#pragma unroll
   for (uint32_t s = 0; s < 16; s++) {
      if (part == 2 && s >= 8) break;   // (uniform per thread; lanes 40..47 follow)
      const uint32_t k = ZH_MIN_MATCH + e0 + s;
      const uint32_t sel = (uint32_t)__popc(w >> s) + above - 1u;
      const uint32_t bb = ((sel < 4 ? pm_lo : pm_hi) >> ((sel & 3u) * 8u)) & 0xffu;
      const uint32_t price = ((lcw[s >> 2] >> (8u * (s & 3u))) & 0xffu) + (bb >> 3);
      const uint32_t v = (price << 9) | (((bb & 7u) << 6) + (40u - k));   // tie bits: (slot, 39 - k) + 1, the literal's are 0
      d[s] = (k <= kmax && k < ZH_LEAVE_ALONE) ? v : ZH_CHAIN_NOPRICE;
   }
   if (part == 2) {
      // the slots stored with length >= 40: tried at their full (clamped) length only (blockdeflate.c:286-297)
      uint32_t lc[ZH_NMATCH];
#pragma unroll
      for (uint32_t s = 0; s < ZH_NMATCH; s++) {
         uint32_t enc = min(raw[s] & 0xffffu, room) - ZH_MIN_MATCH;   // wraps below 3, then saturates (:289, :216-219)
         if (enc > 255) enc = 255;
         lc[s] = ws.lencost[enc];
      }
#pragma unroll
      for (uint32_t s = 0; s < ZH_NMATCH; s++) {
         const uint32_t mlen = min(raw[s] & 0xffffu, room);
         const uint32_t v = (((pos + mlen) & (ZH_CHAIN_RING - 1u)) << 23) | ((lc[s] + oc[s]) << 9) | ((s << 6) + 1u);
         // a slot that does not exist gathers the cost of the tile's top (dumped a tile ago, within 32 positions of p+1): harmless
         d[8u + s] = s < nlong ? v : (((thi & (ZH_CHAIN_RING - 1u)) << 23) | ZH_CHAIN_NOPRICE);
      }
   }
}

// flusher: the winning keys of a priced tile -> parse entries (zh_decode_pick). Thread i = position thi-1-i; the position's match
// row comes from LDS, where the stager left it (a global load here would put HBM latency into every tile period).
__device__ __forceinline__ void zh_chain_flush(zh_chain_ws_t &ws, uint32_t buf, uint32_t rbuf, uint32_t prev, uint32_t thi, uint32_t cnt, uint32_t sb_end, uint32_t i,
                                               uint32_t *best, uint32_t store_hi) {
   if (i >= cnt || thi > store_hi) return;   // (store_hi is a tile boundary)
   const uint32_t pos = thi - 1 - i;
   const uint32_t kk = ws.p.bt[buf][i];
   uint32_t pick = 0;
   if (kk & 511u) {   // low 9 bits: 0 = literal, else (slot << 6 | 39 - length) + 1
      const uint32_t *raw = ws.p.raw[rbuf][i];
      uint32_t nlong = 0;
#pragma unroll
      for (uint32_t q = 0; q < ZH_NMATCH; q++) nlong += (raw[q] & 0xffffu) >= ZH_LEAVE_ALONE ? 1u : 0u;
      const uint32_t tb = (kk & 511u) - 1u;
      const uint32_t m = tb >> 6;
      const uint32_t e = raw[m];
      const uint32_t len = (m < nlong) ? min(e & 0xffffu, sb_end - pos) : (39u - (tb & 63u));
      pick = len | (e & 0xffff0000u);
   }
   best[pos - prev] = pick;
}

// consumer: the recurrence over one tile, one position per step, always ZH_CHAIN_TILE steps (positions below the range's start
// are staged as "literal of price 0, no candidates": their costs are computed and never looked at). State:
//   cv  lane j: cost9[p + 3 + j] for the position p priced next        c1, c2 (wave-uniform): cost9[p + 1], cost9[p + 2]
// The best match of a position depends on costs from cost[p+3] on, i.e. on what was known three steps earlier: its wave-wide
// minimum (six dependent DPP steps) runs in the shadow of the two steps before it. What carries from one step to the next is
// the literal alone: cost[p] = min(cost[p+1] + literal, best match) — an add, a min, a mask. A match's key has tie bits >= 1 in
// its low 9 bits, the literal's are 0: "literal first; a match must be strictly cheaper" (:292,:307) is that same minimum.
struct zh_chain_state_t {
   uint32_t cv, c1, c2;
};

__device__ __forceinline__ void zh_chain_consume(zh_chain_ws_t &ws, uint32_t buf, uint32_t thi, zh_chain_state_t &st) {
   const uint32_t lane = zh_lane();
   const bool is_long = lane >= ZH_CHAIN_LONG0 && lane < ZH_CHAIN_LONG0 + ZH_NMATCH;
   const uint32_t *dp = &ws.p.desc[buf][0][lane];
   const uint8_t *ring8 = (const uint8_t *)ws.p.ring;
   const uint32_t litv = ws.p.lit[buf][lane & (ZH_CHAIN_TILE - 1u)];
   uint32_t wv = 0;
   uint32_t cv = st.cv, c1 = st.c1, c2 = st.c2;
   // Positions go in groups of three: the three wave-wide minima of a group only need costs from before the group (position
   // p-2 looks no nearer than cost[p+1]), so their DPP steps are interleaved; then the three literal decisions chain.
   uint32_t d[3], g[3];
#pragma unroll
   for (uint32_t q = 0; q < 3; q++) {
      d[q] = dp[65u * q];
      g[q] = *(const uint32_t *)(ring8 + (d[q] >> 21));
   }
#pragma unroll
   for (uint32_t t = 0; t < ZH_CHAIN_TILE; t += 3) {
      // (the tile has 32 positions: the last group is a pair; its third member prices a copy of the tile's last row and is dropped)
      uint32_t dn[3], gn[3];
#pragma unroll
      for (uint32_t q = 0; q < 3; q++) {
         dn[q] = dp[65u * min(t + 3u + q, ZH_CHAIN_TILE - 1u)];
         gn[q] = *(const uint32_t *)(ring8 + (dn[q] >> 21));
      }
      const uint32_t cva = cv, cvb = zh_wave_shr1(cva, c2), cvc = zh_wave_shr1(cvb, c1);
      uint32_t ka = (is_long ? g[0] : cva) + d[0], kb = (is_long ? g[1] : cvb) + d[1], kc = (is_long ? g[2] : cvc) + d[2];
      zh_wave_min3_lane63(ka, kb, kc);
      const uint32_t ma = min(c1 + zh_readlane(litv, (int)t), zh_readlane(ka, 63));
      const uint32_t ca = ma & ~511u;
      const uint32_t mb = min(ca + zh_readlane(litv, (int)min(t + 1u, ZH_CHAIN_TILE - 1u)), zh_readlane(kb, 63));
      const uint32_t cb = mb & ~511u;
      wv = zh_wave_shr1(zh_wave_shr1(wv, ma), mb);   // the winning keys travel up the lanes like the costs: lane i = step 31 - i
      if (t + 2 < ZH_CHAIN_TILE) {
         const uint32_t mc = min(cb + zh_readlane(litv, (int)min(t + 2u, ZH_CHAIN_TILE - 1u)), zh_readlane(kc, 63));
         const uint32_t cc = mc & ~511u;
         wv = zh_wave_shr1(wv, mc);
         cv = zh_wave_shr1(cvc, ca);
         c2 = cb;
         c1 = cc;
      }
      else {
         cv = cvc;
         c2 = ca;
         c1 = cb;
      }
#pragma unroll
      for (uint32_t q = 0; q < 3; q++) {
         d[q] = dn[q];
         g[q] = gn[q];
      }
   }
   // the tile's winners for the flusher, its costs for the long slots of later tiles: cost9[thi - 32 + i] is c1, c2, then cv from lane 0 on
   if (lane < ZH_CHAIN_TILE) ws.p.bt[buf][ZH_CHAIN_TILE - 1u - lane] = wv;
   const uint32_t dv = zh_wave_shr1(zh_wave_shr1(cv, c2), c1);
   if (lane < ZH_CHAIN_TILE) {
      const uint32_t idx = (thi - ZH_CHAIN_TILE + lane) & (ZH_CHAIN_RING - 1u);
      ws.p.ring[idx] = dv - (idx << 23);
   }
   st.cv = cv;
   st.c1 = c1;
   st.c2 = c2;
}

// every cost the consumer holds — registers and ring — minus a common base, before the 32-bit keys could wrap
__device__ __forceinline__ void zh_chain_rebase(zh_chain_ws_t &ws, zh_chain_state_t &st) {
   const uint32_t lane = zh_lane();
   const uint32_t delta = (st.c1 - (1u << 24)) & ~511u;   // the live costs lie within 258 x 15 bits of c1: none goes negative
   st.cv -= delta;
   st.c1 -= delta;
   st.c2 -= delta;
   for (uint32_t k = lane; k < ZH_CHAIN_RING; k += 64) ws.p.ring[k] -= delta;
}

// consumer, right after a tile's costs went into the ring: the costs of [x, x + 258] relative to cost[x], x = the tile's low end
__device__ __forceinline__ void zh_chain_export(zh_chain_ws_t &ws, int16_t *out, uint32_t x) {
   const uint32_t i0 = x & (ZH_CHAIN_RING - 1u);
   const uint32_t base = ws.p.ring[i0] + (i0 << 23);
   for (uint32_t i = zh_lane(); i < ZH_VEC_LIVE + 1u; i += 64) {   // (an even number of entries: the vectors are compared word-wise)
      const uint32_t idx = (x + i) & (ZH_CHAIN_RING - 1u);
      out[i] = i < ZH_VEC_LIVE ? (int16_t)((int32_t)(ws.p.ring[idx] + (idx << 23) - base) >> 9) : (int16_t)0;
   }
}

// Parse job.t0 .. job.t1 of a sub-block as one chain. All ZH_CHAIN_THREADS threads call. ws.litprice / lencost / distcost hold the
// prices of the pass.
__device__ inline void zh_chain_parse(zh_chain_ws_t &ws, const uint4 *rows, const uint4 *rows_hi, const uint8_t *win, uint32_t prev, const zh_chain_job_t &job,
                                      uint32_t *best) {
   const uint32_t t0 = job.t0, t1 = job.t1, sb_end = job.clamp;
   const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
   const uint32_t ntiles = t1 > t0 ? (t1 - t0 + ZH_CHAIN_TILE - 1) / ZH_CHAIN_TILE : 0u;
   if (!ntiles) return;
   const bool stager = wave == 1 || wave == 2;
   const uint32_t pl = tid - 64;   // stagers: 0..127
   // the consumer takes a tile in about a microsecond; the stagers' row loads are therefore issued three tile periods ahead
   // (HBM latency under load is two of them)
   uint32_t lcw[4] = {0, 0, 0, 0};
   if (stager) {
#pragma unroll
      for (uint32_t q = 0; q < 4; q++) lcw[q] = *(const uint32_t *)&ws.lencost[16u * (pl & 3u) + 4u * q];
   }
   zh_async_row_t fr[3];   // slot t % 3: the rows of tile t
   // wait until at most y_ row requests (of ZH_ASYNC_ROW_LOADS loads each) are still in flight
#define ZH_CHAIN_AWAIT(y_)                                                \
   do {                                                                   \
      const uint32_t y__ = (y_);                                          \
      if (y__ >= 3u)                                                      \
         zh_async_wait<3 * ZH_ASYNC_ROW_LOADS>();                         \
      else if (y__ == 2u)                                                 \
         zh_async_wait<2 * ZH_ASYNC_ROW_LOADS>();                         \
      else if (y__ == 1u)                                                 \
         zh_async_wait<1 * ZH_ASYNC_ROW_LOADS>();                         \
      else                                                                \
         zh_async_wait<0>();                                              \
   } while (0)
#define ZH_CHAIN_TILE_HI(k_) (t1 - (k_) * ZH_CHAIN_TILE)
#define ZH_CHAIN_TILE_CNT(k_) min(ZH_CHAIN_TILE, ZH_CHAIN_TILE_HI(k_) - t0)
   if (stager) {
#pragma unroll
      for (uint32_t q = 0; q < 3; q++)
         if (q < ntiles) zh_chain_fetch(fr[q], rows, rows_hi, win, prev, ZH_CHAIN_TILE_HI(q), ZH_CHAIN_TILE_CNT(q), pl);
      ZH_CHAIN_AWAIT(min(ntiles, 3u) - 1u);   // tile 0 has landed; up to two younger requests stay in flight
      zh_async_landed(fr[0]);
      zh_chain_stage(ws, 0, 0, fr[0], t1, ZH_CHAIN_TILE_CNT(0u), sb_end, pl, lcw);
   }
   else if (wave == 0) {
      // cost[t1] = 0; entries above it are never asked for (t1 is a barrier or the end: no candidate reaches past it) — or the
      // costs a neighbouring segment left for [t1, t1 + 258]
      for (uint32_t k = lane; k < ZH_CHAIN_RING; k += 64) {
         const uint32_t i = (k - t1) & (ZH_CHAIN_RING - 1u);
         const uint32_t v = (job.import && i < ZH_VEC_LIVE) ? (uint32_t)(ZH_VEC_BIAS + (int32_t)job.import[i]) << 9 : 0u;
         ws.p.ring[k] = v - (k << 23);
      }
   }
   else {
      // consumer lanes 48..63 price nothing
      for (uint32_t k = lane; k < 2u * ZH_CHAIN_TILE * 16u; k += 64) ws.p.desc[k / (ZH_CHAIN_TILE * 16u)][(k / 16u) % ZH_CHAIN_TILE][48u + (k & 15u)] = ZH_CHAIN_NOPRICE;
   }
   __syncthreads();
   // the recurrence is the critical path of the whole batch: its wave outranks the throughput waves it shares a SIMD with
   // (zh_parse_lanes of the same pass, the other run's kernels); the producers must keep up with it
   if (wave == 0)
      zh_set_wave_priority_high();
   else
      zh_set_wave_priority_mid();

#ifdef ZH_CHAIN_PROFILE
   uint64_t busy = 0, tic = 0;
#define ZH_CHAIN_TIC() tic = zh_clock()
#define ZH_CHAIN_TOC() busy += zh_clock() - tic
#else
#define ZH_CHAIN_TIC()
#define ZH_CHAIN_TOC()
#endif
   // One loop per role, one workgroup barrier per tile period in each (every wave meets the same number of barriers).
   if (wave == 0) {
      zh_chain_state_t st;
      st.cv = 0;   // lane j: cost9[t1 + 2 + j]: beyond the end; lane 0 receives cost9[t1] = 0 when position t1-2 is priced
      st.c1 = 0;   // cost9[t1]
      st.c2 = 0;
      if (job.import) {
         st.c1 = (uint32_t)(ZH_VEC_BIAS + (int32_t)job.import[0]) << 9;
         st.c2 = (uint32_t)(ZH_VEC_BIAS + (int32_t)job.import[1]) << 9;
         st.cv = (uint32_t)(ZH_VEC_BIAS + (int32_t)job.import[2 + lane]) << 9;
      }
      for (uint32_t k = 0; k < ntiles; k++) {
         ZH_CHAIN_TIC();
         if (st.c1 >= ZH_CHAIN_REBASE) zh_chain_rebase(ws, st);
         zh_chain_consume(ws, k & 1u, ZH_CHAIN_TILE_HI(k), st);
         if (job.export_spec && ZH_CHAIN_TILE_HI(k) - ZH_CHAIN_TILE == job.store_hi) zh_chain_export(ws, job.export_spec, job.store_hi);
         ZH_CHAIN_TOC();
         zh_sync_lds();
      }
      if (job.export_left) zh_chain_export(ws, job.export_left, t0);
   }
   else if (stager) {
      // period k: the rows of tile k+3 are requested into the slot tile k occupied, then tile k+1 (requested two periods ago)
      // is staged. The slots are addressed statically (three periods per loop iteration): rotating them through register copies
      // would make every period wait for ALL outstanding loads, the youngest included.
      for (uint32_t k3 = 0; k3 < ntiles; k3 += 3) {
#pragma unroll
         for (uint32_t q = 0; q < 3; q++) {
            const uint32_t k = k3 + q;
            if (k < ntiles) {
               ZH_CHAIN_TIC();
               if (k + 3 < ntiles) zh_chain_fetch(fr[q], rows, rows_hi, win, prev, ZH_CHAIN_TILE_HI(k + 3), ZH_CHAIN_TILE_CNT(k + 3), pl);
               if (k + 1 < ntiles) {
                  ZH_CHAIN_AWAIT(min(ntiles - 1u, k + 3u) - (k + 1u));   // requests younger than tile k+1's: tiles k+2 .. min(k+3, last)
                  zh_async_landed(fr[(q + 1) % 3u]);
                  zh_chain_stage(ws, (k + 1) & 1u, (k + 1) & 3u, fr[(q + 1) % 3u], ZH_CHAIN_TILE_HI(k + 1), ZH_CHAIN_TILE_CNT(k + 1), sb_end, pl, lcw);
               }
               ZH_CHAIN_TOC();
               zh_sync_lds();
            }
         }
      }
   }
   else {
      for (uint32_t k = 0; k < ntiles; k++) {
         ZH_CHAIN_TIC();
         if (k) zh_chain_flush(ws, (k - 1) & 1u, (k - 1) & 3u, prev, ZH_CHAIN_TILE_HI(k - 1), ZH_CHAIN_TILE, sb_end, lane, best, job.store_hi);
         ZH_CHAIN_TOC();
         zh_sync_lds();
      }
      const uint32_t k = ntiles - 1;
      zh_chain_flush(ws, k & 1u, k & 3u, prev, ZH_CHAIN_TILE_HI(k), ZH_CHAIN_TILE_CNT(k), sb_end, lane, best, job.store_hi);
   }
#ifdef ZH_CHAIN_PROFILE
   if (lane == 0) zh_chain_profile[wave] = busy;
#endif
#undef ZH_CHAIN_TIC
#undef ZH_CHAIN_TOC
   zh_set_wave_priority_normal();
#undef ZH_CHAIN_AWAIT
#undef ZH_CHAIN_TILE_HI
#undef ZH_CHAIN_TILE_CNT
}

// the tasks zh_parse_lanes leaves alone: one wave per task, the same piece computation. cnt = the run's counters (ZH_CNT_*).
// Tasks of at least seg_min positions that are not periodic are cut into segments (zh_parse.h): listed in segtasks, with one
// entry of segitems per segment (for zh_parse_chain) and one entry of segwaves per four segments (for the segment workgroups of zh_parse_lanes' launch): the
// host picks one of the two ways by the number of segments in the run.
__device__ __forceinline__ void zh_list_huge_one(uint32_t *bnd /* LDS, ZH_MAXPIECES + 1 */, uint32_t gt, const zh_block_t *__restrict__ blocks, const uint64_t *__restrict__ bars, uint64_t bar_stride,
                                                 const zh_work_t *__restrict__ work, const uint2 *__restrict__ taskmap, const uint32_t *__restrict__ longest, uint64_t longest_stride,
                                                 uint32_t *hugelist, uint32_t cap, uint4 *segtasks, uint2 *segitems, uint2 *segwaves, uint32_t seg_min, uint32_t cut_len, uint32_t *cnt,
                                                 uint2 *taskinfo, uint32_t coop_min) {
   const uint2 tm = taskmap[gt];
   const zh_work_t wk = work[tm.x];
   const uint32_t prev = blocks[wk.block].prev;
   const uint64_t *bar = bars + (uint64_t)wk.block * bar_stride;
   const uint32_t lane = zh_lane(), sb_end = wk.start + wk.size;
   const uint32_t t0 = zh_task_boundary(bar, prev, wk.start, sb_end, tm.y, wk.ntasks);
   const uint32_t t1 = zh_task_boundary(bar, prev, wk.start, sb_end, tm.y + 1, wk.ntasks);
   const uint32_t np = zh_task_pieces(bnd, bar, prev, t0, t1, lane);
   zh_sync();
   const bool huge = zh_task_is_huge(bnd, np, lane, coop_min);
   // the task's range and whether it is listed here, for the four passes of zh_parse_lanes: looking a boundary up means scanning the
   // barrier bitmap, 64 positions per dependent load — on data with few barriers that was a third of that kernel's time
   if (lane == 0) taskinfo[gt] = make_uint2(t0, t1 | (huge ? 0x80000000u : 0u));
   if (!huge) return;
   const uint32_t len = t1 - t0;
   if (len >= seg_min) {
      // periodic? (a 258-byte match at most positions: the costs 258 apart copy each other, a speculative start never converges)
      const uint32_t *lg = longest + (uint64_t)wk.block * longest_stride;
      uint32_t full = 0;
      for (uint32_t p = t0 + lane; p < t1; p += 64) full += (lg[4u * (p - prev)] & 0xffffu) >= ZH_MAX_MATCH ? 1u : 0u;   // (slot 0 of the position's row)
      full = zh_wave_sum(full);
      if (2u * full <= len) {
         // K segments of S positions, segment 0 the short one: as many as fill whole segment waves (ZH_CUT_ROWS each)
         // with about cut_len (ZH_CUT_LEN) positions per row
         uint32_t K = ZH_CUT_ROWS * ((len + ZH_CUT_ROWS * cut_len - 1u) / (ZH_CUT_ROWS * cut_len));
         if (len / K < ZH_CUT_WARM) K = max(2u, len / ZH_CUT_WARM);   // (a segment is never shorter than the warm-up)
         const uint32_t S = ((len + K - 1u) / K + 31u) & ~31u;
         K = (len + S - 1u) / S;
         const uint32_t nw = (K + ZH_CUT_ROWS - 1u) / ZH_CUT_ROWS;   // segment waves
         uint32_t ti = 0, w0 = 0, it = 0;
         if (lane == 0) {
            ti = atomicAdd(&cnt[ZH_CNT_SEGTASKS], 1u);
            w0 = atomicAdd(&cnt[ZH_CNT_SEGWAVES], nw);
            it = atomicAdd(&cnt[ZH_CNT_SEGITEMS], K);
            segtasks[ti] = make_uint4(gt, ZH_CUT_PACK(K, S), it, 0u);   // vector slot of segment k = it + k
            atomicAdd(&cnt[ZH_CNT_HUGE_POS], len);      // statistics only (zultra_hip_last_stats)
         }
         ti = zh_readfirstlane(ti);
         w0 = zh_readfirstlane(w0);
         it = zh_readfirstlane(it);
         if (lane < nw) segwaves[w0 + lane] = make_uint2(ti, lane * ZH_CUT_ROWS);
         for (uint32_t k = lane; k < K; k += 64) segitems[it + k] = make_uint2(ti, K - 1u - k);   // (the exact, rightmost one first)
         return;
      }
   }
   if (lane == 0) {
      // three classes by length, handed out longest class first (zh_chain_ticket): what the pass will wait for are its longest
      // chains, so they get the first tickets of zh_parse_chain. Each class has a list of `cap` entries of its own (any of them may
      // hold every task of the run: with one shared list the long and the other tasks ran into each other when more than half of
      // a run's task slots were listed).
      if (len > ZH_CHAIN_VLONG_TASK)
         hugelist[atomicAdd(&cnt[ZH_CNT_VLONG], 1u)] = gt;
      else if (len > ZH_CHAIN_LONG_TASK)
         hugelist[cap + atomicAdd(&cnt[ZH_CNT_LONG], 1u)] = gt;
      else
         hugelist[2u * cap + atomicAdd(&cnt[ZH_CNT_SHORT], 1u)] = gt;
      atomicAdd(&cnt[ZH_CNT_HUGE_POS], len);
   }
}

// One wave per task; the length of the run's task list is known on the device only (cnt[ZH_CNT_TASKS], zh_plan_subblocks): <false> over a grid of what data
// usually gives (surplus workgroups leave at once), <true> a few workgroups that stride over what lies beyond it (see zh_sb_init).
// coop_min: tasks with a longer barrier-free piece are listed; a run of at most small_tasks tasks (one call on a few max-blocks: what counts is
// the longest chain of steps, and a chain workgroup steps faster) lists from coop_small on.
template <bool MORE>
__global__ void __launch_bounds__(64)
zh_list_huge(const zh_block_t *__restrict__ blocks, const uint64_t *__restrict__ bars, uint64_t bar_stride, const zh_work_t *__restrict__ work,
             const uint2 *__restrict__ taskmap, const uint32_t *__restrict__ longest, uint64_t longest_stride, uint32_t *hugelist, uint32_t cap, uint4 *segtasks,
             uint2 *segitems, uint2 *segwaves, uint32_t seg_min, uint32_t cut_len, uint32_t *cnt, uint2 *taskinfo,
             uint32_t coop_min /* <= ZH_COOP_MIN */, uint32_t coop_small /* <= coop_min */, uint32_t small_tasks, uint32_t first) {
   __shared__ uint32_t bnd[ZH_MAXPIECES + 1];
   const uint32_t ntasks = cnt[ZH_CNT_TASKS];
   const uint32_t cm = ntasks <= small_tasks ? coop_small : coop_min;
   if (!MORE) {
      if (blockIdx.x < ntasks) zh_list_huge_one(bnd, blockIdx.x, blocks, bars, bar_stride, work, taskmap, longest, longest_stride, hugelist, cap, segtasks, segitems, segwaves, seg_min, cut_len, cnt, taskinfo, cm);
      return;
   }
   for (uint32_t gt = first + blockIdx.x; gt < ntasks; gt += gridDim.x) {
      zh_sync();   // the task before this one is done with bnd
      zh_list_huge_one(bnd, gt, blocks, bars, bar_stride, work, taskmap, longest, longest_stride, hugelist, cap, segtasks, segitems, segwaves, seg_min, cut_len, cnt, taskinfo, cm);
   }
}

// what a chain workgroup needs to know about its task
struct zh_chain_task_t {
   const uint8_t *win;
   const uint4 *rows, *rows_hi;
   uint32_t *best;
   const zh_sbstate_t *st;
   uint32_t prev, t0, t1, sb_end;
   bool skip;
};

__device__ __forceinline__ zh_chain_task_t zh_chain_task(uint32_t gt, const uint8_t *data, const zh_block_t *blocks, const zh_match_t *match, uint64_t match_stride,
                                                         const uint64_t *bars, uint64_t bar_stride, const zh_work_t *work, const uint2 *taskmap, const zh_sbstate_t *states,
                                                         uint32_t *best_all, uint64_t best_stride, int pass) {
   zh_chain_task_t T;
   const uint2 tm = taskmap[gt];
   const zh_work_t wk = work[tm.x];
   T.st = states + tm.x;
   T.skip = T.st->failed || (!T.st->is_dynamic && pass > 0) || T.st->settled;   // static sub-blocks are parsed once (blockdeflate.c:836-858); settled ones keep the parse they have (zh_sb_build_one)
   const zh_block_t blk = blocks[wk.block];
   T.win = data + blk.win_off;
   T.prev = blk.prev;
   T.rows = (const uint4 *)(match + (uint64_t)wk.block * match_stride);
   T.rows_hi = T.rows + ZH_ROW_HI_OFF(match_stride);
   const uint64_t *bar = bars + (uint64_t)wk.block * bar_stride;
   T.best = best_all + (uint64_t)wk.block * best_stride;
   T.sb_end = wk.start + wk.size;
   T.t0 = zh_task_boundary(bar, T.prev, wk.start, T.sb_end, tm.y, wk.ntasks);
   T.t1 = zh_task_boundary(bar, T.prev, wk.start, T.sb_end, tm.y + 1, wk.ntasks);
   return T;
}

// prices of the codes in force; unused symbols price at 9 / 6 bits (blockdeflate.c:873-881). All threads; ends with a barrier.
__device__ __forceinline__ void zh_chain_prices(zh_chain_ws_t &ws, const zh_sbstate_t *st) {
   const uint32_t tid = threadIdx.x;
   for (uint32_t k = tid; k < ZH_NLIT; k += ZH_CHAIN_THREADS) {
      const uint32_t l = st->lit_len[k];
      ws.litprice[k] = (uint8_t)(l ? l : 9u);
   }
   if (tid < ZH_NDIST) {
      const uint32_t l = st->dist_len[tid];
      ws.distcost[tid] = (uint8_t)((l ? l : 6u) + (uint32_t)zh_dist_xbits((int)tid));
   }
   __syncthreads();
   for (uint32_t e = tid; e < 256; e += ZH_CHAIN_THREADS) {
      const int idx = zh_len_idx(e + 3);
      ws.lencost[e] = (uint8_t)(ws.litprice[257 + idx] + zh_lenidx_xbits(idx));
   }
   __syncthreads();
}

// histogram of the task's parse (the per-sub-block sum is taken by zh_sb_build). All threads.
__device__ __forceinline__ void zh_chain_histogram(zh_chain_ws_t &ws, const zh_chain_task_t &T, uint32_t *hp) {
   const uint32_t tid = threadIdx.x;
   __threadfence_block();
   __syncthreads();
   for (uint32_t k = tid; k < ZH_NSYM; k += ZH_CHAIN_THREADS) ws.hist[k] = 0;
   __syncthreads();
   zh_walk_histogram_wave(ws.hist, T.win, T.prev, (tid >> 6) == 0 ? T.t0 : T.t1, T.t1, T.best);   // the other waves walk nothing
   zh_sync();   // (the walk ends with a sync of its own wave only)
   for (uint32_t k = tid; k < ZH_NSYM; k += ZH_CHAIN_THREADS) hp[k] = ws.hist[k];
}

// The workgroup that finishes the last segment of a cut task checks the task: accept the segments whose speculated costs match
// what their right neighbour left, parse the others again from there, then take the task's histogram. All threads call; the
// prices of the pass are in ws.
__device__ inline void zh_chain_check_task(zh_chain_ws_t &ws, uint32_t *s_bad_p, const zh_chain_task_t &T, uint32_t gt, uint32_t K, uint32_t S, uint32_t slot0, int16_t *vecs,
                                           uint32_t *cnt, uint32_t *hist_part) {
   const uint32_t tid = threadIdx.x;
   for (uint32_t k = K - 1u; k-- > 0;) {
      int16_t *v = vecs + (uint64_t)(slot0 + k) * (2u * ZH_VEC);
      const int16_t *right = v + 2u * ZH_VEC + ZH_VEC;   // the left-end vector of segment k + 1: exact by now
      if (tid == 0) *s_bad_p = 0;
      __syncthreads();
      for (uint32_t i = tid; i < (ZH_VEC_LIVE + 1u) / 2u; i += ZH_CHAIN_THREADS)
         if (zh_load_agent_u32((const uint32_t *)v + i) != zh_load_agent_u32((const uint32_t *)right + i)) *s_bad_p = 1;
      __syncthreads();
      if (!*s_bad_p) continue;
      // parse the segment again, from the true costs
      zh_chain_job_t job;
      const uint32_t b = T.t1 - (K - 1u - k) * S;
      job.t0 = k ? b - S : T.t0;
      job.t1 = b;
      job.clamp = T.sb_end;
      job.store_hi = 0xFFFFFFFFu;
      job.import = right;
      job.export_spec = NULL;
      job.export_left = k ? v + ZH_VEC : (int16_t *)NULL;
      __syncthreads();   // (s_bad has been read)
      zh_chain_parse(ws, T.rows, T.rows_hi, T.win, T.prev, job, T.best);
      if (tid == 0) atomicAdd(&cnt[ZH_CNT_SEG_FAILED], 1u);
      __threadfence();
      __syncthreads();   // the new left vector is visible to the next comparison; the workspace is free
   }
   if (T.st->is_dynamic) zh_chain_histogram(ws, T, hist_part + (uint64_t)gt * ZH_NSYM);
}

// Persistent workgroups take the listed items from a ticket: the grid is small and fixed (ZH_CHAIN_GRID), so it is dispatched at
// once — next to zh_parse_lanes' thousands of waves — and every chain starts at the beginning of the pass. Ticket order:
// the whole tasks of the two long classes, the segments of the cut tasks, the short whole tasks.
#define ZH_CHAIN_GRID 1536
#ifndef ZH_CHAIN_VGPR_ATTR
#define ZH_CHAIN_VGPR_ATTR
#endif
__global__ void __launch_bounds__(ZH_CHAIN_THREADS) ZH_CHAIN_VGPR_ATTR
zh_parse_chain(const uint8_t *__restrict__ data, const zh_block_t *__restrict__ blocks, const zh_match_t *__restrict__ match, uint64_t match_stride,
               const uint64_t *__restrict__ bars, uint64_t bar_stride, const zh_work_t *__restrict__ work, const uint2 *__restrict__ taskmap,
               const uint32_t *__restrict__ hugelist, uint32_t cap, uint4 *segtasks, const uint2 *__restrict__ segitems, int16_t *vecs,
               uint32_t seg_wide_min /* a run with fewer segments than this has them parsed here, one job each (zh_segments_are_wide) */, uint32_t seg_whole /* ... cut tasks shorter than this as one job */,
               uint32_t *cnt, const zh_sbstate_t *__restrict__ states,
               uint32_t *best_all, uint64_t best_stride, uint32_t *hist_part, int pass, uint32_t *ticket, uint64_t *trace /* diagnostics (ZULTRA_HIP_CHAIN_TRACE): per ticket {positions, start, end} on the 100 MHz clock, or NULL */) {
   __shared__ zh_chain_ws_t ws;
   __shared__ uint32_t s_item, s_bad;
   const uint32_t tid = threadIdx.x;
   // the cut tasks that the segment workgroups gave up on in the passes before this one (zh_parse_one_task: the fourth list, in the order of the
   // passes): whole chains, and the longest of the pass — they get the first tickets
   uint32_t ndem = 0;
   for (int q = 0; q < pass; q++) ndem += cnt[ZH_CNT_DEMOTED_PASS + q];
   const uint32_t nseg = (cnt[ZH_CNT_SEGTASKS] != 0 && !zh_segments_are_wide(cnt, seg_wide_min)) ? cnt[ZH_CNT_SEGITEMS] : 0u;   // the run's segments if they are parsed here
   const uint32_t nvlong = ndem + cnt[ZH_CNT_VLONG], nlong = cnt[ZH_CNT_LONG], count = nvlong + nlong + nseg + cnt[ZH_CNT_SHORT];
   if (count == 0) return;   // (the grid is launched whatever the run holds: nothing listed, nothing to take a ticket for)
   for (;;) {
      __syncthreads();   // the previous task's histogram has left LDS, s_item has been read
      if (tid == 0) s_item = atomicAdd(ticket, 1u);
      __syncthreads();
      const uint32_t item = s_item;
      if (item >= count) return;
      const bool is_seg = item >= nvlong + nlong && item < nvlong + nlong + nseg;
      uint32_t gt, K = 1, S = 0, k = 0, slot = 0, ti = 0;
      if (is_seg) {
         slot = item - nvlong - nlong;
         const uint2 si = segitems[slot];
         ti = si.x;
         const uint4 stask = segtasks[ti];
         gt = stask.x;
         K = ZH_CUT_K(stask.y);
         S = ZH_CUT_S(stask.y);
         k = si.y;
         slot = stask.z + k;
      }
      else
         gt = item < ndem ? hugelist[3u * cap + item]
                          : (item < nvlong ? hugelist[item - ndem] : (item < nvlong + nlong ? hugelist[cap + (item - nvlong)] : hugelist[2u * cap + (item - nvlong - nlong - nseg)]));
      const zh_chain_task_t T = zh_chain_task(gt, data, blocks, match, match_stride, bars, bar_stride, work, taskmap, states, best_all, best_stride, pass);
      if (T.skip) continue;
      zh_chain_prices(ws, T.st);
      zh_chain_job_t job;
      job.t0 = T.t0;
      job.t1 = T.t1;
      job.clamp = T.sb_end;
      job.store_hi = 0xFFFFFFFFu;
      job.import = NULL;
      job.export_spec = NULL;
      job.export_left = NULL;
      bool whole = !is_seg;
      if (is_seg && T.t1 - T.t0 < seg_whole) {
         // a few short cut tasks: a chain of this length is over before the check of its segments would be
         if (k + 1u != K) continue;
         whole = true;
      }
      else if (is_seg) {
         int16_t *v = vecs + (uint64_t)slot * (2u * ZH_VEC);   // [0]: speculated at the segment's right end, [1]: its left end
         const uint32_t b = T.t1 - (K - 1u - k) * S;     // the segment's right end
         job.t0 = k ? b - S : T.t0;
         if (k) job.export_left = v + ZH_VEC;
         if (k + 1u < K) {
            job.t1 = job.clamp = b + ZH_CUT_WARM;            // as if the sub-block ended there
            job.store_hi = b;
            job.export_spec = v;
         }
      }
      const uint64_t trace_t0 = trace ? zh_wall_clock() : 0;
      zh_chain_parse(ws, T.rows, T.rows_hi, T.win, T.prev, job, T.best);
      if (trace && tid == 0 && item < ZH_TRACE_SLOTS) {
         trace[3 * (uint64_t)item] = job.t1 - job.t0;
         trace[3 * (uint64_t)item + 1] = trace_t0;
         trace[3 * (uint64_t)item + 2] = zh_wall_clock();
      }
      if (whole) {
         if (T.st->is_dynamic) zh_chain_histogram(ws, T, hist_part + (uint64_t)gt * ZH_NSYM);
         continue;
      }
      // a segment: the task's segments are counted (the counter runs on over the passes), the workgroup that finishes the last one checks them
      __threadfence();   // this segment's parse entries and vectors are out
      __syncthreads();
      if (tid == 0) s_bad = ((atomicAdd(&segtasks[ti].w, 1u) & 0xffffu) + 1u) % K;   // (high half: the task's failed cuts, zh_chain_check_task)
      __syncthreads();
      if (s_bad != 0) continue;
      __threadfence();   // ... and the other segments' are in
      zh_chain_check_task(ws, &s_bad, T, gt, K, S, slot - k, vecs, cnt, hist_part);
   }
}

