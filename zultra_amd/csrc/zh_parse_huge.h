// zh_parse_huge.h — the optimal parse (zh_parse.h) for tasks that contain a barrier-free run of more than ZH_COOP_MIN
// positions: repeated boilerplate, near-copies of earlier data, long matches everywhere. A 64 KiB max-block of such
// data can be ONE run, and then the recurrence cost[p] <- cost[p+1..p+258] is the whole run time: nothing but the
// instruction count of one step decides how long the batch takes (measured on gfx950, one wave alone on a CU: about
// 5 cycles per instruction whatever it is, 11 per DPP step, 30 per taken branch, 75 per dependent LDS access).
//
// So the step is stripped to what depends on the costs, and everything else moves to other waves of the workgroup:
//   producer waves load the match rows of the NEXT tile of 48 positions, digest the 8 slots of each position and
//                  write, for every (position, candidate) pair, one 32-bit descriptor: where in the cost ring the
//                  candidate's cost[p + len] lives, its price (length code + distance code), and the low key bits
//                  (slot, length) that break ties the way the reference's evaluation order does. They also turn the
//                  winning keys of the PREVIOUS tile into parse entries and store them.
//   consumer wave  prices three positions per step (rows 0..2 of the wave: positions p, p-1, p-2; lane s of a row holds
//                  candidates 3+s, 19+s, 35+s and long slot s): per candidate one ring read (issued a step ahead), four
//                  ALU operations, then the row minimum, three readlanes and the scalar literal/match chain.
// Producers and consumer meet at one workgroup barrier per tile. A task is parsed as ONE chain from its end to its start — the
// barriers inside it are not needed for exactness, only for parallelism, and a task on this path has too few of them.
//
// Reads issued a step ahead come before the three cost writes of the current step in program (= LDS) order, so of all
// the ring reads of step t+1 exactly those of cost[p0], cost[p0-1], cost[p0-2] (the positions of step t) are stale. Only
// length 3+s of lanes s <= row can ask for them, and those lanes take the value from the scalar results of step t.
#pragma once
#include "zh_parse.h"

#define ZH_HUGE_THREADS 256          // wave 0: the recurrence; waves 1..3: 48 positions x 4 candidate groups of the next tile
#define ZH_HUGE_GRID 2048            // workgroups; each takes the listed tasks round-robin
#define ZH_HUGE_TILE 48
#define ZH_DESC_INVALID 0x08000000u  // descriptor of a candidate that does not exist: its key is above every real one
#define ZH_DESC_KEYMASK 0x0801ffffu  // bits 0..8 low key bits | 9..16 price | 17..26 ring byte address | 27 invalid

struct zh_huge_ws_t {
   union {
      struct {
         uint16_t ring[512];          // cost[p & 511] mod 2^16 (zh_parse.h: ZH_KEY_BIAS)
         uint4 desc[2][16][64];       // [tile parity][step][consumer lane]: the lane's four candidate descriptors
         uint32_t bt[2][16][4];       // winning key per (step, row); all ones = literal
         uint32_t lit[2][16][4];      // literal price per (step, row)
      } p;
      uint32_t hist[ZH_NSYM];         // after the parse: histogram of the task
   };
   uint8_t litprice[ZH_NLIT];
   uint8_t lencost[256];
   uint8_t distcost[ZH_NDIST];
};

// what the producer loads from HBM for one position (issued a whole tile period before it is digested)
struct zh_huge_fetch_t {
   uint4 a, b;
   uint32_t byte;
};

__device__ __forceinline__ void zh_huge_fetch(zh_huge_fetch_t &f, const uint4 *rows, const uint8_t *win, uint32_t prev, uint32_t thi, uint32_t cnt,
                                              uint32_t pl) {
   const uint32_t j = pl >> 2;
   const uint32_t pos = j < cnt ? thi - 1 - j : thi - 1;   // clamped: the loads are always legal
   f.a = rows[(uint64_t)(pos - prev) * 2];
   f.b = rows[(uint64_t)(pos - prev) * 2 + 1];
   f.byte = win[pos];
}

// the producer's digest of one position
struct zh_huge_pos_t {
   uint4 a, b;                // the 8 slots: len | offset << 16, longest first
   uint32_t nlong, room;
};

// producer: stage tile [thi - cnt, thi) into parity `buf`. Producer thread pl = 4 j + part: position thi-1-j, candidate
// group `part` (0: lengths 3..18, 1: 19..34, 2: 35..39, 3: the slots stored with length >= 40). The four threads of a
// position each digest its slots (cheap) and write their own 16 descriptors.
__device__ __forceinline__ void zh_huge_stage(zh_huge_ws_t &ws, uint32_t buf, const zh_huge_fetch_t &f, uint32_t thi, uint32_t cnt, uint32_t sb_end,
                                              uint32_t pl, zh_huge_pos_t &st) {
   const uint32_t j = pl >> 2, part = pl & 3u;
   const uint32_t t = j / 3, row = j - 3 * t;
   uint32_t *d = (uint32_t *)&ws.p.desc[buf][t][row * 16] + part;   // [s * 4]
   const bool ok = j < cnt;
   const uint32_t pos = ok ? thi - 1 - j : thi - 1;
   const uint4 a = f.a, b = f.b;
   const uint32_t byte = f.byte;
   const uint32_t room = sb_end - pos;
   st.a.x = ok ? a.x : 0u; st.a.y = ok ? a.y : 0u; st.a.z = ok ? a.z : 0u; st.a.w = ok ? a.w : 0u;
   st.b.x = ok ? b.x : 0u; st.b.y = ok ? b.y : 0u; st.b.z = ok ? b.z : 0u; st.b.w = ok ? b.w : 0u;
   st.room = room;
   const uint32_t raw[ZH_NMATCH] = {st.a.x, st.a.y, st.a.z, st.a.w, st.b.x, st.b.y, st.b.z, st.b.w};
   // the same digest as zh_stage_position: bitmap of short slot lengths, running minima of (distance price, slot)
   uint32_t nlong = 0, nshort = 0, kmax = 0, run = 0xFFu;
   uint64_t pm = 0, lmask = 0;
   uint32_t oc[ZH_NMATCH];
#pragma unroll
   for (uint32_t m = 0; m < ZH_NMATCH; m++) {
      const uint32_t len = raw[m] & 0xffffu, off = raw[m] >> 16;
      const bool valid = len >= ZH_MIN_MATCH;
      const bool is_long = len >= ZH_LEAVE_ALONE;
      const bool is_short = valid && !is_long;
      oc[m] = (uint32_t)ws.distcost[zh_dist_sym(valid ? off : 1u)];
      nlong += is_long ? 1u : 0u;
      kmax = max(kmax, is_short ? len : 0u);               // the first short slot is the longest
      lmask |= is_short ? (1ull << (len - ZH_MIN_MATCH)) : 0ull;
      run = is_short ? min(run, (oc[m] << 3) | m) : run;
      pm |= is_short ? ((uint64_t)run << (8 * nshort)) : 0ull;
      nshort += is_short ? 1u : 0u;
   }
   st.nlong = nlong;
   kmax = min(kmax, room);                                  // end clamp (blockdeflate.c:283-284)
   if (part < 3) {
      // lengths k = 3 + 16 part + s: the last short slot reaching k has the cheapest distance among those that can provide it
      const uint32_t e0 = 16u * part;
      const uint32_t w = (uint32_t)(lmask >> e0);                         // lengths e0+3 .. e0+34
      const uint32_t above = (uint32_t)__popcll(lmask >> (e0 + 16u) >> 16u);   // slots longer than that window
      const uint32_t pm_lo = (uint32_t)pm, pm_hi = (uint32_t)(pm >> 32);
#pragma unroll
      for (uint32_t i = 0; i < 16; i++) {
         const uint32_t s = (i + j) & 15u;   // the 64 threads of a store hit 64 different LDS banks
         const uint32_t k = ZH_MIN_MATCH + e0 + s;
         const uint32_t sel = (uint32_t)__popc(w >> s) + above - 1u;
         const uint32_t bb = ((sel < 4 ? pm_lo : pm_hi) >> ((sel & 3u) * 8u)) & 0xffu;
         const uint32_t price = (uint32_t)ws.lencost[e0 + s] + (bb >> 3);
         const uint32_t v = ((((pos + k) & 511u) * 2u) << 17) | (price << 9) | ((bb & 7u) << 6) | (39u - k);
         d[s * 4u] = (k <= kmax && k < ZH_LEAVE_ALONE) ? v : ZH_DESC_INVALID;
      }
   }
   else {
      // slots stored with length >= 40: tried at their full (clamped) length only (blockdeflate.c:286-297)
#pragma unroll
      for (uint32_t s = 0; s < ZH_NMATCH; s++) {
         const uint32_t mlen = min(raw[s] & 0xffffu, room);
         uint32_t enc = mlen - ZH_MIN_MATCH;                 // wraps below 3, then saturates (:289, :216-219)
         if (enc > 255) enc = 255;
         const uint32_t price = (uint32_t)ws.lencost[enc] + oc[s];
         const uint32_t v = ((((pos + mlen) & 511u) * 2u) << 17) | (price << 9) | (s << 6);
         d[s * 4u] = s < nlong ? v : ZH_DESC_INVALID;
         d[(s + ZH_NMATCH) * 4u] = ZH_DESC_INVALID;
      }
      ws.p.lit[buf][t][row] = ok ? (uint32_t)ws.litprice[byte & 0xffu] : 0u;
   }
}

// producer: the winning keys of a priced tile -> parse entries (zh_decode_pick)
__device__ __forceinline__ void zh_huge_flush(zh_huge_ws_t &ws, uint32_t buf, uint32_t prev, uint32_t thi, uint32_t cnt, uint32_t pl,
                                              const zh_huge_pos_t &st, uint32_t *best) {
   const uint32_t lane = pl >> 2;   // position thi-1-lane; one of its four threads stores
   if ((pl & 3u) || lane >= cnt) return;
   const uint32_t t = lane / 3, row = lane - 3 * t;
   const uint32_t kk = ws.p.bt[buf][t][row];
   uint32_t pick = 0;
   if (kk != 0xFFFFFFFFu) {
      const uint32_t m = (kk >> 6) & 7u;
      const uint32_t raw[ZH_NMATCH] = {st.a.x, st.a.y, st.a.z, st.a.w, st.b.x, st.b.y, st.b.z, st.b.w};
      uint32_t e = raw[0];
#pragma unroll
      for (uint32_t q = 1; q < ZH_NMATCH; q++) e = (m == q) ? raw[q] : e;
      const uint32_t len = (m < st.nlong) ? min(e & 0xffffu, st.room) : (39u - (kk & 63u));
      pick = len | (e & 0xffff0000u);
   }
   best[(thi - 1 - lane) - prev] = pick;
}

__device__ __forceinline__ uint32_t zh_huge_ring_at(const zh_huge_ws_t &ws, uint32_t desc) {
   return *(const uint16_t *)((const uint8_t *)ws.p.ring + (desc >> 17 & 0x3feu));
}

// the tasks zh_parse_tasks leaves alone: one wave per task, the same piece computation
__global__ void __launch_bounds__(64)
zh_list_huge(const zh_block_t *__restrict__ blocks, const uint64_t *__restrict__ bars, uint64_t bar_stride, const zh_work_t *__restrict__ work,
             const uint2 *__restrict__ taskmap, const uint32_t *__restrict__ ntasks_total, uint32_t *hugelist, uint32_t *nhuge,
             uint32_t *huge_positions) {
   __shared__ uint32_t bnd[ZH_MAXPIECES + 1];
   const uint32_t gt = blockIdx.x;
   if (gt >= *ntasks_total) return;
   const uint2 tm = taskmap[gt];
   const zh_work_t wk = work[tm.x];
   const uint32_t prev = blocks[wk.block].prev;
   const uint64_t *bar = bars + (uint64_t)wk.block * bar_stride;
   const uint32_t lane = zh_lane(), sb_end = wk.start + wk.size;
   const uint32_t t0 = zh_task_boundary(bar, prev, wk.start, sb_end, tm.y, wk.ntasks);
   const uint32_t t1 = zh_task_boundary(bar, prev, wk.start, sb_end, tm.y + 1, wk.ntasks);
   const uint32_t np = zh_task_pieces(bnd, bar, prev, t0, t1, lane);
   zh_sync();
   if (zh_task_is_huge(bnd, np, lane) && lane == 0) {
      hugelist[atomicAdd(nhuge, 1u)] = gt;
      atomicAdd(huge_positions, t1 - t0);   // statistics only (zultra_hip_last_stats)
   }
}

__global__ void __launch_bounds__(ZH_HUGE_THREADS)
zh_parse_huge(const uint8_t *__restrict__ data, const zh_block_t *__restrict__ blocks, const zh_match_t *__restrict__ match,
              uint64_t match_stride, const uint64_t *__restrict__ bars, uint64_t bar_stride, const zh_work_t *__restrict__ work,
              const uint2 *__restrict__ taskmap, const uint32_t *__restrict__ hugelist, const uint32_t *__restrict__ nhuge,
              const zh_sbstate_t *__restrict__ states, uint32_t *best_all, uint64_t best_stride, uint32_t *hist_part, int pass) {
   __shared__ zh_huge_ws_t ws;
   const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
   const uint32_t row = lane >> 4, s = lane & 15;
   const uint32_t count = *nhuge;
   for (uint32_t item = blockIdx.x; item < count; item += gridDim.x) {
      const uint32_t gt = hugelist[item];
      const uint2 tm = taskmap[gt];
      const zh_work_t wk = work[tm.x];
      const zh_sbstate_t *st = states + tm.x;
      if (st->failed) continue;
      if (!st->is_dynamic && pass > 0) continue;   // static sub-blocks are parsed once (blockdeflate.c:836-858)
      const zh_block_t blk = blocks[wk.block];
      const uint8_t *win = data + blk.win_off;
      const uint32_t prev = blk.prev;
      const uint4 *rows = (const uint4 *)(match + (uint64_t)wk.block * match_stride);
      const uint64_t *bar = bars + (uint64_t)wk.block * bar_stride;
      uint32_t *best = best_all + (uint64_t)wk.block * best_stride;
      const uint32_t sb_end = wk.start + wk.size;

      // ---- prices of the codes in force; unused symbols price at 9 / 6 bits (blockdeflate.c:873-881) ---------------
      __syncthreads();   // the previous task's histogram has left LDS
      for (uint32_t k = tid; k < ZH_NLIT; k += ZH_HUGE_THREADS) {
         const uint32_t l = st->lit_len[k];
         ws.litprice[k] = (uint8_t)(l ? l : 9u);
      }
      if (tid < ZH_NDIST) {
         const uint32_t l = st->dist_len[tid];
         ws.distcost[tid] = (uint8_t)((l ? l : 6u) + (uint32_t)zh_dist_xbits((int)tid));
      }
      __syncthreads();
      for (uint32_t e = tid; e < 256; e += ZH_HUGE_THREADS) {
         const int idx = zh_len_idx(e + 3);
         ws.lencost[e] = (uint8_t)(ws.litprice[257 + idx] + zh_lenidx_xbits(idx));
      }
      const uint32_t t0 = zh_task_boundary(bar, prev, wk.start, sb_end, tm.y, wk.ntasks);
      const uint32_t t1 = zh_task_boundary(bar, prev, wk.start, sb_end, tm.y + 1, wk.ntasks);
      const uint32_t ntiles = t1 > t0 ? (t1 - t0 + ZH_HUGE_TILE - 1) / ZH_HUGE_TILE : 0u;
      __syncthreads();

      zh_huge_pos_t cur, old;   // producer: the digest of the tile staged last / the one before (whose keys are due)
      cur.a.x = cur.a.y = cur.a.z = cur.a.w = cur.b.x = cur.b.y = cur.b.z = cur.b.w = cur.nlong = cur.room = 0;
      old = cur;
      zh_huge_fetch_t fnext;   // producer: the HBM loads of the tile after the one being staged
      fnext.a = cur.a;
      fnext.b = cur.b;
      fnext.byte = 0;
      if (ntiles) {
         if (wave) {
            zh_huge_fetch_t f0;
            zh_huge_fetch(f0, rows, win, prev, t1, min((uint32_t)ZH_HUGE_TILE, t1 - t0), tid - 64);
            if (ntiles > 1) zh_huge_fetch(fnext, rows, win, prev, t1 - ZH_HUGE_TILE, min((uint32_t)ZH_HUGE_TILE, t1 - ZH_HUGE_TILE - t0), tid - 64);
            zh_huge_stage(ws, 0, f0, t1, min((uint32_t)ZH_HUGE_TILE, t1 - t0), sb_end, tid - 64, cur);
         }
         else if (lane == 0)
            ws.p.ring[t1 & 511] = 0;   // cost[task end] = 0: a barrier, or the sub-block end
      }
      __syncthreads();

      uint32_t cnext = 0;                  // consumer: cost[p0 + 1] mod 2^16, p0 = the position row 0 prices next
      uint32_t C0 = 0, C1 = 0, C2 = 0;     // costs of the three positions of the last step (cost[t1] = 0 stands in for C2)
      const bool sel0 = s == row, sel1 = s + 1 == row, patched = s <= row && row < 3;
      for (uint32_t k = 0; k < ntiles; k++) {
         const uint32_t thi = t1 - k * ZH_HUGE_TILE;
         const uint32_t cnt = min((uint32_t)ZH_HUGE_TILE, thi - t0);
         const uint32_t buf = k & 1u;
         if (wave) {
            if (k) zh_huge_flush(ws, buf ^ 1u, prev, thi + ZH_HUGE_TILE, ZH_HUGE_TILE, tid - 64, old, best);
            old = cur;
            if (k + 1 < ntiles) {
               const uint32_t nhi = thi - ZH_HUGE_TILE;
               const zh_huge_fetch_t f = fnext;
               if (k + 2 < ntiles) zh_huge_fetch(fnext, rows, win, prev, nhi - ZH_HUGE_TILE, min((uint32_t)ZH_HUGE_TILE, nhi - ZH_HUGE_TILE - t0), tid - 64);
               zh_huge_stage(ws, buf ^ 1u, f, nhi, min((uint32_t)ZH_HUGE_TILE, nhi - t0), sb_end, tid - 64, cur);
            }
         }
         else {
            const uint32_t steps = (cnt + 2) / 3;
            uint4 D = ws.p.desc[buf][0][lane], D1 = ws.p.desc[buf][1][lane];
            uint32_t lt = ws.p.lit[buf][0][row];
            uint32_t g0 = zh_huge_ring_at(ws, D.x), g1 = zh_huge_ring_at(ws, D.y), g2 = zh_huge_ring_at(ws, D.z), g3 = zh_huge_ring_at(ws, D.w);
            for (uint32_t t = 0; t < steps; t++) {
               // ---- next step's LDS reads, before this step's costs are written --------------------------------------
               const uint4 D2 = ws.p.desc[buf][min(t + 2, 15u)][lane];
               const uint32_t ltn = ws.p.lit[buf][min(t + 1, 15u)][row];
               const uint32_t n0 = zh_huge_ring_at(ws, D1.x), n1 = zh_huge_ring_at(ws, D1.y), n2 = zh_huge_ring_at(ws, D1.z), n3 = zh_huge_ring_at(ws, D1.w);
               // ---- the recurrence --------------------------------------------------------------------------------------
               const uint32_t base = cnext - ZH_KEY_BIAS;   // key cost = (candidate cost - base) mod 2^16, below 2^15
               const uint32_t cv = sel0 ? C0 : (sel1 ? C1 : C2);
               const uint32_t r0 = patched ? cv : g0;
               const uint32_t k0 = (((r0 - base) & 0xffffu) << 9) + (D.x & ZH_DESC_KEYMASK);
               const uint32_t k1 = (((g1 - base) & 0xffffu) << 9) + (D.y & ZH_DESC_KEYMASK);
               const uint32_t k2 = (((g2 - base) & 0xffffu) << 9) + (D.z & ZH_DESC_KEYMASK);
               const uint32_t k3 = (((g3 - base) & 0xffffu) << 9) + (D.w & ZH_DESC_KEYMASK);
               const uint32_t rkey = zh_row_min(min(min(k0, k1), min(k2, k3)));
               // literal first; a match must be strictly cheaper (:292,:307); the three decisions chain: cost[p-1] needs cost[p]
               const uint32_t m0 = zh_readlane(rkey, 0) >> 9, m1 = zh_readlane(rkey, 16) >> 9, m2 = zh_readlane(rkey, 32) >> 9;
               const uint32_t l0 = zh_readlane(lt, 0) + ZH_KEY_BIAS;
               const uint32_t c0 = min(l0, m0);
               const uint32_t l1 = zh_readlane(lt, 16) + c0;
               const uint32_t c1 = min(l1, m1);
               const uint32_t l2 = zh_readlane(lt, 32) + c1;
               const uint32_t c2 = min(l2, m2);
               C0 = (base + c0) & 0xffffu;
               C1 = (base + c1) & 0xffffu;
               C2 = (base + c2) & 0xffffu;
               const bool act1 = 3 * t + 1 < cnt, act2 = 3 * t + 2 < cnt;
               {
                  // every lane of a row holds that row's best key: the row's own decision without leaving the vector unit
                  const uint32_t lrow = row == 0 ? l0 : (row == 1 ? l1 : l2), mrow = rkey >> 9;
                  if (s == 0 && row < 3 && 3 * t + row < cnt) {
                     ws.p.ring[(thi - 1 - 3 * t - row) & 511] = (uint16_t)(base + min(lrow, mrow));
                     ws.p.bt[buf][t][row] = mrow < lrow ? rkey : 0xFFFFFFFFu;
                  }
               }
               zh_wave_sync();
               cnext = act2 ? C2 : (act1 ? C1 : C0);
               D = D1;
               D1 = D2;
               lt = ltn;
               g0 = n0; g1 = n1; g2 = n2; g3 = n3;
            }
         }
         __syncthreads();
      }
      if (ntiles && wave) {
         const uint32_t k = ntiles - 1;
         const uint32_t thi = t1 - k * ZH_HUGE_TILE;
         zh_huge_flush(ws, k & 1u, prev, thi, thi - t0, tid - 64, old, best);
      }

      // ---- histogram of the task's parse; the per-sub-block sum is taken by zh_sb_build -------------------------------
      if (st->is_dynamic) {
         __threadfence_block();
         __syncthreads();
         for (uint32_t k = tid; k < ZH_NSYM; k += ZH_HUGE_THREADS) ws.hist[k] = 0;
         __syncthreads();
         zh_walk_histogram_wave(ws.hist, win, prev, wave == 0 ? t0 : t1, t1, best);   // the other waves walk nothing (they join the barrier)
         uint32_t *hp = hist_part + (uint64_t)gt * ZH_NSYM;
         for (uint32_t k = tid; k < ZH_NSYM; k += ZH_HUGE_THREADS) hp[k] = ws.hist[k];
      }
   }
}
