// zh_mf_group_lds.h — zh_mf_group's orders, refined in LDS (round 4).
//
// What zh_mf_frontier needs from this kernel (zh_matchfinder.h) is the 6-gram order of a segment's window — classes contiguous and
// ascending in position, class heads marked — with, next to every entry, the distances to the nearest earlier position sharing 3, 4
// and 5 bytes. Rounds 1-3 built it with six stable counting passes over the whole window, every one of them a round trip of the
// order (and its payload) through HBM: 82 bytes of traffic per window position, the waves parked on scattered stores and gathers
// 60 % of their cycles (profiles/r03_sq_counters_c2.csv) — although the window itself sits in LDS.
//
// Here only the BIGRAM order goes through HBM (two passes: byte 1, then byte 0; the first one carries byte 0 in the element's spare
// bits, so the second gathers nothing). A bigram class is a few dozen to a few thousand positions, ascending; consecutive classes are
// taken into LDS as a *chunk* of up to `cap` elements (3936 next to a 96 KiB window) and refined there by the whole workgroup:
//   pass on byte 2 -> trigram classes        (the order before it: bigram classes)
//   pass on byte 3 -> 4-gram classes, and while it reads the trigram order: d3 = distance to the element before, if in the same class
//   pass on byte 4 -> 5-gram classes, d4 likewise
//   pass on byte 5 -> 6-gram classes, d5 likewise
// A pass over a chunk is a stable counting sort of all its classes at once (the digit is the primary key afterwards, which nobody
// minds: classes stay contiguous and ascending in position). An element is position | id << 17, id = its index in the chunk as
// loaded; the distances go to tables indexed by id, so only one word per element moves in a pass. A sweep over the final order marks
// the class heads and writes entries and distances to HBM, coalesced: 28 bytes of HBM traffic per window position all told.
// A window of at most `cap` positions (files mode: 4 KiB inputs) never leaves LDS: two more passes there give the bigram order.
// A class larger than `cap` (byte runs, alphabets of two or three symbols) is refined by the generic passes of zh_matchfinder.h on
// its range of the bigram order, through HBM as before.
#pragma once

#define ZH_MFL_POS_MASK 0x1ffffu   // a segment window is at most ZH_SEG_WINDOW = 96 Ki positions
#define ZH_MFL_ID_SHIFT 17
#define ZH_MFL_MAXCAP 4096u        // 16 waves x 4 steps of 64 elements
#ifndef ZH_MFL_CAP_LIMIT
#define ZH_MFL_CAP_LIMIT ZH_MFL_MAXCAP   // (the emulator build takes a small one: several chunks and oversized classes in a window of a few KB)
#endif
#ifndef ZH_MFL_DEBUG
#define ZH_MFL_DEBUG 0   // timing experiments (wrong output; with ZH_MF_STOP=5): 1 no distances, 2 no ranks, 4 no digit gather
#endif
#define ZH_MFL_NOTE_WORDS 59u        // the run table's slack: run_stride = sort_stride + 576 words, zh_mf_build_runs writes at most W + 517
#define ZH_MFL_MAX_NOTES ((ZH_MFL_NOTE_WORDS - 1u) / 2u)   // 29: a class above `cap` (~3900 of at most 98304 entries) happens at most 25 times
#define ZH_MF_LDS_TOTAL 163840u    // all of a CU's LDS: one workgroup per CU anyway (1024 threads)
static_assert(ZH_SEG_WINDOW <= (1u << ZH_MFL_ID_SHIFT), "window positions must fit the element's position field");
static_assert(ZH_MFL_MAXCAP <= (1u << (32 - ZH_MFL_ID_SHIFT)), "chunk ids must fit the element's id field");
static_assert(ZH_MFL_MAXCAP <= ZH_MF_WAVES * 256u, "a wave takes at most four steps of 64 elements of a chunk");

struct zh_mfl_t {
   uint16_t *hist;   // [wave][256]: digit counts of the wave's stretch, then where its elements of that digit go
   uint32_t *tot;    // [256] digit totals
   uint32_t *cur;    // [256] their exclusive scan
   uint32_t *misc;   // [8] small workgroup-wide variables
   uint32_t *X, *Y;  // [cap] the order, ping and pong
   uint32_t *D34;    // [cap] by id: d3 | d4 << 16
   uint16_t *D5;     // [cap] by id: d5
   uint32_t cap;
};

// the arrays behind the window copy (W bytes + 16 of slack at the start of the dynamic LDS)
__device__ __forceinline__ zh_mfl_t zh_mfl_layout(uint32_t *dyn_lds, uint32_t W, uint32_t cap_limit) {
   zh_mfl_t L;
   uint32_t *p = dyn_lds + (((W + 16u + 15u) >> 4) << 2);
   L.hist = (uint16_t *)p;
   p += ZH_MF_WAVES * 128u;
   L.tot = p;
   p += 256;
   L.cur = p;
   p += 256;
   L.misc = p;
   p += 8;
   const uint32_t avail = ZH_MF_LDS_TOTAL / 4u - (uint32_t)(p - dyn_lds);
   uint32_t cap = ((avail * 2u) / 7u) & ~15u;   // X, Y, D34: a word each; D5: half a word
   cap = min(cap, min((uint32_t)ZH_MFL_MAXCAP, max(cap_limit, 16u)));
   L.cap = cap;
   L.X = p;
   p += cap;
   L.Y = p;
   p += cap;
   L.D34 = p;
   p += cap;
   L.D5 = (uint16_t *)p;
   return L;
}

// One stable counting pass over the n elements IN[0 .. n) of a chunk in LDS, by the whole workgroup: digit = byte K of the string at the
// element's position; elements whose string ends before that byte drop out (they are the last positions of the window). Returns the
// number of elements written to OUT. PREVK = 3, 4, 5: IN is the PREVK-gram order — the pass records, for every element of IN, the
// distance to the element before it when that one is in the same class (distance - 1 in 16 bits, 0xffff: none within ZH_MAX_DIST),
// in D34 / D5 under the element's id; the last five window positions never reach the 6-gram order, theirs go to tail[position] (see
// zh_mf_sort_pass). Wave w takes the w-th stretch of IN; the lanes of a 64-element step with the same digit find each other with eight
// ballots and the first of them counts them all (no atomics), a 256-thread scan turns the counts into places.
template <int K, int PREVK>
__device__ inline uint32_t zh_mfl_pass(const zh_mfl_t &L, const uint32_t *lwin32, uint32_t W, uint32_t n, const uint32_t *IN, uint32_t *OUT, uint2 *tail, bool excl, uint64_t &mfg_t_) {
   const uint8_t *gwin = (const uint8_t *)lwin32;
   const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
   const uint32_t stretch = (((n + ZH_MF_WAVES - 1u) / ZH_MF_WAVES) + 63u) & ~63u;   // <= 256
   const uint32_t lo = wave * stretch, hi = min(n, lo + stretch);
   uint16_t *myhist = L.hist + wave * 256u;
   uint32_t *myhist32 = (uint32_t *)myhist;
   myhist32[lane] = 0;
   myhist32[64u + lane] = 0;
   zh_lockstep_sync();
   // A lane's place among the wave's elements of its digit: the lanes of a step with the same digit find each other with eight ballots
   // (zh_peers8), the first of them moves the wave's counter of the digit — two digits to a word, 16-bit stores — and everybody reads it
   // before: what the steps before put there. (Measured on MI355X, tools/mfg_profile.py: a returning LDS atomic per lane — the LDS serves
   // the lanes that meet at an address in lane order, tools/probes/lds_rank_probe.hip — costs a wave about a cycle per lane whatever the
   // addresses: 5200 of a pass's 9500 cycles with sixteen waves at it.)
   uint32_t e4[4], d4[4], o4[4];
#pragma unroll
   for (uint32_t u = 0; u < 4; u++) {
      e4[u] = 0;
      d4[u] = 0xffffffffu;
      o4[u] = 0;
      if (u * 64u < stretch) {   // (wave-uniform)
         const uint32_t idx = lo + u * 64u + lane;
         if (idx < hi) {
            const uint32_t e = IN[idx], pos = e & ZH_MFL_POS_MASK;
            e4[u] = e;
            if (PREVK && !(ZH_MFL_DEBUG & 1)) {
               const uint32_t id = e >> ZH_MFL_ID_SHIFT;
               uint32_t dd = 0xffffu;
               if (idx > 0) {
                  const uint32_t q = IN[idx - 1] & ZH_MFL_POS_MASK;
                  const uint32_t x = zh_load32_at(lwin32, q) ^ zh_load32_at(lwin32, pos);
                  bool same = PREVK == 3 ? (x & 0xffffffu) == 0 : x == 0;
                  if (PREVK == 5) same = same && gwin[q + 4] == gwin[pos + 4];
                  const uint32_t dist = pos - q;
                  if (same && dist <= ZH_MAX_DIST) dd = dist - 1u;
               }
               if (excl && zh_mf_prev_is_neighbour<PREVK>(gwin, pos)) dd = 0;   // (the position before it may not be in the order: zh_mf_run_interior; `excl`: this window has such positions)
               uint32_t both = dd | 0xffff0000u;
               if (PREVK == 3) L.D34[id] = both;
               if (PREVK == 4) {
                  ((uint16_t *)L.D34)[2u * id + 1u] = (uint16_t)dd;
                  if (pos + 5u >= W) both = (L.D34[id] & 0xffffu) | (dd << 16);
               }
               if (PREVK == 5) L.D5[id] = (uint16_t)dd;
               if (pos + 5u >= W) {
                  if (PREVK == 3)
                     tail[pos] = make_uint2(both, 0xffffu);
                  else if (PREVK == 4)
                     tail[pos].x = both;
                  else
                     tail[pos].y = dd;
               }
            }
            if (pos + (uint32_t)K < W) d4[u] = (ZH_MFL_DEBUG & 4) ? (pos >> 3) & 0xffu : (uint32_t)gwin[pos + (uint32_t)K];
         }
      }
   }
#pragma unroll
   for (uint32_t u = 0; u < 4; u++) {
      if (u * 64u < stretch) {
         const uint32_t d = d4[u];
         const bool valid = d != 0xffffffffu;
         const uint64_t peers = zh_peers8(d, valid);
         const uint32_t rank = zh_rank_below(peers);
         const uint32_t before = valid ? (uint32_t)myhist[d] : 0u;
         zh_lockstep_point();   // (every lane has read the counter before the first of its peers moves it)
         if (valid && rank == 0) myhist[d] = (uint16_t)(before + (uint32_t)zh_popc64(peers));
         zh_lockstep_point();
         o4[u] = before + rank;
      }
   }
   ZH_MFG_LAP(16);
   zh_sync_lds();
   ZH_MFG_LAP(17);
   uint32_t c[ZH_MF_WAVES];
   if (tid < 256u) {
      uint32_t t = 0;
#pragma unroll
      for (uint32_t w2 = 0; w2 < ZH_MF_WAVES; w2++) {
         c[w2] = L.hist[w2 * 256u + tid];
         t += c[w2];
      }
      L.tot[tid] = t;
   }
   ZH_MFG_LAP(18);
   zh_sync_lds();
   ZH_MFG_LAP(19);
   if (tid < 256u) {
      // every one of the four waves scans all 256 totals (four per lane) and keeps the 64 it needs: no barrier between scan and use
      uint32_t v[4], t = 0;
#pragma unroll
      for (uint32_t q = 0; q < 4; q++) {
         v[q] = L.tot[lane * 4u + q];
         t += v[q];
      }
      uint32_t ex = zh_wave_excl_sum(t);
      if ((lane >> 4) == wave) {   // digits 4 lane .. 4 lane + 3 belong to wave (4 lane) >> 6
#pragma unroll
         for (uint32_t q = 0; q < 4; q++) {
            L.cur[lane * 4u + q] = ex;
            ex += v[q];
         }
      }
      zh_lockstep_sync();
      uint32_t run = L.cur[tid];
#pragma unroll
      for (uint32_t w2 = 0; w2 < ZH_MF_WAVES; w2++) {
         L.hist[w2 * 256u + tid] = (uint16_t)run;
         run += c[w2];
      }
      if (tid == 255u) L.misc[2] = run;   // everything that has a digit
   }
   ZH_MFG_LAP(20);
   zh_sync_lds();
   ZH_MFG_LAP(21);
#pragma unroll
   for (uint32_t u = 0; u < 4; u++) {
      if (u * 64u < stretch) {
         const uint32_t d = d4[u];
         if (d != 0xffffffffu) OUT[(uint32_t)myhist[d] + o4[u]] = e4[u];
      }
   }
   ZH_MFG_LAP(22);
   zh_sync_lds();
   ZH_MFG_LAP(23);
   return L.misc[2];
}

// The chunk X[0 .. n) — whole bigram classes, each ascending in position — to its part of the 6-gram order: S_out / P_out point at the
// chunk's first entry of the segment's order. Returns the number of entries written.
__device__ inline uint32_t zh_mfl_refine(const zh_mfl_t &L, const uint32_t *lwin32, uint32_t W, uint32_t n, uint32_t *S_out, uint2 *P_out, uint2 *tail, bool excl, uint64_t &mfg_t_) {
   const uint32_t n1 = zh_mfl_pass<2, 0>(L, lwin32, W, n, L.X, L.Y, tail, excl, mfg_t_);
   const uint32_t n2 = zh_mfl_pass<3, 3>(L, lwin32, W, n1, L.Y, L.X, tail, excl, mfg_t_);
   const uint32_t n3 = zh_mfl_pass<4, 4>(L, lwin32, W, n2, L.X, L.Y, tail, excl, mfg_t_);
   const uint32_t n4 = zh_mfl_pass<5, 5>(L, lwin32, W, n3, L.Y, L.X, tail, excl, mfg_t_);
   ZH_MFG_LAP(3);
   for (uint32_t idx = threadIdx.x; idx < n4; idx += ZH_MF_THREADS) {
      const uint32_t e = L.X[idx], pos = e & ZH_MFL_POS_MASK, id = e >> ZH_MFL_ID_SHIFT;
      bool head = idx == 0;
      if (!head) {
         const uint32_t q = L.X[idx - 1] & ZH_MFL_POS_MASK;
         head = zh_load32_at(lwin32, q) != zh_load32_at(lwin32, pos) || ((zh_load32_at(lwin32, q + 4u) ^ zh_load32_at(lwin32, pos + 4u)) & 0xffffu) != 0;
      }
      S_out[idx] = pos | (head ? ZH_MF_HEAD : 0u);   // the scan stops after consuming a marked entry
      P_out[idx] = make_uint2(L.D34[id], (uint32_t)L.D5[id]);
   }
   zh_sync_lds();   // the chunk's arrays are free again
   ZH_MFG_LAP(4);
   return n4;
}

// A class too large for a chunk (zh_mf_group_big): the entries [0, n0) of SBc (its range of the bigram order) through the generic passes in
// HBM. Returns the number of entries written to S_out / P_out.
__device__ inline uint32_t zh_mfl_oversized(const uint8_t *gwin, uint32_t W, uint32_t n0, uint32_t *SBc, uint32_t *S_out, uint2 *P_out, uint2 *tail, uint32_t *pay,
                                            uint32_t *hist, uint32_t *wave_tot) {
   // scratch, all in the workgroup's `pay` (3 x sort_stride words; n0 <= 0.72 W: a class of a bigram of two different bytes holds at most every
   // other position, one of two equal bytes lost its run-interior positions): the pong T, the payload Pa, and the final pairs P2 — whose
   // first half serves as the payload Pb before
   uint32_t *T = pay, *Pa = pay + n0, *Pb = pay + 2u * n0;
   uint2 *P2 = (uint2 *)(pay + 2u * n0);
   const uint32_t *cursor = wave_tot + ZH_MF_WAVES + 1 + ZH_MF_WAVES * 256;   // zh_mf_sort_pass leaves the end of digit d's run in cursor[d]
   zh_mf_sort_pass<4>(gwin, gwin, n0, SBc, T, hist, wave_tot, W);
   zh_mf_sort_pass<5, false, -1, 1, 3>(gwin, gwin, n0, T, SBc, hist, wave_tot, W, nullptr, 0, nullptr, nullptr, nullptr, Pb, nullptr, nullptr, tail);
   const uint32_t n2 = cursor[255];
   zh_mf_sort_pass<6, false, -1, 1, 4>(gwin, gwin, n2, SBc, T, hist, wave_tot, W, nullptr, 0, nullptr, nullptr, Pb, Pa, nullptr, nullptr, tail);
   const uint32_t n3 = cursor[255];
   zh_mf_sort_pass<7, false, -1, 2, 5>(gwin, gwin, n3, T, SBc, hist, wave_tot, W, nullptr, 0, nullptr, nullptr, Pa, nullptr, nullptr, P2, tail);
   const uint32_t n4 = cursor[255];
   const uint32_t *lwin32 = (const uint32_t *)gwin;
   for (uint32_t idx = threadIdx.x; idx < n4; idx += ZH_MF_THREADS) {
      const uint32_t pos = SBc[idx];
      bool head = idx == 0;
      if (!head) {
         const uint32_t q = SBc[idx - 1];
         head = zh_load32_at(lwin32, q) != zh_load32_at(lwin32, pos) || ((zh_load32_at(lwin32, q + 4u) ^ zh_load32_at(lwin32, pos + 4u)) & 0xffffu) != 0;
      }
      S_out[idx] = pos | (head ? ZH_MF_HEAD : 0u);
      P_out[idx] = P2[idx];
   }
   __threadfence_block();
   __syncthreads();
   return n4;
}

// the window is staged in LDS at dyn_lds (W bytes); SA receives the 6-gram order, prev the distances next to it
__device__ inline void zh_mf_group_body_lds(const uint8_t *win, uint32_t *dyn_lds, uint32_t W, uint32_t Qn, uint32_t *SA, uint32_t *SB, uint2 *prev, uint32_t *pay,
                                            uint64_t pay_stride, uint32_t *runs, uint32_t *notes /* ZH_MFL_NOTE_WORDS words behind the run table */, uint32_t *hist, uint32_t *wave_tot, int stop, uint32_t cap_limit, uint64_t &mfg_t_) {
   const uint32_t *lwin32 = dyn_lds;
   const uint8_t *gwin = (const uint8_t *)dyn_lds;
   const uint32_t tid = threadIdx.x;
   const uint32_t M3 = min(Qn, W >= 3 ? W - 2 : 0u);   // positions that start a trigram
   const zh_mfl_t L = zh_mfl_layout(dyn_lds, W, cap_limit);
   uint32_t nover = 0;   // classes too large for a chunk, noted for zh_mf_group_big
   if (stop == 1) return;
   ZH_MFG_LAP(0);
   ZH_MFG_COUNT(15, 1);
   if (M3 != 0 && M3 <= L.cap) {
      // the whole window is one chunk: the bigram order is two more passes in LDS
      for (uint32_t k = tid; k < M3; k += ZH_MF_THREADS) L.X[k] = k | (k << ZH_MFL_ID_SHIFT);
      zh_sync_lds();
      zh_mfl_pass<1, 0>(L, lwin32, W, M3, L.X, L.Y, prev, false, mfg_t_);
      zh_mfl_pass<0, 0>(L, lwin32, W, M3, L.Y, L.X, prev, false, mfg_t_);
      zh_mfl_refine(L, lwin32, W, M3, SA, prev, prev, false, mfg_t_);
   }
   else if (M3 != 0) {
      uint32_t *hist2 = hist + ZH_MF_WAVES * 256 + ZH_MF_WAVES + 1;
      // (run-interior positions get no digit in the first pass and drop out: zh_mf_run_interior)
      zh_mf_sort_pass<10, false, 11>(gwin, gwin, M3, nullptr, SA, hist, wave_tot, W, hist2, M3);   // by byte 1; elements position | byte 0 << 24
      const uint32_t Mb = (wave_tot + ZH_MF_WAVES + 1 + ZH_MF_WAVES * 256)[255];                  // what it wrote (the end of the last digit's run: cursor[255])
      const bool excl = Mb != M3;                                                                // this window has run-interior positions
      zh_mf_sort_pass<11, true>(gwin, gwin, Mb, SA, SB, hist2, wave_tot, W);                        // by byte 0: SB = the bigram order, Mb entries
      if (stop == 2) return;
      ZH_MFG_LAP(1);
      uint32_t s = 0, out_base = 0;
      uint32_t en[4];
#define ZH_MFL_REQUEST(from_)                                                          \
   do {                                                                               \
      _Pragma("unroll") for (uint32_t j = 0; j < 4; j++) {                            \
         const uint32_t k_ = tid + j * ZH_MF_THREADS;                                 \
         en[j] = (k_ < L.cap && (from_) + k_ < Mb) ? SB[(from_) + k_] : 0u;           \
      }                                                                               \
   } while (0)
      ZH_MFL_REQUEST(0u);
      while (s < Mb) {
         const uint32_t nload = min(L.cap, Mb - s);
#pragma unroll
         for (uint32_t j = 0; j < 4; j++) {
            const uint32_t k = tid + j * ZH_MF_THREADS;
            if (k < nload) L.X[k] = (en[j] & ZH_MFL_POS_MASK) | (k << ZH_MFL_ID_SHIFT);
         }
         if (tid == 0) {
            L.misc[0] = 0;
            L.misc[1] = 0xffffffffu;
         }
         zh_sync_lds();
         // the chunk ends at the last class boundary among the loaded elements (or with the order)
#pragma unroll
         for (uint32_t j = 0; j < 4; j++) {
            const uint32_t k = tid + j * ZH_MF_THREADS;
            if (k >= 1 && k < nload) {
               const uint32_t a = zh_load32_at(lwin32, L.X[k] & ZH_MFL_POS_MASK), b = zh_load32_at(lwin32, L.X[k - 1] & ZH_MFL_POS_MASK);
               if ((a ^ b) & 0xffffu) atomicMax(&L.misc[0], k);
            }
         }
         zh_sync_lds();
         const uint32_t n = (s + nload == Mb) ? nload : L.misc[0];
         if (n == 0) {
            ZH_MFG_LAP(2);
            // one class fills the chunk and goes on: its end is the first entry of another bigram, found in two rounds of probes
            const uint32_t big0 = zh_load32_at(lwin32, L.X[0] & ZH_MFL_POS_MASK) & 0xffffu;
            const uint32_t rem = Mb - s;
            const uint32_t step = (rem + ZH_MF_THREADS - 1u) / ZH_MF_THREADS;
            {
               const uint32_t i = min((tid + 1u) * step, rem);
               const bool differs = i >= rem || (zh_load32_at(lwin32, SB[s + i]) & 0xffffu) != big0;
               if (differs) atomicMin(&L.misc[1], i);
            }
            zh_sync_lds();
            const uint32_t hi_i = L.misc[1], lo_i = hi_i - min(hi_i, step);   // the class ends in (lo_i, hi_i]
            zh_sync_lds();
            if (tid < hi_i - lo_i) {
               const uint32_t i = lo_i + 1u + tid;
               const bool differs = i >= rem || (zh_load32_at(lwin32, SB[s + i]) & 0xffffu) != big0;
               if (differs) atomicMin(&L.misc[1], i);
            }
            zh_sync_lds();
            const uint32_t n0 = L.misc[1];
            zh_sync_lds();   // (everybody has read it before the next round resets it)
            // The class is refined through HBM by a kernel of its own after this one (zh_mf_group_big: its four generic passes in here cost this
            // kernel its registers, 76 bytes of scratch per lane): it is noted — start, size, where its entries go — and its part of the order is
            // left free. How many entries that is: all but those of the last five window positions (the 6-gram order has none of them).
            uint32_t nv = n0;
            for (uint32_t pl = (W >= 5 ? W - 5 : 0u); pl < min(W, Qn); pl++)
               if (pl + 2u < W && (zh_load32_at(lwin32, pl) & 0xffffu) == big0 && !zh_mf_run_interior(gwin, pl, W)) nv--;
            if (tid == 0 && nover < ZH_MFL_MAX_NOTES) {
               uint32_t *note = notes + 1u + 2u * nover;
               note[0] = s | (n0 << 17);            // start (17 bits) | low 15 bits of the size
               note[1] = out_base | ((n0 >> 15) << 17);   // where its entries go | the size's high bits
            }
            nover++;
            out_base += nv;
            s += n0;
            ZH_MFG_LAP(5);
            ZH_MFG_COUNT(13, 1);
            ZH_MFG_COUNT(14, n0);
            ZH_MFL_REQUEST(s);
            continue;
         }
         ZH_MFL_REQUEST(s + n);   // the next chunk's elements are on their way while this one is refined
         ZH_MFG_LAP(2);
         ZH_MFG_COUNT(12, 1);
         out_base += zh_mfl_refine(L, lwin32, W, n, SA + out_base, prev + out_base, prev, excl, mfg_t_);
         s += n;
      }
#undef ZH_MFL_REQUEST
      ZH_MFG_LAP(5);
      // the run-interior positions: entries of their own at the top of the order (the sorted part ends exactly where they begin: together
      // they are the M6 positions with six bytes ahead), each marked as a class head — nothing walks into them, zh_mf_frontier takes their
      // frontier from the run table — with the position before them as the nearest earlier occurrence of their 3-, 4- and 5-gram
      if (excl) {
         const uint32_t M6 = min(Qn, W >= 6 ? W - 5 : 0u);
         if (tid == 0) L.misc[3] = 0;
         zh_sync_lds();
         for (uint32_t base = 0; base < M6; base += ZH_MF_THREADS) {
            const uint32_t pi = base + tid;
            const bool inter = pi < M6 && zh_mf_run_interior(gwin, pi, W);
            const uint64_t m = zh_ballot(inter);
            uint32_t at = 0;
            if (m && (tid & 63u) == 0) at = zh_atomic_add_lds(&L.misc[3], (uint32_t)zh_popc64(m));
            at = zh_readfirstlane(at);
            if (inter) {
               const uint32_t slot = M6 - 1u - (at + zh_rank_below(m));
               SA[slot] = pi | ZH_MF_HEAD;
               prev[slot] = make_uint2(0u, 0u);
            }
         }
      }
   }
   if (stop == 5) return;
   __threadfence_block();
   __syncthreads();
   zh_mf_build_runs(win, gwin, W, Qn, pay, runs, hist, wave_tot, mfg_t_);   // (scratch for the run starts: the workgroup's `pay` — SB still holds the bigram order of the classes noted for zh_mf_group_big)
   if (tid == 0) notes[0] = min(nover, (uint32_t)ZH_MFL_MAX_NOTES);   // (behind what zh_mf_build_runs writes)
}
