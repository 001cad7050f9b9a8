// zh_matchfinder.h — stage 1 of the hot path on MI355X: the match rows of a max-block.
//
// Replaces zultra_build_suffix_array + zultra_skip_matches + zultra_find_all_matches
// (reference src/matchfinder.c:49-286, called at src/libzultra.c:287-293).
//
// The reference's rows have a closed form: for block position i, scan the earlier window
// positions p from nearest to farthest; whenever the match length L(p) = min(LCP(i,p), 258, windowEnd-i)
// is >= 3 and strictly longer than everything seen so far, p is a row entry; entries farther than 32768 are
// dropped; the 8 longest survive, longest first. No suffix array is needed for that — only "all earlier
// occurrences of my first three bytes, nearest first", which is what these two kernels build and scan:
//
//   zh_mf_group    one workgroup per max-block: stable LSD radix sort (3 passes x 8 bits, the three bytes
//                  themselves are the digits) of the window positions by their trigram. Afterwards every
//                  trigram class is a contiguous run of positions in ascending order — candidate lists are
//                  contiguous memory, no pointer chasing (hash chains would serialise on memory latency).
//   zh_mf_frontier waves pull 64-entry chunks of the sorted array from a workgroup counter (load balance:
//                  neighbouring entries belong to the same class and have near-equal candidate counts, so
//                  the lanes of a wave stay busy together). The scan is wave-synchronous: lane l's k-th candidate is
//                  lane l-1's (k-1)-th, so the candidate stream is passed up the lanes with one DPP wave shift per
//                  step and refilled from one coalesced 64-entry load per 64 steps; the window sits in LDS.
//
// HBM traffic per max-block: window read once (L2 serves the re-reads), 2x4 B per window position for the
// sort ping-pong, 32 B per block position for the rows.
#pragma once
#include <zh_platform.h>
#include "zh_common.h"

#define ZH_MF_THREADS 1024
#define ZH_MF_WAVES (ZH_MF_THREADS / 64)
#define ZH_MF_HEAD 0x80000000u       // sorted entry: first position of its trigram class
#define ZH_MF_POS_MASK 0x7fffffffu
#define ZH_MF_SENTINEL 0xffffffffu   // "no entry": marked, and farther than any legal distance

// ---------------------------------------------------------------------------------------------------------
// zh_mf_group
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(ZH_MF_THREADS)
zh_mf_group(const uint8_t *__restrict__ data, const zh_block_t *__restrict__ blocks, uint32_t *sort_a,
            uint32_t *sort_b, uint64_t sort_stride) {
   __shared__ uint32_t hist[ZH_MF_WAVES * 256];
   __shared__ uint32_t wave_tot[ZH_MF_WAVES];

   const zh_block_t blk = blocks[blockIdx.x];
   const uint8_t *win = data + blk.win_off;
   const uint32_t W = blk.prev + blk.n;
   const uint32_t M = W >= 3 ? W - 2 : 0;   // positions that start a trigram
   uint32_t *A = sort_a + (uint64_t)blockIdx.x * sort_stride;
   uint32_t *B = sort_b + (uint64_t)blockIdx.x * sort_stride;

   const uint32_t tid = threadIdx.x;
   const uint32_t lane = tid & 63, wave = tid >> 6;
   const uint32_t seg = (((M + ZH_MF_WAVES - 1) / ZH_MF_WAVES) + 63) & ~63u;
   const uint32_t lo = wave * seg;
   const uint32_t hi = min(M, lo + seg);
   const uint64_t lt_mask = (1ull << lane) - 1;

   for (int pass = 0; pass < 3; pass++) {
      const uint32_t *src = (pass == 1) ? A : B;          // pass 0 reads the identity permutation
      uint32_t *dst = (pass == 1) ? B : A;                // final order lands in A
      const uint32_t boff = 2 - (uint32_t)pass;            // least significant digit first

      for (uint32_t k = tid; k < ZH_MF_WAVES * 256; k += ZH_MF_THREADS) hist[k] = 0;
      __syncthreads();

      // per-wave digit histogram of the wave's contiguous slice
      for (uint32_t base = lo; base < hi; base += 64) {
         uint32_t idx = base + lane;
         if (idx < hi) {
            uint32_t pos = pass ? src[idx] : idx;
            atomicAdd(&hist[wave * 256 + win[pos + boff]], 1u);
         }
      }
      __syncthreads();

      // exclusive scan in (digit, wave) order: entry e = digit*16 + wave; each thread owns 4 entries
      {
         uint32_t v[4], s = 0;
         for (int q = 0; q < 4; q++) {
            uint32_t e = tid * 4 + (uint32_t)q;
            v[q] = hist[(e & 15) * 256 + (e >> 4)];
            s += v[q];
         }
         uint32_t ex = zh_wave_excl_sum(s);
         if (lane == 63) wave_tot[wave] = ex + s;
         __syncthreads();
         uint32_t pre = 0;
         for (uint32_t w2 = 0; w2 < wave; w2++) pre += wave_tot[w2];
         ex += pre;
         for (int q = 0; q < 4; q++) {
            uint32_t e = tid * 4 + (uint32_t)q;
            hist[(e & 15) * 256 + (e >> 4)] = ex;
            ex += v[q];
         }
      }
      __syncthreads();

      // stable scatter: each wave walks its slice in order, 64 positions per step
      for (uint32_t base = lo; base < hi; base += 64) {
         uint32_t idx = base + lane;
         bool valid = idx < hi;
         uint32_t pos = 0, d = 0;
         if (valid) {
            pos = pass ? src[idx] : idx;
            d = win[pos + boff];
         }
         uint32_t slot = valid ? hist[wave * 256 + d] : 0;
         uint64_t peers = zh_ballot(valid);
         for (int bit = 0; bit < 8; bit++) {
            bool one = (d >> bit) & 1u;
            uint64_t m = zh_ballot(valid && one);
            peers &= one ? m : ~m;
         }
         if (valid) {
            dst[slot + (uint32_t)zh_popc64(peers & lt_mask)] = pos;
            if ((peers & lt_mask) == 0) hist[wave * 256 + d] = slot + (uint32_t)zh_popc64(peers);
         }
         zh_ballot(true);   // orders the counter update before the next step's read (lock-step anyway on the GPU)
      }
      __syncthreads();
   }
   // mark the first entry of every trigram class (bit 31): the scan stops after consuming a marked entry
   __threadfence_block();
   __syncthreads();
   for (uint32_t idx = tid; idx < M; idx += ZH_MF_THREADS) {
      const uint32_t pos = A[idx] & ZH_MF_POS_MASK;
      bool head = idx == 0;
      if (!head) {
         const uint32_t q = A[idx - 1] & ZH_MF_POS_MASK;
         head = win[pos] != win[q] || win[pos + 1] != win[q + 1] || win[pos + 2] != win[q + 2];
      }
      if (head) A[idx] = pos | ZH_MF_HEAD;
   }
}

// ---------------------------------------------------------------------------------------------------------
// zh_mf_frontier
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t zh_trigram(const uint8_t *p) {
   return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
}

// 4 bytes at an arbitrary byte offset x of a dword-aligned buffer: two aligned dword reads + funnel shift
// (the buffer must be readable up to 7 bytes past x).
__device__ __forceinline__ uint32_t zh_load32_at(const uint32_t *w32, uint32_t x) {
   const uint32_t lo = w32[x >> 2], hi = w32[(x >> 2) + 1];
   const uint32_t sh = (x & 3u) * 8u;
   return (uint32_t)((((uint64_t)hi << 32) | lo) >> sh);
}

// LDS_WIN: the whole window (<= ZH_MF_LDS_WINDOW bytes: 64 KiB max-blocks + 32 KiB history) is staged in LDS once
// per workgroup with coalesced dword loads, so the byte probes of the scan hit the 160 KiB LDS instead of L1/L2.
#define ZH_MF_LDS_WINDOW 98304

template <bool LDS_WIN>
__global__ void __launch_bounds__(ZH_MF_THREADS)
zh_mf_frontier(const uint8_t *__restrict__ data, const zh_block_t *__restrict__ blocks,
               const uint32_t *__restrict__ sorted, uint64_t sort_stride, zh_match_t *match,
               uint64_t match_stride) {
   __shared__ uint32_t next_chunk;
   __shared__ uint32_t lwin32[LDS_WIN ? (ZH_MF_LDS_WINDOW / 4 + 4) : 1];

   const zh_block_t blk = blocks[blockIdx.x];
   const uint8_t *gwin = data + blk.win_off;
   const uint32_t prev = blk.prev;
   const uint32_t W = blk.prev + blk.n;
   const uint32_t M = W >= 3 ? W - 2 : 0;
   const uint32_t *S = sorted + (uint64_t)blockIdx.x * sort_stride;
   zh_match_t *rows = match + (uint64_t)blockIdx.x * match_stride;   // row r = block position prev + r
   const uint32_t lane = threadIdx.x & 63;
   const uint8_t *win = gwin;

   if (LDS_WIN) {
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
      win = lw;
   }
   if (threadIdx.x == 0) next_chunk = 0;
   __syncthreads();

   // the last two window positions cannot start a match (matchfinder.c:71: LCP bounded by the window end)
   if (threadIdx.x < 2 && W >= 1) {
      uint32_t i = W - 1 - threadIdx.x;
      if (W >= 1 + threadIdx.x && i >= prev) {
         uint2 *r = (uint2 *)(rows + (uint64_t)(i - prev) * ZH_NMATCH);
         uint2 z;
         z.x = 0;
         z.y = 0;
         r[0] = z;
         r[1] = z;
         r[2] = z;
         r[3] = z;
      }
   }

   // Wave-synchronous sliding window over the sorted array: the wave owns sorted entries [c, c+64). At step k lane l
   // looks at entry c+l-1-k, which is what lane l-1 looked at one step earlier: the candidates travel up the lanes with
   // one DPP wave shift per step and enter at lane 0 from a 64-entry vector fetched with one coalesced load per 64 steps.
   // No per-candidate memory access except the one LDS byte probe.
   for (;;) {
      uint32_t c = 0;
      if (lane == 0) c = atomicAdd(&next_chunk, 64u);
      c = zh_readfirstlane(c);
      if (c >= M) break;

      const uint32_t t = c + lane;
      const uint32_t own = t < M ? S[t] : ZH_MF_SENTINEL;
      const uint32_t i = own & ZH_MF_POS_MASK;
      bool alive = t < M && i >= prev && !(own & ZH_MF_HEAD);   // a class head has no earlier occurrence
      const uint32_t maxlen = t < M ? min((uint32_t)ZH_MAX_MATCH, W - i) : 0;
      uint32_t cur = ZH_MIN_MATCH - 1;
      // what a candidate must match to beat `cur`: byte cur, or (LDS window, cur >= 3) the four bytes cur-3..cur —
      // a 100x sharper filter than one byte, so the divergent extension below runs for few candidates
      uint32_t ci = alive ? win[i + cur] : 0;
      uint32_t m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0, m5 = 0, m6 = 0, m7 = 0;   // length | offset<<16

      uint32_t cand = own;
      int64_t vbase = (int64_t)c - 64;                            // sorted index of lane 0 of the feed vector
      uint32_t vec = (vbase + lane >= 0) ? S[vbase + lane] : ZH_MF_SENTINEL;
      int vi = 63;
      while (zh_ballot(alive)) {
         // advance: entry c+l-1-k arrives at lane l; lane 0 takes the next entry below the chunk
         const uint32_t feed = zh_readlane(vec, vi);
         cand = zh_wave_shr1(cand, feed);
         if (--vi < 0) {
            vbase -= 64;
            vec = (vbase + lane >= 0) ? S[vbase + lane] : ZH_MF_SENTINEL;
            vi = 63;
         }
         if (alive) {
            const uint32_t p = cand & ZH_MF_POS_MASK;
            if (i - p > ZH_MAX_DIST)
               alive = false;                                    // also catches the sentinel
            else {
               bool pass;
               if (LDS_WIN) {
                  // one byte probe for everybody (few LDS bank conflicts), the 4-byte probe only for the ~10% that pass it
                  pass = cur < 3 || win[p + cur] == (ci >> 24);
                  if (pass && cur >= 3) pass = zh_load32_at(lwin32, p + cur - 3) == ci;
               }
               else
                  pass = win[p + cur] == ci;
               if (pass) {                                       // only then can it beat the incumbent
                  uint32_t l = ZH_MIN_MATCH;                    // the class guarantees the first three bytes
                  if (LDS_WIN) {
                     // four bytes per probe pair; bytes past the window end are garbage but l is clamped to maxlen
                     while (l < maxlen) {
                        const uint32_t x = zh_load32_at(lwin32, p + l) ^ zh_load32_at(lwin32, i + l);
                        if (x) {
                           l += (uint32_t)(__ffs((int)x) - 1) >> 3;
                           break;
                        }
                        l += 4;
                     }
                     l = min(l, maxlen);
                  }
                  else {
                     while (l < maxlen && win[p + l] == win[i + l]) l++;
                  }
                  if (l > cur) {
                     m7 = m6; m6 = m5; m5 = m4; m4 = m3; m3 = m2; m2 = m1; m1 = m0;
                     m0 = l | ((i - p) << 16);                    // offset 32768 needs all 16 bits
                     cur = l;
                     if (cur >= maxlen)
                        alive = false;
                     else
                        ci = LDS_WIN ? zh_load32_at(lwin32, i + cur - 3) : (uint32_t)win[i + cur];
                  }
               }
               if (cand & ZH_MF_HEAD) alive = false;             // that was the first occurrence of the class
            }
         }
      }
      if (t < M && i >= prev) {
         uint4 *r = (uint4 *)(rows + (uint64_t)(i - prev) * ZH_NMATCH);
         uint4 a, b2;
         a.x = m0; a.y = m1; a.z = m2; a.w = m3;
         b2.x = m4; b2.y = m5; b2.z = m6; b2.w = m7;
         r[0] = a;
         r[1] = b2;
      }
   }
}
