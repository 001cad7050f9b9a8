// zh_huffman.h — entropy-coding primitives of the hot path as device code (one wave = one alphabet owner).
//
// Replaces reference src/huffman/huffencoder.c (code lengths :157-270, length limit + canonical codes
// :279-375, code-length RLE :446-735, table trimming :400-406,:532-538) and src/huffman/huffutils.c:34-114.
// The arrays live in LDS. Two calling conventions:
//   *_wave   : called by all 64 lanes of one wave on that wave's own LDS data (synchronised with zh_wave_sync, so several
//              waves of a workgroup can run different instances at once); sorting is lane-parallel (rank sort), the inherently serial
//              two-queue merge runs on lane 0;
//   *_lane   : plain single-lane code on a private LDS slice, for small alphabets (the 19-symbol code-length
//              alphabet) where several independent instances run in different lanes at once.
#pragma once
#include <zh_platform.h>
#include "zh_common.h"

// ---------------------------------------------------------------------------------------------------------
// Moffat-Katajainen in-place minimum-redundancy code lengths on an ascending weight array
// (huffencoder.c:198-255). Tie rule: an internal node is preferred only when strictly lighter than the
// next leaf. Single lane.
// ---------------------------------------------------------------------------------------------------------
__device__ inline void zh_mk_depths(int32_t *A, int n) {
   int leaf = 0, node = 0, t;
   for (t = 0; t < n - 1; t++) {
      int w = 0;
      for (int pick = 0; pick < 2; pick++) {
         if (leaf >= n || (node < t && A[node] < A[leaf])) {
            w += A[node];
            A[node] = t;
            node++;
         }
         else {
            w += A[leaf];
            leaf++;
         }
      }
      A[t] = w;
   }
   A[n - 2] = 0;
   for (t = n - 3; t >= 0; t--) A[t] = A[A[t]] + 1;
   int avail = 1, used = 0, depth = 0, next = n - 1;
   t = n - 2;
   while (avail > 0) {
      while (t >= 0 && A[t] == depth) {
         used++;
         t--;
      }
      while (avail > used) {
         A[next--] = depth;
         avail--;
      }
      avail = used << 1;
      depth++;
      used = 0;
   }
}

// Scratch for the wave-cooperative builders (per wave, LDS)
struct zh_huff_scratch_t {
   uint32_t keys[ZH_NLIT];
   uint32_t sorted[ZH_NLIT];
   int32_t A[ZH_NLIT];
   uint32_t count;
   uint32_t small[36];   // per code length: symbols / first index / first code (canonical code assignment)
};

// The same result as zh_mk_depths, by all 64 lanes of a wave on A[0..n) in LDS (n >= 2; `tmp` = n words of LDS scratch).
// Phase 1 (the two-queue merge, huffencoder.c:198-223) is serial and stays on lane 0; what follows it in the reference
// are two more serial walks over the array, each step a dependent LDS access (~75 cycles on gfx950) — they go wide:
//   phase 2 (:227-231)  depth of every internal node = depth of its parent + 1: pointer jumping, ceil(log2(depth)) rounds;
//   phase 3 (:235-254)  leaves per depth from the number of internal nodes per depth (a histogram), then every depth's
//                       leaves are written by the lanes at once.
__device__ inline void zh_mk_depths_wave(int32_t *A, int n, uint32_t *tmp) {
   const int lane = (int)zh_lane();
   if (lane == 0) {
      // The heads of the two queues and the entries behind them stay in registers: a step of the reference's loop reads both
      // heads from the array after every pick (two dependent LDS round trips per pick); here a queue's register pair is
      // refilled when it advances and the load has the other pick's time to land.
      const int inf = 0x7fffffff;
      int leaf = 0, node = 0;
      int lw0 = A[0], lw1 = A[1];          // n >= 2
      int nw0 = inf, nw1 = inf;            // node queue: empty
      for (int t = 0; t < n - 1; t++) {
         int w = 0;
#pragma unroll
         for (int pick = 0; pick < 2; pick++) {
            if (leaf >= n || (node < t && nw0 < lw0)) {   // an internal node only when strictly lighter than the next leaf (:203,215)
               w += nw0;
               A[node] = t;
               node++;
               nw0 = nw1;
               nw1 = node + 1 < t ? A[node + 1] : inf;
            }
            else {
               w += lw0;
               leaf++;
               lw0 = lw1;
               lw1 = leaf + 1 < n ? A[leaf + 1] : inf;
            }
         }
         A[t] = w;
         if (node == t)
            nw0 = w;                        // the queue was empty: the new node is its head
         else if (node + 1 == t)
            nw1 = w;
      }
   }
   zh_wave_sync();
   // ---- phase 2: A[t] = parent of internal node t (t <= n-3), node n-2 is the root -> A[t] = depth of t ------------
   const int ni = n - 1;   // internal nodes
   const int root = n - 2;
   int par[5], dep[5];     // ZH_NLIT <= 5 * 64
#pragma unroll
   for (int q = 0; q < 5; q++) {
      const int t = lane + 64 * q;
      par[q] = (t < root) ? A[t] : root;
      dep[q] = (t < root) ? 1 : 0;
   }
   for (;;) {
      bool moving = false;
#pragma unroll
      for (int q = 0; q < 5; q++) {
         const int t = lane + 64 * q;
         if (t < ni) {
            A[t] = par[q];
            tmp[t] = (uint32_t)dep[q];
            moving = moving || par[q] != root;
         }
      }
      if (!zh_ballot(moving)) break;   // (also orders the stores before the gathers below)
      zh_wave_sync();
      int pp[5], dd[5];
#pragma unroll
      for (int q = 0; q < 5; q++) {
         const int t = lane + 64 * q;
         pp[q] = root;
         dd[q] = 0;
         if (t < ni) {
            pp[q] = A[par[q]];
            dd[q] = (int)tmp[par[q]];
         }
      }
      zh_wave_sync();
#pragma unroll
      for (int q = 0; q < 5; q++) {
         dep[q] += dd[q];
         par[q] = pp[q];
      }
   }
   zh_wave_sync();
   // ---- phase 3: internal nodes per depth -> leaves per depth; the heaviest leaves (highest indices) are the shallowest ----
   for (int k = lane; k < n; k += 64) tmp[k] = 0;
   zh_wave_sync();
#pragma unroll
   for (int q = 0; q < 5; q++)
      if (lane + 64 * q < ni) atomicAdd(&tmp[dep[q]], 1u);
   zh_wave_sync();
   int avail = 1, next = n - 1;
   for (int depth = 0; avail > 0; depth++) {
      const int used = depth < n ? (int)tmp[depth] : 0;
      const int nleaf = avail - used;
      for (int k = lane; k < nleaf; k += 64) A[next - k] = depth;
      next -= nleaf;
      avail = used << 1;
   }
   zh_wave_sync();
}

// Sort of n unique 32-bit keys: sorted[0..n) ascending. All lanes call. `keys` and `sorted` are the scratch arrays of
// zh_huff_scratch_t, contiguous (2 x ZH_NLIT words); keys is destroyed. Up to 64 keys: rank sort (every lane counts the keys
// below its own). Above: a bitonic network over the next power of two, padded with all-ones — 45 compare-exchange stages of
// 4 pairs per lane for 512 entries against 288 x 5 counting steps.
__device__ inline void zh_rank_sort_wave(uint32_t *keys, uint32_t *sorted, int n) {
   const int lane = (int)zh_lane();
   if (n <= 64) {
      if (lane < n) {
         const uint32_t k = keys[lane];
         int rank = 0;
         for (int j = 0; j < n; j++) rank += (keys[j] < k) ? 1 : 0;
         sorted[rank] = k;
      }
      zh_wave_sync();
      return;
   }
   int P = 128;
   while (P < n) P <<= 1;                      // 128, 256 or 512 <= 2 * ZH_NLIT
   uint32_t *a = keys;                         // work area: keys[0..P), running into `sorted` for P = 512
   for (int e = n + lane; e < P; e += 64) a[e] = 0xFFFFFFFFu;
   zh_wave_sync();
   for (int size = 2; size <= P; size <<= 1) {
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
         for (int pr = lane; pr < (P >> 1); pr += 64) {
            const int lo = ((pr & ~(stride - 1)) << 1) | (pr & (stride - 1)), hi = lo + stride;
            const bool up = (lo & size) == 0;
            const uint32_t x = a[lo], y = a[hi];
            if ((x > y) == up) {
               a[lo] = y;
               a[hi] = x;
            }
         }
         zh_wave_sync();
      }
   }
   // the first n entries are the keys in order; `sorted` may overlap the work area: through registers
   uint32_t v[5];
#pragma unroll
   for (int q = 0; q < 5; q++) v[q] = lane + 64 * q < n ? a[lane + 64 * q] : 0u;
   zh_wave_sync();
#pragma unroll
   for (int q = 0; q < 5; q++)
      if (lane + 64 * q < n) sorted[lane + 64 * q] = v[q];
   zh_wave_sync();
}

// Collect (value<<9 | symbol) keys of the symbols with value != 0, in symbol order. Returns the count.
template <typename T>
__device__ inline int zh_collect_keys_wave(const T *value, int nsym, zh_huff_scratch_t *sc) {
   const int lane = (int)zh_lane();
   int n = 0;
   for (int base = 0; base < nsym; base += 64) {
      int s = base + lane;
      uint32_t v = (s < nsym) ? (uint32_t)value[s] : 0;
      uint64_t m = zh_ballot(v != 0);
      if (v != 0) sc->keys[n + zh_popc64(m & ((1ull << lane) - 1))] = (v << 9) | (uint32_t)s;
      n += zh_popc64(m);
   }
   zh_wave_sync();
   return n;
}

// huffencoder.c:157-270: code lengths without limit. `len` receives nsym entries. All lanes call.
__device__ inline void zh_huff_lengths_wave(const int32_t *freq, uint8_t *len, int nsym, zh_huff_scratch_t *sc) {
   const int lane = (int)zh_lane();
   int n = zh_collect_keys_wave(freq, nsym, sc);
   for (int s = lane; s < nsym; s += 64) len[s] = 0;
   zh_wave_sync();
   if (n <= 1) {
      if (lane == 0) len[0] = 1;   // huffencoder.c:263-267: symbol 0, whichever symbol was used
      zh_wave_sync();
      return;
   }
   zh_rank_sort_wave(sc->keys, sc->sorted, n);
   for (int e = lane; e < n; e += 64) sc->A[e] = (int32_t)(sc->sorted[e] >> 9);
   zh_wave_sync();
   zh_mk_depths_wave(sc->A, n, sc->keys);   // the keys have been sorted into sc->sorted: their array is free
   for (int e = lane; e < n; e += 64) len[sc->sorted[e] & 511u] = (uint8_t)sc->A[e];
   zh_wave_sync();
}

__device__ __forceinline__ uint32_t zh_bitrev16(uint32_t v, int nbits) {
   v = ((v & 0x5555u) << 1) | ((v & 0xaaaau) >> 1);
   v = ((v & 0x3333u) << 2) | ((v & 0xccccu) >> 2);
   v = ((v & 0x0f0fu) << 4) | ((v & 0xf0f0u) >> 4);
   v = ((v & 0x00ffu) << 8) | ((v & 0xff00u) >> 8);
   return v >> (16 - nbits);
}

// huffencoder.c:310-344 on a list of symbols ordered by (length, symbol). Single lane. Returns 0, or -1
// where the reference would walk past its array (:334; believed unreachable).
__device__ inline int zh_limit_lengths(uint8_t *len, const uint32_t *order, int n, int maxbits) {
   const int full = 1 << maxbits;
   int kraft = 0, i;
   for (i = n - 1; i >= 0; i--) {
      int s = (int)(order[i] & 511u);
      if (len[s] > maxbits) len[s] = (uint8_t)maxbits;
      kraft += full >> len[s];
   }
   for (i = n - 1; kraft > full && i >= 0; i--) {
      int s = (int)(order[i] & 511u);
      while (len[s] < maxbits && kraft > full) {
         len[s]++;
         kraft -= full >> len[s];
      }
   }
   for (i = 0; kraft < full; i++) {
      if (i >= n) return -1;
      int s = (int)(order[i] & 511u);
      while (kraft + (full >> len[s]) <= full) {
         kraft += full >> len[s];
         len[s]--;
      }
   }
   return 0;
}

// huffencoder.c:348-372: canonical codes along a (length, symbol)-ordered list, stored bit-reversed.
__device__ inline void zh_assign_codes(const uint8_t *len, uint16_t *code, const uint32_t *order, int n) {
   uint32_t c = 0;
   for (int i = 0; i < n; i++) {
      int s = (int)(order[i] & 511u);
      code[s] = (uint16_t)zh_bitrev16(c, len[s]);
      if (i + 1 < n) c = (c + 1) << (len[order[i + 1] & 511u] - len[s]);
   }
}

// The same codes as zh_assign_codes by all lanes of a wave: along a (length, symbol)-ordered list the reference's recurrence
// c' = (c + 1) << (next length - length) is the canonical code — first code of a length = (first code + count of the previous
// length) << 1 — so a symbol's code is its length's first code plus its rank inside the length. Lengths <= 15.
__device__ inline void zh_assign_codes_wave(const uint8_t *len, uint16_t *code, const uint32_t *order, int n, uint32_t *small /* 36 words */) {
   const int lane = (int)zh_lane();
   if (lane < 36) small[lane] = 0;
   zh_wave_sync();
   for (int i = lane; i < n; i += 64) atomicAdd(&small[len[order[i] & 511u]], 1u);   // [1..15]: symbols per length
   zh_wave_sync();
   if (lane == 0) {
      uint32_t first_index = 0, first_code = 0, prev_count = 0;
      bool started = false;
      for (int l = 1; l <= 15; l++) {
         const uint32_t cnt = small[l];
         // lengths before the shortest one in use start the code at 0 (zh_assign_codes: c = 0 at the first symbol)
         first_code = started ? (first_code + prev_count) << 1 : 0u;
         started = started || cnt != 0;
         small[16 + l] = first_index;
         prev_count = cnt;
         first_index += cnt;
         small[l] = first_code;   // the count has been consumed: its slot takes the first code
      }
   }
   zh_wave_sync();
   for (int i = lane; i < n; i += 64) {
      const int sym = (int)(order[i] & 511u);
      const int l = len[sym];
      code[sym] = (uint16_t)zh_bitrev16(small[l] + ((uint32_t)i - small[16 + l]), l);
   }
   zh_wave_sync();
}

// huffencoder.c:279-375: lengths, limit to maxbits, canonical codes. All lanes call. Returns 0 / -1 (uniform).
__device__ __forceinline__ int zh_huff_build_wave(const int32_t *freq, uint8_t *len, uint16_t *code, int nsym, int maxbits,
                                         zh_huff_scratch_t *sc) {
   const int lane = (int)zh_lane();
   zh_huff_lengths_wave(freq, len, nsym, sc);
   int n = zh_collect_keys_wave(len, nsym, sc);
   zh_rank_sort_wave(sc->keys, sc->sorted, n);
   int over = (n > 0 && (int)(sc->sorted[n - 1] >> 9) > maxbits) ? 1 : 0;
   int rc = 0;
   if (over) {
      if (lane == 0) sc->count = (uint32_t)zh_limit_lengths(len, sc->sorted, n, maxbits);
      zh_wave_sync();
      rc = (int)sc->count;
      zh_wave_sync();
      n = zh_collect_keys_wave(len, nsym, sc);   // huffencoder.c:344: order again after the repair
      zh_rank_sort_wave(sc->keys, sc->sorted, n);
   }
   if (n > 0) zh_assign_codes_wave(len, code, sc->sorted, n, sc->small);
   return rc;
}

// huffencoder.c:107-148: every symbol has a length (static tables). All lanes call.
__device__ inline void zh_huff_static_codes_wave(const uint8_t *len, uint16_t *code, int nsym, zh_huff_scratch_t *sc) {
   const int lane = (int)zh_lane();
   for (int s = lane; s < nsym; s += 64) sc->keys[s] = ((uint32_t)len[s] << 9) | (uint32_t)s;
   zh_wave_sync();
   zh_rank_sort_wave(sc->keys, sc->sorted, nsym);
   zh_assign_codes_wave(len, code, sc->sorted, nsym, sc->small);
}

// huffencoder.c:532-538
__device__ inline int zh_defined_count(const uint8_t *len, int nsym, int min_syms) {
   int i = nsym;
   while (i > min_syms && !len[i - 1]) i--;
   return i;
}

// ---------------------------------------------------------------------------------------------------------
// Code-length alphabet (19 symbols): everything single-lane on a private slice, so that several mask
// candidates can be evaluated by different lanes at the same time.
// ---------------------------------------------------------------------------------------------------------
struct zh_cl_t {
   int32_t freq[ZH_NCL];
   uint8_t len[ZH_NCL];
   uint16_t code[ZH_NCL];
   uint32_t keys[ZH_NCL];   // work arrays of the single-lane builds below: in the slice (LDS), not in private memory — arrays indexed at run
   int32_t A[ZH_NCL];       // time live in scratch, and a kernel with scratch pays a memory round trip per access
};

__device__ __forceinline__ int zh_cl_order(int k) {
   // RFC 1951 §3.2.7 order: 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
   const uint64_t lo = 0x0B050A0609070800ull | 0x0000000000000000ull;   // entries 3..10  (0,8,7,9,6,10,5,11)
   if (k < 3) return 16 + k;
   if (k < 11) return (int)((lo >> (8 * (k - 3))) & 0xff);
   const uint64_t hi = 0x0F010E020D030C04ull;                              // entries 11..18 (4,12,3,13,2,14,1,15)
   return (int)((hi >> (8 * (k - 11))) & 0xff);
}

// ordered list of (value<<9|sym) for the symbols with value != 0, by insertion sort. Returns count.
template <typename T>
__device__ inline int zh_small_sorted_keys(const T *value, int nsym, uint32_t *out) {
   int n = 0;
   for (int s = 0; s < nsym; s++) {
      uint32_t v = (uint32_t)value[s];
      if (!v) continue;
      uint32_t k = (v << 9) | (uint32_t)s;
      int j = n++;
      while (j > 0 && out[j - 1] > k) {
         out[j] = out[j - 1];
         j--;
      }
      out[j] = k;
   }
   return n;
}

__device__ inline void zh_cl_lengths_lane(zh_cl_t *h) {
   uint32_t *keys = h->keys;
   int32_t *A = h->A;
   int n = zh_small_sorted_keys(h->freq, ZH_NCL, keys);
   for (int s = 0; s < ZH_NCL; s++) h->len[s] = 0;
   if (n <= 1) {
      h->len[0] = 1;
      return;
   }
   for (int i = 0; i < n; i++) A[i] = (int32_t)(keys[i] >> 9);
   zh_mk_depths(A, n);
   for (int i = 0; i < n; i++) h->len[keys[i] & 511u] = (uint8_t)A[i];
}

__device__ inline int zh_cl_build_lane(zh_cl_t *h, int maxbits) {
   uint32_t *keys = h->keys;
   zh_cl_lengths_lane(h);
   int n = zh_small_sorted_keys(h->len, ZH_NCL, keys);
   if (n > 0 && (int)(keys[n - 1] >> 9) > maxbits) {
      if (zh_limit_lengths(h->len, keys, n, maxbits) < 0) return -1;
      n = zh_small_sorted_keys(h->len, ZH_NCL, keys);
   }
   if (n > 0) zh_assign_codes(h->len, h->code, keys, n);
   return 0;
}

// huffencoder.c:400-406
__device__ inline int zh_cl_raw_table_size(const zh_cl_t *h) {
   int i = ZH_NCL;
   while (i > 4 && !h->len[zh_cl_order(i - 1)]) i--;
   return i;
}

// One tokenizer for huffencoder.c:446-522 (count), :549-628 (size), :640-735 (write).
// mask: 1 = code 16 usable, 2 = code 17, 4 = code 18, 8 = no 4+3 split of a 7-repeat, 16 = no 4+4 split of 8.
template <typename Sink>
__device__ inline void zh_cl_tokenize(const uint8_t *lens, int n, unsigned mask, Sink &sink) {
   int i = 0;
   while (i < n) {
      const int v = lens[i];
      int run = 1;
      while (i + run < n && lens[i + run] == v) run++;
      if (v == 0) {
         if (run >= 3) {
            while (run >= 11 && (mask & 4)) {
               int take = run > 138 ? 138 : run;
               sink.put(18, take - 11, 7);
               run -= take;
               i += take;
            }
            while (run >= 3 && (mask & 2)) {
               int take = run > 10 ? 10 : run;
               sink.put(17, take - 3, 3);
               run -= take;
               i += take;
            }
            if (run) {
               sink.put(0, 0, 0);
               i++;
            }
         }
         else {
            sink.put(0, 0, 0);
            i++;
         }
      }
      else {
         sink.put(v > 15 ? 15 : v, 0, 0);
         i++;
         run--;
         if (run == 7 && (mask & 1) && !(mask & 8)) {
            sink.put(16, 1, 2);
            sink.put(16, 0, 2);
            run = 0;
            i += 7;
         }
         else if (run == 8 && (mask & 1) && !(mask & 16)) {
            sink.put(16, 1, 2);
            sink.put(16, 1, 2);
            run = 0;
            i += 8;
         }
         while (run >= 3 && (mask & 1)) {
            int take = run > 6 ? 6 : run;
            sink.put(16, take - 3, 2);
            run -= take;
            i += take;
         }
      }
   }
}

// The same tokenizer over a precomputed run list (entry = value | run length << 6, runs of equal code lengths <= 63): the
// symbol-by-symbol version above rescans the rest of a run after every token it emits; on the run list every step emits
// a token. Each iteration of the outer loops below is one iteration of the loop above applied to what is left of the run.
template <typename Sink>
__device__ inline void zh_cl_tokenize_runs(const uint16_t *runs, int nruns, unsigned mask, Sink &sink) {
   for (int q = 0; q < nruns; q++) {
      const int v = runs[q] & 63;
      int run = runs[q] >> 6;
      if (v == 0) {
         while (run > 0) {
            if (run >= 3) {
               while (run >= 11 && (mask & 4)) {
                  int take = run > 138 ? 138 : run;
                  sink.put(18, take - 11, 7);
                  run -= take;
               }
               while (run >= 3 && (mask & 2)) {
                  int take = run > 10 ? 10 : run;
                  sink.put(17, take - 3, 3);
                  run -= take;
               }
               if (run) {
                  sink.put(0, 0, 0);
                  run--;
               }
            }
            else {
               sink.put(0, 0, 0);
               run--;
            }
         }
      }
      else {
         while (run > 0) {
            sink.put(v > 15 ? 15 : v, 0, 0);
            run--;
            if (run == 7 && (mask & 1) && !(mask & 8)) {
               sink.put(16, 1, 2);
               sink.put(16, 0, 2);
               run = 0;
            }
            else if (run == 8 && (mask & 1) && !(mask & 16)) {
               sink.put(16, 1, 2);
               sink.put(16, 1, 2);
               run = 0;
            }
            while (run >= 3 && (mask & 1)) {
               int take = run > 6 ? 6 : run;
               sink.put(16, take - 3, 2);
               run -= take;
            }
         }
      }
   }
}

// run list of n code lengths (single lane); returns the number of runs (<= n)
__device__ inline int zh_cl_make_runs(const uint8_t *lens, int n, uint16_t *runs) {
   int nruns = 0, i = 0;
   while (i < n) {
      const int v = lens[i];
      int run = 1;
      while (i + run < n && lens[i + run] == v) run++;
      // v <= 63 (the cost estimates price unlimited lengths: a 2 MiB sub-block of text reaches depth 20), run <= 320
      runs[nruns++] = (uint16_t)(v | (run << 6));
      i += run;
   }
   return nruns;
}

// The run list by all lanes of a wave (same entries as zh_cl_make_runs); returns the number of runs (uniform).
// `starts` = n words of LDS scratch.
__device__ inline int zh_cl_make_runs_wave(const uint8_t *lens, int n, uint16_t *runs, uint32_t *starts) {
   const int lane = (int)zh_lane();
   int nruns = 0;
   for (int base = 0; base < n; base += 64) {
      const int i = base + lane;
      const bool head = i < n && (i == 0 || lens[i] != lens[i - 1]);
      const uint64_t m = zh_ballot(head);
      if (head) starts[nruns + zh_popc64(m & ((1ull << lane) - 1))] = (uint32_t)i;
      nruns += zh_popc64(m);
   }
   zh_wave_sync();
   for (int q = lane; q < nruns; q += 64) {
      const int i0 = (int)starts[q], i1 = q + 1 < nruns ? (int)starts[q + 1] : n;
      runs[q] = (uint16_t)(lens[i0] | ((i1 - i0) << 6));
   }
   zh_wave_sync();
   return nruns;
}

struct zh_cl_count_sink {
   zh_cl_t *h;
   __device__ __forceinline__ void put(int sym, int, int) { h->freq[sym]++; }
};
struct zh_cl_size_sink {
   const zh_cl_t *h;
   int bits;
   __device__ __forceinline__ void put(int sym, int, int xbits) { bits += h->len[sym] + xbits; }
};

__device__ inline void zh_cl_reset(zh_cl_t *h) {
   for (int s = 0; s < ZH_NCL; s++) {
      h->freq[s] = 0;
      h->len[s] = 0;
      h->code[s] = 0;
   }
}

// blockdeflate.c:594-613: header cost of a (lit, dist) pair — code-length alphabet histogrammed with mask 7,
// sized with mask 31, its own lengths from the *unlimited* estimate. Single lane; lens = concatenated lengths.
__device__ inline int zh_table_cost_lane(const uint8_t *lens, int n, zh_cl_t *h, uint16_t *runs /* LDS scratch, n entries */) {
   const int nruns = zh_cl_make_runs(lens, n, runs);
   zh_cl_reset(h);
   zh_cl_count_sink cs{h};
   zh_cl_tokenize_runs(runs, nruns, 7, cs);
   zh_cl_lengths_lane(h);
   zh_cl_size_sink ss{h, 0};
   zh_cl_tokenize_runs(runs, nruns, 31, ss);
   return 5 + 5 + 4 + 3 * zh_cl_raw_table_size(h) + ss.bits;
}

// The same value by all lanes of a wave: the tokens of a run depend on nothing but the run, so the two tokenizer passes go
// one run per lane — symbol counts through LDS atomics, bits through a wave sum; only the 19-symbol length build between
// them stays on one lane. `scratch` = n words of LDS. Returns the same value in every lane.
struct zh_cl_atomic_count_sink {
   zh_cl_t *h;
   __device__ __forceinline__ void put(int sym, int, int) { atomicAdd(&h->freq[sym], 1); }
};
__device__ inline int zh_table_cost_wave(const uint8_t *lens, int n, zh_cl_t *h, uint32_t *scratch) {
   const int lane = (int)zh_lane();
   uint16_t *runs = (uint16_t *)scratch;                 // n entries of 2 bytes in the first half
   const int nruns = zh_cl_make_runs_wave(lens, n, runs, scratch + (n + 1) / 2);
   if (lane == 0) zh_cl_reset(h);
   zh_wave_sync();
   {
      zh_cl_atomic_count_sink cs{h};
      for (int q = lane; q < nruns; q += 64) zh_cl_tokenize_runs(runs + q, 1, 7, cs);
   }
   zh_wave_sync();
   if (lane == 0) zh_cl_lengths_lane(h);
   zh_wave_sync();
   zh_cl_size_sink ss{h, 0};
   for (int q = lane; q < nruns; q += 64) zh_cl_tokenize_runs(runs + q, 1, 31, ss);
   const int bits = (int)zh_wave_sum((uint32_t)ss.bits);
   const int r = 5 + 5 + 4 + 3 * zh_cl_raw_table_size(h) + bits;
   zh_wave_sync();
   return r;
}

// ---------------------------------------------------------------------------------------------------------
// huffutils.c:34-114 (zopfli's OptimizeHuffmanForRle), single lane, `keep` = scratch of `length` bytes (4-byte aligned).
// Both walks are sequential in their state (the run so far; stride, limit and sum) but every value they read is an ORIGINAL count — the fills
// only ever write below the position the walk has reached — so the counts (and the keep flags) are fetched eight positions at a time, the loads
// of a chunk in flight together, and the eight steps run on registers: a step was three dependent LDS round trips before.
// ---------------------------------------------------------------------------------------------------------
__device__ inline void zh_smooth_for_rle_lane(int length, int32_t *counts, uint8_t *keep) {
   while (length > 0 && counts[length - 1] == 0) length--;
   if (length == 0) return;
   for (int i = 0; i < length; i += 4) *(uint32_t *)(keep + i) = 0u;   // (whole words: the scratch is longer than any alphabet)
   {
      uint32_t symbol = (uint32_t)counts[0];
      int stride = 0;
      for (int base = 0; base <= length; base += 8) {
         uint32_t c[8];
#pragma unroll
         for (int j = 0; j < 8; j++) c[j] = (uint32_t)counts[min(base + j, length - 1)];
#pragma unroll
         for (int j = 0; j < 8; j++) {
            const int i = base + j;
            if (i <= length) {
               if (i == length || c[j] != symbol) {
                  if ((symbol == 0 && stride >= 5) || (symbol != 0 && stride >= 7))
                     for (int k = 0; k < stride; k++) keep[i - k - 1] = 1;
                  stride = 1;
                  if (i != length) symbol = c[j];
               }
               else
                  stride++;
            }
         }
      }
   }
   int stride = 0;
   uint32_t limit = (uint32_t)counts[0], sum = 0;
   for (int base = 0; base <= length; base += 8) {
      uint32_t c[11];
#pragma unroll
      for (int j = 0; j < 11; j++) c[j] = (uint32_t)counts[min(base + j, length - 1)];
      const uint32_t k0 = *(const uint32_t *)(keep + base), k1 = *(const uint32_t *)(keep + base + 4);
#pragma unroll
      for (int j = 0; j < 8; j++) {
         const int i = base + j;
         if (i <= length) {
            bool brk = (i == length) || (((j < 4 ? k0 >> (8 * j) : k1 >> (8 * (j - 4))) & 0xffu) != 0u);
            if (!brk) brk = (c[j] > limit ? c[j] - limit : limit - c[j]) >= 4;
            if (brk) {
               if (stride >= 4 || (stride >= 3 && sum == 0)) {
                  int count = (int)((sum + (uint32_t)(stride / 2)) / (uint32_t)stride);
                  if (count < 1) count = 1;
                  if (sum == 0) count = 0;
                  for (int k = 0; k < stride; k++) counts[i - k - 1] = count;
               }
               stride = 0;
               sum = 0;
               if (i < length - 3)
                  limit = (uint32_t)(((int32_t)c[j] + (int32_t)c[j + 1] + (int32_t)c[j + 2] + (int32_t)c[j + 3] + 2) / 4);
               else if (i < length)
                  limit = c[j];
               else
                  limit = 0;
            }
            stride++;
            if (i != length) sum += c[j];
         }
      }
   }
}
