/*
 * zultra_oracle.c — CPU ORACLE. TEST INFRASTRUCTURE ONLY (see zultra_oracle.h).
 *
 * Restates, in our own words and structure, what the reference computes on its per-max-block hot path.
 * Each function cites the reference file:line it follows (paths relative to /root/reference/).
 * Parity is PINNED against the compiled reference (oracle/_ref) by tests/test_oracle_vs_ref.py.
 *
 * Deliberate differences in *how* (never in *what*):
 *  - the matchfinder does not build a suffix array / lcp-interval tree; it computes the same rows in
 *    closed form (nearest-first scan of earlier occurrences, keeping every strictly longer match);
 *  - the RFC 1951 length/distance tables are generated, not spelled out;
 *  - the three flavours of the code-length RLE tokenizer share one tokenizer with a sink;
 *  - sub-blocks are encoded from bit phase 0 into their own buffer and stitched afterwards, which is
 *    how the device path works too.
 */
#include "zultra_oracle.h"
#include <stdlib.h>
#include <string.h>

#define MIN_MATCH 3
#define MAX_MATCH 258
#define MAX_DIST 32768
#define LEAVE_ALONE 40         /* src/private.h:52 */
#define NLIT 288
#define NDIST 32
#define NCL 19
#define EOB 256

/* ------------------------------------------------------------------------------------------------ */
/* RFC 1951 symbol tables (values equal to src/blockdeflate.c:45-85, generated instead of listed)     */
/* ------------------------------------------------------------------------------------------------ */

static uint16_t g_len_sym[256];     /* index = length-3 */
static uint8_t  g_len_xbits[256];
static uint16_t g_len_base[256];    /* base, in units of (length-3) */
static uint8_t  g_lensym_xbits[29]; /* per length symbol 257.. */
static uint8_t  g_dist_sym[32768];  /* index = distance-1 */
static uint8_t  g_distsym_xbits[32];
static uint16_t g_distsym_base[32];
static int g_tables_ready = 0;

static void zo_init_tables(void) {
   static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59,
                                      67, 83, 99, 115, 131, 163, 195, 227, 258};
   static const uint8_t lx[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
   int s, l, d;
   if (g_tables_ready) return;
   for (s = 0; s < 29; s++) g_lensym_xbits[s] = lx[s];
   for (l = 3; l <= 258; l++) {
      s = 28;
      if (l < 258) {
         s = 0;
         while (s + 1 < 28 && lbase[s + 1] <= l) s++;
      }
      g_len_sym[l - 3] = (uint16_t)(257 + s);
      g_len_xbits[l - 3] = lx[s];
      g_len_base[l - 3] = (uint16_t)(lbase[s] - 3);
   }
   for (s = 0; s < 32; s++) {
      g_distsym_xbits[s] = 0;
      g_distsym_base[s] = 0;
   }
   d = 1;
   for (s = 0; s < 30; s++) {
      int xb = (s < 4) ? 0 : (s / 2 - 1);
      int cnt = 1 << xb, k;
      g_distsym_xbits[s] = (uint8_t)xb;
      g_distsym_base[s] = (uint16_t)d;
      for (k = 0; k < cnt; k++) g_dist_sym[d - 1 + k] = (uint8_t)s;
      d += cnt;
   }
   g_tables_ready = 1;
}

/* ------------------------------------------------------------------------------------------------ */
/* Stage 1: matchfinder — closed form of src/matchfinder.c:49-286                                     */
/* ------------------------------------------------------------------------------------------------ */

static __thread uint64_t g_last_candidates;
uint64_t zo_last_match_candidates(void) { return g_last_candidates; }

/*
 * What src/matchfinder.c:171-234 (lazy ascent of the lcp-interval tree built at :98-155 over LCPs
 * clamped to [3,258], :85-88) yields for position i: walking up from the deepest interval, each interval
 * contributes the most recently visited suffix inside it, unless a deeper interval already produced that
 * same or a later position. Equivalently: scan earlier positions p from nearest to farthest and keep p
 * whenever its match length L(p) = min(LCP(i,p), 258, end-i) is >= 3 and strictly greater than every
 * length seen so far. The reference emits the rows longest-first, skips offsets > 32768 without using a
 * slot (:217-225) and stops storing after 8 (:217); src/matchfinder.c:262-286 pads with (0,0).
 */
void zo_find_matches(const uint8_t *win, int prev, int n, zo_match_t *match) {
   const int W = prev + n;
   const int HB = 16;
   int *head = (int *)malloc(sizeof(int) << HB);
   int *chain = (int *)malloc(sizeof(int) * (size_t)(W > 0 ? W : 1));
   uint64_t visits = 0;
   int i;

   for (i = 0; i < (1 << HB); i++) head[i] = -1;

   for (i = 0; i < W; i++) {
      const int maxlen = (W - i < MAX_MATCH) ? (W - i) : MAX_MATCH;
      uint32_t h = 0;

      if (maxlen >= MIN_MATCH)
         h = (((uint32_t)win[i] << 16 | (uint32_t)win[i + 1] << 8 | win[i + 2]) * 2654435761u) >> (32 - HB);

      if (i >= prev) {
         zo_match_t rec[ZO_NMATCHES];   /* ring of the last 8 records */
         int nrec = 0, cur = MIN_MATCH - 1, m;
         zo_match_t *row = match + ((size_t)(i - prev) << 3);

         if (maxlen >= MIN_MATCH) {
            int p = head[h];
            while (p >= 0 && (i - p) <= MAX_DIST && cur < maxlen) {
               visits++;
               if (win[p + cur] == win[i + cur] && win[p] == win[i] && win[p + 1] == win[i + 1] && win[p + 2] == win[i + 2]) {
                  int l = MIN_MATCH;
                  while (l < maxlen && win[p + l] == win[i + l]) l++;
                  if (l > cur) {
                     rec[nrec & 7].length = (uint16_t)l;
                     rec[nrec & 7].offset = (uint16_t)(i - p);   /* <= 32768 fits */
                     nrec++;
                     cur = l;
                  }
               }
               p = chain[p];
            }
         }
         for (m = 0; m < ZO_NMATCHES; m++) {
            if (m < nrec && m < ZO_NMATCHES) {
               row[m] = rec[(nrec - 1 - m) & 7];
            }
            else {
               row[m].length = 0;
               row[m].offset = 0;
            }
         }
      }

      if (maxlen >= MIN_MATCH) {
         chain[i] = head[h];
         head[h] = i;
      }
   }
   g_last_candidates = visits;
   free(chain);
   free(head);
}

/* ------------------------------------------------------------------------------------------------ */
/* Huffman primitives — src/huffman/huffencoder.c                                                     */
/* ------------------------------------------------------------------------------------------------ */

typedef struct {
   int nsym;
   int maxbits;
   int freq[NLIT];
   int len[NLIT];
   uint32_t code[NLIT];
} zo_huff_t;

/* huffencoder.c:73-98 */
static void huff_init(zo_huff_t *h, int nsym, int maxbits) {
   memset(h, 0, sizeof(*h));
   h->nsym = nsym;
   h->maxbits = maxbits;
}

/* Ordering used everywhere the reference sorts symbols: key ascending, then symbol index ascending
 * (huffencoder.c:34-61, comparison at :48). It is a total order, so any sort gives the same result. */
static const int *g_sort_key;
static int sort_cmp(const void *a, const void *b) {
   int ia = *(const int *)a, ib = *(const int *)b;
   if (g_sort_key[ia] != g_sort_key[ib]) return (g_sort_key[ia] < g_sort_key[ib]) ? -1 : 1;
   return (ia < ib) ? -1 : (ia > ib);
}
static void sort_syms(const int *key, int *idx, int n) {
   g_sort_key = key;
   qsort(idx, (size_t)n, sizeof(int), sort_cmp);
}

/* huffencoder.c:157-270 — Moffat-Katajainen in-place code lengths, no length limit.
 * Tie rule (:203,:215): an internal node is taken only if it is strictly lighter than the next leaf.
 * With at most one used symbol, symbol 0 gets length 1 whichever symbol was used (:263-267). */
static void huff_lengths(zo_huff_t *h) {
   int order[NLIT], A[NLIT];
   int n = 0, i;

   for (i = 0; i < h->nsym; i++)
      if (h->freq[i]) order[n++] = i;

   memset(h->len, 0, sizeof(h->len));
   if (n <= 1) {
      h->len[0] = 1;
      return;
   }

   sort_syms(h->freq, order, n);
   for (i = 0; i < n; i++) A[i] = h->freq[order[i]];

   {
      int leaf = 0, node = 0, t;
      /* phase 1: pair the two lightest items; A[node] := parent index of a consumed internal node */
      for (t = 0; t < n - 1; t++) {
         int w, pick;
         for (pick = 0, w = 0; pick < 2; pick++) {
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
      /* phase 2: parent pointers -> internal node depths */
      A[n - 2] = 0;
      for (t = n - 3; t >= 0; t--) A[t] = A[A[t]] + 1;
      /* phase 3: internal depths -> leaf depths */
      {
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
   }
   for (i = 0; i < n; i++) h->len[order[i]] = A[i];
}

static uint32_t bitrev16(uint32_t v, int nbits) {
   v = ((v & 0x5555u) << 1) | ((v & 0xaaaau) >> 1);
   v = ((v & 0x3333u) << 2) | ((v & 0xccccu) >> 2);
   v = ((v & 0x0f0fu) << 4) | ((v & 0xf0f0u) >> 4);
   v = ((v & 0x00ffu) << 8) | ((v & 0xff00u) >> 8);
   return v >> (16 - nbits);
}

/* huffencoder.c:348-372 (also :118-145): canonical codes over symbols already ordered by (len, index),
 * stored bit-reversed so they can be emitted LSB-first. */
static void huff_assign_codes(zo_huff_t *h, const int *order, int n) {
   uint32_t code = 0;
   int i;
   for (i = 0; i < n; i++) {
      int s = order[i];
      h->code[s] = bitrev16(code, h->len[s]);
      if (i + 1 < n) code = (code + 1) << (h->len[order[i + 1]] - h->len[s]);
   }
}

/* huffencoder.c:279-375. Returns -1 if the reference would have run off its sorted array (:334). */
static int huff_build_dynamic(zo_huff_t *h) {
   int order[NLIT + 1];
   int n = 0, i;

   huff_lengths(h);
   for (i = 0; i < h->nsym; i++)
      if (h->len[i]) order[n++] = i;

   if (n > 0 && h->maxbits > 0) {
      sort_syms(h->len, order, n);
      if (h->len[order[n - 1]] > h->maxbits) {
         /* heuristic limiter (:310-344): clamp, then repair the Kraft sum from the tail, then give
          * slack back from the head */
         const int full = 1 << h->maxbits;
         int kraft = 0;
         for (i = n - 1; i >= 0; i--) {
            int s = order[i];
            if (h->len[s] > h->maxbits) h->len[s] = h->maxbits;
            kraft += full >> h->len[s];
         }
         for (i = n - 1; kraft > full && i >= 0; i--) {
            int s = order[i];
            while (h->len[s] < h->maxbits && kraft > full) {
               h->len[s]++;
               kraft -= full >> h->len[s];
            }
         }
         for (i = 0; kraft < full; i++) {
            int s;
            if (i >= n) return -1;   /* reference would read nMinQueue[nNumSorted]: believed unreachable */
            s = order[i];
            while (kraft + (full >> h->len[s]) <= full) {
               kraft += full >> h->len[s];
               h->len[s]--;
            }
         }
         sort_syms(h->len, order, n);
      }
   }
   if (n > 0) huff_assign_codes(h, order, n);
   return 0;
}

/* huffencoder.c:107-148: every symbol has a length; canonical over all of them. */
static void huff_build_static(zo_huff_t *h) {
   int order[NLIT], i;
   for (i = 0; i < h->nsym; i++) order[i] = i;
   sort_syms(h->len, order, h->nsym);
   huff_assign_codes(h, order, h->nsym);
}

/* huffencoder.c:532-538 */
static int huff_defined_count(const zo_huff_t *h, int min_syms) {
   int i = h->nsym;
   while (i > min_syms && !h->len[i - 1]) i--;
   return i;
}

static const uint8_t g_cl_order[NCL] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

/* huffencoder.c:400-406 */
static int huff_raw_table_size(const zo_huff_t *h) {
   int i = h->nsym;
   while (i > 4 && !h->len[g_cl_order[i - 1]]) i--;
   return i;
}

/* ------------------------------------------------------------------------------------------------ */
/* Bit sink (phase-0 buffer, counts past the capacity) — src/huffman/bitwriter.c:63-98               */
/* ------------------------------------------------------------------------------------------------ */

typedef struct {
   uint8_t *buf;
   size_t cap;
   uint64_t nbits;
} zo_bits_t;

static void bits_put(zo_bits_t *b, uint32_t value, int nbits) {
   int k;
   for (k = 0; k < nbits; k++) {
      uint64_t pos = b->nbits + (uint64_t)k;
      if ((pos >> 3) < b->cap) {
         uint8_t bit = (uint8_t)((value >> k) & 1u);
         if ((pos & 7) == 0) b->buf[pos >> 3] = 0;
         b->buf[pos >> 3] |= (uint8_t)(bit << (pos & 7));
      }
   }
   b->nbits += (uint64_t)nbits;
}

/* ------------------------------------------------------------------------------------------------ */
/* Code-length RLE tokenizer — huffencoder.c:446-522 / :549-628 / :640-735 are one tokenizer          */
/* ------------------------------------------------------------------------------------------------ */

typedef void (*cl_sink_fn)(void *ctx, int sym, int extra_value, int extra_bits);

/* mask bits: 1 = symbol 16 allowed, 2 = symbol 17, 4 = symbol 18, 8 = no 4+3 split of a 7-repeat,
 * 16 = no 4+4 split of an 8-repeat. */
static void cl_tokenize(const int *lens, int n, unsigned mask, cl_sink_fn sink, void *ctx) {
   int i = 0;
   while (i < n) {
      int run = 1;
      while (i + run < n && lens[i + run] == lens[i]) run++;

      if (lens[i] == 0) {
         if (run >= 3) {
            while (run >= 11 && (mask & 4)) {
               int take = run > 138 ? 138 : run;
               sink(ctx, 18, take - 11, 7);
               run -= take;
               i += take;
            }
            while (run >= 3 && (mask & 2)) {
               int take = run > 10 ? 10 : run;
               sink(ctx, 17, take - 3, 3);
               run -= take;
               i += take;
            }
            if (run) {
               sink(ctx, lens[i], 0, 0);
               i++;
            }
         }
         else {
            sink(ctx, lens[i], 0, 0);
            i++;
         }
      }
      else {
         int v = lens[i] > 15 ? 15 : lens[i];
         sink(ctx, v, 0, 0);
         i++;
         run--;
         if (run == 7 && (mask & 1) && !(mask & 8)) {
            sink(ctx, 16, 4 - 3, 2);
            sink(ctx, 16, 3 - 3, 2);
            run = 0;
            i += 7;
         }
         else if (run == 8 && (mask & 1) && !(mask & 16)) {
            sink(ctx, 16, 4 - 3, 2);
            sink(ctx, 16, 4 - 3, 2);
            run = 0;
            i += 8;
         }
         while (run >= 3 && (mask & 1)) {
            int take = run > 6 ? 6 : run;
            sink(ctx, 16, take - 3, 2);
            run -= take;
            i += take;
         }
      }
   }
}

static void cl_sink_count(void *ctx, int sym, int xv, int xb) {
   (void)xv;
   (void)xb;
   ((zo_huff_t *)ctx)->freq[sym]++;
}

typedef struct {
   const zo_huff_t *h;
   int bits;
} cl_size_ctx;
static void cl_sink_size(void *ctx, int sym, int xv, int xb) {
   cl_size_ctx *c = (cl_size_ctx *)ctx;
   (void)xv;
   c->bits += c->h->len[sym] + xb;
}

typedef struct {
   const zo_huff_t *h;
   zo_bits_t *out;
} cl_write_ctx;
static void cl_sink_write(void *ctx, int sym, int xv, int xb) {
   cl_write_ctx *c = (cl_write_ctx *)ctx;
   bits_put(c->out, c->h->code[sym], c->h->len[sym]);
   if (xb) bits_put(c->out, (uint32_t)xv, xb);
}

static int cl_size(const zo_huff_t *tables, const int *lens, int n, unsigned mask) {
   cl_size_ctx c;
   c.h = tables;
   c.bits = 0;
   cl_tokenize(lens, n, mask, cl_sink_size, &c);
   return c.bits;
}

/* ------------------------------------------------------------------------------------------------ */
/* zopfli-style histogram smoothing — src/huffman/huffutils.c:34-114                                  */
/* ------------------------------------------------------------------------------------------------ */

static void smooth_for_rle(int length, int *counts) {
   int keep[NLIT];
   int i, k, stride;
   size_t symbol, sum, limit;

   while (length > 0 && counts[length - 1] == 0) length--;   /* trailing zeros untouched */
   if (length == 0) return;

   /* runs that RLE already handles well: >=5 zeros or >=7 equal non-zeros */
   for (i = 0; i < length; i++) keep[i] = 0;
   symbol = (size_t)counts[0];
   stride = 0;
   for (i = 0; i <= length; i++) {
      if (i == length || (size_t)counts[i] != symbol) {
         if ((symbol == 0 && stride >= 5) || (symbol != 0 && stride >= 7))
            for (k = 0; k < stride; k++) keep[i - k - 1] = 1;
         stride = 1;
         if (i != length) symbol = (size_t)counts[i];
      }
      else
         stride++;
   }

   /* collapse strides of near-equal counts to their rounded mean */
   stride = 0;
   limit = (size_t)counts[0];
   sum = 0;
   for (i = 0; i <= length; i++) {
      int brk = (i == length) || keep[i];
      if (!brk) {
         size_t c = (size_t)counts[i];
         size_t diff = c > limit ? c - limit : limit - c;
         brk = diff >= 4;
      }
      if (brk) {
         if (stride >= 4 || (stride >= 3 && sum == 0)) {
            int count = (int)((sum + (size_t)(stride / 2)) / (size_t)stride);
            if (count < 1) count = 1;
            if (sum == 0) count = 0;
            for (k = 0; k < stride; k++) counts[i - k - 1] = count;
         }
         stride = 0;
         sum = 0;
         if (i < length - 3)
            limit = (size_t)((counts[i] + counts[i + 1] + counts[i + 2] + counts[i + 3] + 2) / 4);
         else if (i < length)
            limit = (size_t)counts[i];
         else
            limit = 0;
      }
      stride++;
      if (i != length) sum += (size_t)counts[i];
   }
}

/* ------------------------------------------------------------------------------------------------ */
/* Per-block working context                                                                          */
/* ------------------------------------------------------------------------------------------------ */

typedef struct {
   const uint8_t *win;
   const zo_match_t *match;   /* rows of block positions */
   int prev;
   zo_huff_t lit, dist;
   int *cost;                 /* [W+1] absolute window index */
   zo_match_t *best;          /* [W]   absolute window index */
} zo_ctx_t;

#define MROW(c, i) ((c)->match + ((size_t)((i) - (c)->prev) << 3))

static inline int dist_symbol(int d) { return g_dist_sym[d - 1]; }
static inline int dist_cost(const zo_ctx_t *c, int d) {
   int s = g_dist_sym[d - 1];
   return c->dist.len[s] + g_distsym_xbits[s];
}
static inline int len_cost(const zo_ctx_t *c, unsigned enc /* length-3, wraps like the reference */) {
   if (enc > 255) enc = 255;   /* blockdeflate.c:216-219 */
   return c->lit.len[g_len_sym[enc]] + g_len_xbits[enc];
}

/* blockdeflate.c:333-361: histogram of the greedy parse (longest match at every step, else literal). */
static void greedy_histogram(zo_ctx_t *c, int start, int end) {
   int i = start;
   while (i < end) {
      const zo_match_t *r = MROW(c, i);
      if (r[0].length >= MIN_MATCH) {
         c->lit.freq[g_len_sym[r[0].length - MIN_MATCH]]++;
         c->dist.freq[dist_symbol(r[0].offset)]++;
         i += r[0].length;
      }
      else {
         c->lit.freq[c->win[i]]++;
         i++;
      }
   }
   c->lit.freq[EOB]++;
}

/* blockdeflate.c:519-527 */
static void prepare_cost_evaluation(zo_ctx_t *c, int start, int size) {
   huff_init(&c->lit, NLIT, 15);
   huff_init(&c->dist, NDIST, 15);
   greedy_histogram(c, start, start + size);
}

static int static_lit_len(int s) { return s < 144 ? 8 : (s < 256 ? 9 : (s < 280 ? 7 : 8)); }

/* blockdeflate.c:538-566 */
static int static_cost(const zo_huff_t *lit, const zo_huff_t *dist) {
   int bits = 0, s;
   for (s = 0; s < 257; s++) bits += lit->freq[s] * static_lit_len(s);
   for (; s < 257 + 29; s++) bits += lit->freq[s] * (static_lit_len(s) + g_lensym_xbits[s - 257]);
   for (s = 0; s < NDIST; s++) bits += dist->freq[s] * (5 + g_distsym_xbits[s]);
   return bits + 3;
}

/* blockdeflate.c:577-618: note the code-length alphabet is histogrammed with mask 7 (:602), sized
 * with mask 31 (:613), and its own lengths come from the *unlimited* estimate (:603). */
static int dynamic_cost(const zo_huff_t *lit, const zo_huff_t *dist) {
   zo_huff_t tables;
   int lens[NLIT + NDIST];
   int bits = 0, s, nlit, ndist;

   for (s = 0; s < 257; s++) bits += lit->freq[s] * lit->len[s];
   for (; s < 257 + 29; s++) bits += lit->freq[s] * (lit->len[s] + g_lensym_xbits[s - 257]);
   for (s = 0; s < NDIST; s++) bits += dist->freq[s] * (dist->len[s] + g_distsym_xbits[s]);

   nlit = huff_defined_count(lit, 257);
   ndist = huff_defined_count(dist, 1);
   memcpy(lens, lit->len, (size_t)nlit * sizeof(int));
   memcpy(lens + nlit, dist->len, (size_t)ndist * sizeof(int));

   huff_init(&tables, NCL, 7);
   cl_tokenize(lens, nlit + ndist, 7, cl_sink_count, &tables);
   huff_lengths(&tables);

   bits += 5 + 5 + 4;
   bits += 3 * huff_raw_table_size(&tables);
   bits += cl_size(&tables, lens, nlit + ndist, 31);
   return bits + 3;
}

/* ------------------------------------------------------------------------------------------------ */
/* Stage 2: block splitter — src/blockdeflate.c:634-813                                               */
/* ------------------------------------------------------------------------------------------------ */

static int split_range(zo_ctx_t *c, int start, int size, int depth, int max_splits, int *count, int *out) {
   unsigned seen[18], fresh[18];
   unsigned nseen = 0, nfresh = 0;
   zo_huff_t total_lit, total_dist, left_lit, left_dist, right_lit, right_dist;
   int total_cost, i, j;
   int checkpoint = -1;          /* nLastGoodSplitIdx */
   int left_end = start;         /* nLastLeftEndOffset */
   const int end = start + size;
   int best_off = end, best_gain = 0;

   if (*count >= max_splits) return 0;
   if (depth >= 6 || size < 8192) return 0;

   memset(seen, 0, sizeof(seen));
   memset(fresh, 0, sizeof(fresh));

   prepare_cost_evaluation(c, start, size);
   huff_lengths(&c->lit);
   huff_lengths(&c->dist);
   total_cost = dynamic_cost(&c->lit, &c->dist);
   total_lit = c->lit;
   total_dist = c->dist;
   huff_init(&left_lit, NLIT, 15);
   huff_init(&left_dist, NDIST, 15);
   huff_init(&right_lit, NLIT, 15);
   huff_init(&right_dist, NDIST, 15);

   i = start;
   while (i < end) {
      const zo_match_t *r = MROW(c, i);
      if (r[0].length >= MIN_MATCH) {
         fresh[r[0].length >= 9 ? 17 : 16]++;
         i += r[0].length;
      }
      else {
         unsigned b = c->win[i];
         fresh[((b >> 4) & 0xc) | (b & 3)]++;
         i++;
      }
      nfresh++;

      if (nfresh >= 256 && (i - start) >= 512) {
         if (nseen) {
            unsigned drift = 0;   /* uint32 wrap-around is part of the behaviour (:710-718) */
            for (j = 0; j < 18; j++) {
               unsigned expected = seen[j] * nfresh;
               unsigned actual = fresh[j] * nseen;
               drift += expected > actual ? expected - actual : actual - expected;
            }
            if ((drift / nfresh) >= (nseen * 45 / 100) && checkpoint >= 0) {
               int lcost, rcost, gain;
               /* left histogram grows by the greedy tokens of [left_end, checkpoint) (:732-737) */
               prepare_cost_evaluation(c, left_end, checkpoint - left_end);
               for (j = 0; j < NLIT; j++) left_lit.freq[j] += c->lit.freq[j];
               for (j = 0; j < NDIST; j++) left_dist.freq[j] += c->dist.freq[j];
               left_lit.freq[EOB] = 1;
               for (j = 0; j < NLIT; j++) right_lit.freq[j] = total_lit.freq[j] - left_lit.freq[j];
               for (j = 0; j < NDIST; j++) right_dist.freq[j] = total_dist.freq[j] - left_dist.freq[j];
               right_lit.freq[EOB] = 1;

               huff_lengths(&left_lit);
               huff_lengths(&left_dist);
               lcost = dynamic_cost(&left_lit, &left_dist);
               huff_lengths(&right_lit);
               huff_lengths(&right_dist);
               rcost = dynamic_cost(&right_lit, &right_dist);
               gain = total_cost - (lcost + rcost);
               if (gain >= 0 && (best_off == end || best_gain < gain)) {
                  best_off = checkpoint;
                  best_gain = gain;
               }
               left_end = checkpoint;
            }
         }
         for (j = 0; j < 18; j++) {
            nseen += fresh[j];
            seen[j] += fresh[j];
            fresh[j] = 0;
         }
         nfresh = 0;
         checkpoint = i;
      }
   }

   if (best_off != end) {
      if (split_range(c, start, best_off - start, depth + 1, max_splits, count, out) < 0) return -1;
      if (*count < max_splits) out[(*count)++] = best_off;
      if (split_range(c, best_off, end - best_off, depth + 1, max_splits, count, out) < 0) return -1;
   }
   return 0;
}

int zo_block_split(const uint8_t *win, const zo_match_t *match, int prev, int n, int *split_off) {
   zo_ctx_t c;
   int count = 0;
   zo_init_tables();
   memset(&c, 0, sizeof(c));
   c.win = win;
   c.match = match;
   c.prev = prev;
   if (split_range(&c, prev, n, 0, ZO_MAX_SPLITS - 1, &count, split_off) < 0) return -1;
   if (count >= ZO_MAX_SPLITS) return -1;
   split_off[count++] = prev + n;
   return count;
}

/* ------------------------------------------------------------------------------------------------ */
/* Stage 3: per-sub-block coder — src/blockdeflate.c:254-507, 827-997                                 */
/* ------------------------------------------------------------------------------------------------ */

int zo_subblock_costs(const uint8_t *win, const zo_match_t *match, int prev, int start, int size,
                      int *pstatic, int *pdynamic) {
   zo_ctx_t c;
   int s, d;
   zo_init_tables();
   memset(&c, 0, sizeof(c));
   c.win = win;
   c.match = match;
   c.prev = prev;
   prepare_cost_evaluation(&c, start, size);
   s = static_cost(&c.lit, &c.dist);
   huff_lengths(&c.lit);
   huff_lengths(&c.dist);
   d = dynamic_cost(&c.lit, &c.dist);
   if (pstatic) *pstatic = s;
   if (pdynamic) *pdynamic = d;
   return (s <= d) ? 0 : 1;   /* libzultra.c:323 */
}

/* blockdeflate.c:254-323: backward optimal parse. Candidates are tried in the order literal, then
 * match slot 0..7, each from its (end-clamped) length down to 3; a candidate replaces the incumbent
 * only if strictly cheaper. Stored lengths >= 40 are tried at full length only — and if the end clamp
 * brings that length below 3 the length cost wraps to the cost of symbol 285 (:289 with :216-219). */
static void optimal_parse(zo_ctx_t *c, int start, int end) {
   int lencache[LEAVE_ALONE];
   int i, k;
   if (end <= start) return;
   for (k = 0; k < LEAVE_ALONE; k++) lencache[k] = len_cost(c, (unsigned)k);

   c->cost[end] = 0;
   for (i = end - 1; i >= start; i--) {
      const zo_match_t *r = MROW(c, i);
      int best_cost = c->lit.len[c->win[i]] + c->cost[i + 1];
      int best_len = 0, best_off = 0, m;

      for (m = 0; m < ZO_NMATCHES && r[m].length >= MIN_MATCH; m++) {
         int oc = dist_cost(c, r[m].offset);
         int mlen = r[m].length;
         if (i + mlen > end) mlen = end - i;

         if (r[m].length >= LEAVE_ALONE) {
            int cc = len_cost(c, (unsigned)(mlen - MIN_MATCH)) + oc + c->cost[i + mlen];
            if (best_cost > cc) {
               best_cost = cc;
               best_len = mlen;
               best_off = r[m].offset;
            }
         }
         else {
            for (k = mlen; k >= MIN_MATCH; k--) {
               int cc = lencache[k - MIN_MATCH] + oc + c->cost[i + k];
               if (best_cost > cc) {
                  best_cost = cc;
                  best_len = k;
                  best_off = r[m].offset;
               }
            }
         }
      }
      c->cost[i] = best_cost;
      c->best[i].length = (uint16_t)best_len;
      c->best[i].offset = (uint16_t)best_off;
   }
}

/* blockdeflate.c:371-400 */
static void parse_histogram(zo_ctx_t *c, int start, int end) {
   int i = start;
   while (i < end) {
      const zo_match_t *b = &c->best[i];
      if (b->length >= MIN_MATCH) {
         c->lit.freq[g_len_sym[b->length - MIN_MATCH]]++;
         c->dist.freq[dist_symbol(b->offset)]++;
         i += b->length;
      }
      else {
         c->lit.freq[c->win[i]]++;
         i++;
      }
   }
   c->lit.freq[EOB]++;
}

/* blockdeflate.c:410-458: turn a chosen match back into literals when that is strictly cheaper under
 * the final code lengths and every one of its bytes has a code. */
static void literalize_cheap_matches(zo_ctx_t *c, int start, int end) {
   int i = start;
   while (i < end) {
      zo_match_t *b = &c->best[i];
      if (b->length >= MIN_MATCH) {
         unsigned mlen = b->length, off = b->offset, j;
         unsigned mcost, lcost = 0;
         int at = i, usable = 1;
         i += (int)mlen;
         if (off < 1 || off > MAX_DIST) continue;
         mcost = (unsigned)len_cost(c, mlen - MIN_MATCH) + (unsigned)dist_cost(c, (int)off);
         for (j = 0; j < mlen && lcost < mcost; j++) {
            unsigned l = (unsigned)c->lit.len[c->win[at + (int)j]];
            if (l == 0) {
               usable = 0;
               break;
            }
            lcost += l;
         }
         if (usable && lcost < mcost)
            for (j = 0; j < mlen; j++) c->best[at + (int)j].length = 0;
      }
      else
         i++;
   }
}

/* blockdeflate.c:471-507 */
static int emit_tokens(zo_ctx_t *c, zo_bits_t *out, int start, int end) {
   int i = start;
   while (i < end) {
      const zo_match_t *b = &c->best[i];
      if (b->length >= MIN_MATCH) {
         unsigned enc = (unsigned)(b->length - MIN_MATCH);
         int d = b->offset, ds;
         if (d < 1 || d > MAX_DIST) return -1;
         if (enc > 255) enc = 255;
         bits_put(out, c->lit.code[g_len_sym[enc]], c->lit.len[g_len_sym[enc]]);
         bits_put(out, (uint32_t)(b->length - MIN_MATCH) - g_len_base[enc], g_len_xbits[enc]);
         ds = dist_symbol(d);
         bits_put(out, c->dist.code[ds], c->dist.len[ds]);
         bits_put(out, (uint32_t)(d - g_distsym_base[ds]), g_distsym_xbits[ds]);
         i += b->length;
      }
      else {
         bits_put(out, c->lit.code[c->win[i]], c->lit.len[c->win[i]]);
         i++;
      }
   }
   bits_put(out, c->lit.code[EOB], c->lit.len[EOB]);
   return 0;
}

static int deflate_subblock(zo_ctx_t *c, zo_bits_t *out, int start, int size, int is_dynamic) {
   const int end = start + size;
   int s, pass;

   huff_init(&c->lit, NLIT, 15);
   huff_init(&c->dist, NDIST, 15);

   if (!is_dynamic) {
      /* blockdeflate.c:836-858 */
      for (s = 0; s < NLIT; s++) c->lit.len[s] = static_lit_len(s);
      for (s = 0; s < NDIST; s++) c->dist.len[s] = 5;
      huff_build_static(&c->lit);
      huff_build_static(&c->dist);
      optimal_parse(c, start, end);
      return emit_tokens(c, out, start, end);
   }

   /* blockdeflate.c:859-920 */
   greedy_histogram(c, start, end);
   if (huff_build_dynamic(&c->lit) < 0 || huff_build_dynamic(&c->dist) < 0) return -1;

   for (pass = 0; pass <= 3; pass++) {
      for (s = 0; s < NLIT; s++)
         if (!c->lit.len[s]) c->lit.len[s] = 9;
      for (s = 0; s < NDIST; s++)
         if (!c->dist.len[s]) c->dist.len[s] = 6;

      optimal_parse(c, start, end);

      memset(c->lit.freq, 0, sizeof(c->lit.freq));
      memset(c->dist.freq, 0, sizeof(c->dist.freq));
      parse_histogram(c, start, end);

      if (pass == 3) {
         /* at least two distance codes among 0..29, for pre-1.2.1.1 zlib inflate (:893-913) */
         int used = 0;
         for (s = 0; used < 2 && s < NDIST - 2; s++)
            if (c->dist.freq[s]) used++;
         if (used == 0)
            c->dist.freq[0] = c->dist.freq[1] = 1;
         else if (used == 1) {
            if (c->dist.freq[0])
               c->dist.freq[1] = 1;
            else
               c->dist.freq[0] = 1;
         }
      }
      if (huff_build_dynamic(&c->lit) < 0 || huff_build_dynamic(&c->dist) < 0) return -1;
   }

   literalize_cheap_matches(c, start, end);   /* histograms intentionally left stale (:923) */

   {
      /* blockdeflate.c:925-945: RLE-friendlier tables, adopted only if strictly cheaper overall */
      zo_huff_t alt_lit = c->lit, alt_dist = c->dist;
      int cur_cost = dynamic_cost(&alt_lit, &alt_dist), alt_cost;
      smooth_for_rle(NLIT, alt_lit.freq);
      smooth_for_rle(NDIST, alt_dist.freq);
      if (huff_build_dynamic(&alt_lit) < 0 || huff_build_dynamic(&alt_dist) < 0) return -1;
      alt_cost = dynamic_cost(&alt_lit, &alt_dist);
      if (alt_cost < cur_cost) {
         c->lit = alt_lit;
         c->dist = alt_dist;
      }
   }

   {
      /* blockdeflate.c:947-992: header */
      zo_huff_t tables;
      int lens[NLIT + NDIST];
      int nlit = huff_defined_count(&c->lit, 257);
      int ndist = huff_defined_count(&c->dist, 1);
      int best_mask = -1, best_cost = 0, mask, ncl, k;
      cl_write_ctx w;

      memcpy(lens, c->lit.len, (size_t)nlit * sizeof(int));
      memcpy(lens + nlit, c->dist.len, (size_t)ndist * sizeof(int));

      huff_init(&tables, NCL, 7);
      for (mask = 0; mask <= 31; mask += (mask >= 7) ? 2 : 1) {
         int cost;
         cl_tokenize(lens, nlit + ndist, (unsigned)mask, cl_sink_count, &tables);
         if (huff_build_dynamic(&tables) < 0) return -1;
         cost = cl_size(&tables, lens, nlit + ndist, (unsigned)mask);
         if (best_mask == -1 || best_cost >= cost) {   /* last minimal mask wins (:966) */
            best_mask = mask;
            best_cost = cost;
         }
         memset(tables.freq, 0, sizeof(tables.freq));
      }
      cl_tokenize(lens, nlit + ndist, (unsigned)best_mask, cl_sink_count, &tables);
      if (huff_build_dynamic(&tables) < 0) return -1;

      ncl = huff_raw_table_size(&tables);
      if (nlit > 286 || ndist > 30 || ncl > NCL) return -1;
      bits_put(out, (uint32_t)(nlit - 257), 5);
      bits_put(out, (uint32_t)(ndist - 1), 5);
      bits_put(out, (uint32_t)(ncl - 4), 4);
      for (k = 0; k < ncl; k++) bits_put(out, (uint32_t)tables.len[g_cl_order[k]], 3);
      w.h = &tables;
      w.out = out;
      cl_tokenize(lens, nlit + ndist, (unsigned)best_mask, cl_sink_write, &w);
   }

   return emit_tokens(c, out, start, end);
}

int zo_subblock_deflate(const uint8_t *win, const zo_match_t *match, int prev, int start, int size,
                        int is_dynamic, uint8_t *outbuf, size_t cap, uint64_t *nbits,
                        zo_match_t *best, int *lit_len, int *dist_len) {
   zo_ctx_t c;
   zo_bits_t out;
   int r;
   const int W = start + size;

   zo_init_tables();
   memset(&c, 0, sizeof(c));
   c.win = win;
   c.match = match;
   c.prev = prev;
   c.cost = (int *)malloc(sizeof(int) * (size_t)(W + 1));
   c.best = (zo_match_t *)calloc((size_t)(W + 1), sizeof(zo_match_t));
   out.buf = outbuf;
   out.cap = cap;
   out.nbits = 0;
   r = deflate_subblock(&c, &out, start, size, is_dynamic);
   if (nbits) *nbits = out.nbits;
   if (best) memcpy(best, c.best + start, sizeof(zo_match_t) * (size_t)size);
   if (lit_len) memcpy(lit_len, c.lit.len, sizeof(int) * NLIT);
   if (dist_len) memcpy(dist_len, c.dist.len, sizeof(int) * NDIST);
   free(c.cost);
   free(c.best);
   return r;
}

/* ------------------------------------------------------------------------------------------------ */
/* Checksums — src/frame.c:74-138 (adler32), :324-354 (crc32, reflected poly 0xEDB88320)              */
/* ------------------------------------------------------------------------------------------------ */

uint32_t zo_adler32(uint32_t adler, const uint8_t *buf, size_t len) {
   uint32_t a = adler & 0xffff, b = (adler >> 16) & 0xffff;
   while (len) {
      size_t chunk = len > 5552 ? 5552 : len, k;
      for (k = 0; k < chunk; k++) {
         a += buf[k];
         b += a;
      }
      a %= 65521u;
      b %= 65521u;
      buf += chunk;
      len -= chunk;
   }
   return a | (b << 16);
}

uint32_t zo_crc32(uint32_t crc, const uint8_t *buf, size_t len) {
   static uint32_t table[256];
   static int ready = 0;
   size_t k;
   if (!ready) {
      uint32_t i, j;
      for (i = 0; i < 256; i++) {
         uint32_t c = i;
         for (j = 0; j < 8; j++) c = (c >> 1) ^ ((c & 1) ? 0xEDB88320u : 0);
         table[i] = c;
      }
      ready = 1;
   }
   crc = ~crc;
   for (k = 0; k < len; k++) crc = (crc >> 8) ^ table[(crc ^ buf[k]) & 0xff];
   return ~crc;
}

/* ------------------------------------------------------------------------------------------------ */
/* Stage 4: stream assembly — src/libzultra.c:200-514, 576-619; framing src/frame.c:387-547           */
/* ------------------------------------------------------------------------------------------------ */

typedef struct {
   uint8_t *out;
   size_t cap;
   size_t pos;        /* bytes committed to out */
   uint32_t acc;      /* pending bits, LSB first */
   int nacc;          /* 0..7 between calls */
   int overflow;
} zo_stream_t;

static void stream_byte(zo_stream_t *s, uint8_t b) {
   if (s->pos < s->cap)
      s->out[s->pos] = b;
   else
      s->overflow = 1;
   s->pos++;
}

static void stream_bits(zo_stream_t *s, uint32_t v, int n) {
   s->acc |= v << s->nacc;
   s->nacc += n;
   while (s->nacc >= 8) {
      stream_byte(s, (uint8_t)s->acc);
      s->acc >>= 8;
      s->nacc -= 8;
   }
}

static void stream_pad(zo_stream_t *s) {
   if (s->nacc > 0) {
      stream_byte(s, (uint8_t)(s->acc & ((1u << s->nacc) - 1)));
      s->acc = 0;
      s->nacc = 0;
   }
}

static void stream_append(zo_stream_t *s, const uint8_t *bits, uint64_t nbits) {
   uint64_t full = nbits >> 3, k;
   for (k = 0; k < full; k++) stream_bits(s, bits[k], 8);
   if (nbits & 7) stream_bits(s, bits[full] & ((1u << (nbits & 7)) - 1), (int)(nbits & 7));
}

static unsigned clamp_block(unsigned max_block) {
   if (!max_block) max_block = 1048576;      /* libzultra.c:87-92 */
   if (max_block < 32768) max_block = 32768;
   if (max_block > 2097152) max_block = 2097152;
   return max_block;
}

size_t zo_memory_bound(size_t n, unsigned flags, unsigned max_block) {
   size_t hdr = (flags & 2) ? 10 : ((flags & 1) ? 2 : 0);
   size_t ftr = (flags & 2) ? 8 : ((flags & 1) ? 4 : 0);
   max_block = clamp_block(max_block);
   return hdr + ((n + (max_block - 1)) / max_block) * (1 + 4 + 1) * ZO_MAX_SPLITS + n + 1 + ftr;
}

size_t zo_memory_compress_dict(const uint8_t *in, size_t n, uint8_t *out, size_t cap, unsigned flags,
                               unsigned max_block, const uint8_t *dict, int dict_size) {
   const unsigned bs = clamp_block(max_block);
   const size_t blockbuf_cap = 1 + (size_t)bs + 5 * ((size_t)bs / 65535 + 1);   /* libzultra.c:115 */
   zo_stream_t s;
   uint32_t sum;
   size_t done = 0;
   int prev = 0, fail = 0;
   uint8_t *first_win = NULL, *tmp;
   zo_match_t *match;

   zo_init_tables();
   if (n == 0) return (size_t)-1;   /* libzultra.c:275: nothing is ever finalized for empty input */
   if (!dict || dict_size <= 0) {
      dict = NULL;
      dict_size = 0;
   }
   if (dict_size > ZO_HISTORY) return (size_t)-1;   /* caller contract (dictionary.c:49 trims to 32 KiB) */

   memset(&s, 0, sizeof(s));
   s.out = out;
   s.cap = cap;

   /* header — frame.c:387-446 */
   if (flags & 2) {
      static const uint8_t gz[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 2, 255};
      int k;
      for (k = 0; k < 10; k++) stream_byte(&s, gz[k]);
      sum = 0;
   }
   else if (flags & 1) {
      unsigned b0 = 0x78, b1 = 0xc0 | (dict ? 0x20 : 0);
      b1 |= (31 - ((b0 << 8 | b1) % 31)) & 0x1f;
      stream_byte(&s, (uint8_t)b0);
      stream_byte(&s, (uint8_t)b1);
      if (dict) {
         uint32_t da = zo_adler32(1, dict, (size_t)dict_size);
         stream_byte(&s, (uint8_t)(da >> 24));
         stream_byte(&s, (uint8_t)(da >> 16));
         stream_byte(&s, (uint8_t)(da >> 8));
         stream_byte(&s, (uint8_t)da);
      }
      sum = 1;
   }
   else
      sum = 0;

   match = (zo_match_t *)malloc(sizeof(zo_match_t) * 8 * (size_t)bs);
   tmp = (uint8_t *)malloc((size_t)bs + 64);
   if (dict) {
      size_t first = n < bs ? n : bs;
      first_win = (uint8_t *)malloc((size_t)dict_size + first);
      memcpy(first_win, dict, (size_t)dict_size);
      memcpy(first_win + dict_size, in, first);
      prev = dict_size;
   }

   while (done < n && !fail) {
      const size_t nblk = (n - done) < bs ? (n - done) : bs;
      const int last_block = (done + nblk) >= n;
      const uint8_t *win = (done == 0 && first_win) ? first_win : (in + done - prev);
      int splits[ZO_MAX_SPLITS], nsplits, si, at = 0;
      size_t block_base = s.pos;   /* the reference's out_buffer restarts at offset 0 per max-block */

      if (flags & 2)
         sum = zo_crc32(sum, in + done, nblk);
      else if (flags & 1)
         sum = zo_adler32(sum, in + done, nblk);

      zo_find_matches(win, prev, (int)nblk, match);
      nsplits = zo_block_split(win, match, prev, (int)nblk, splits);
      if (nsplits < 0) {
         fail = 1;
         break;
      }

      for (si = 0; si < nsplits && !fail; si++) {
         const int sub = splits[si] - (at + prev);
         const int is_final = last_block && (at + sub) >= (int)nblk;   /* libzultra.c:328 */
         int sc, dc, dyn, r;
         uint64_t nbits = 0;
         size_t o0;
         int c0;

         dyn = zo_subblock_costs(win, match, prev, prev + at, sub, &sc, &dc);
         r = zo_subblock_deflate(win, match, prev, prev + at, sub, dyn, tmp, (size_t)sub + 8, &nbits, NULL, NULL, NULL);

         /* where the 3 header bits would leave the writer (libzultra.c:329-337) */
         c0 = (s.nacc + 3) & 7;
         o0 = (s.pos - block_base) + (size_t)((s.nacc + 3) >> 3);
         if (o0 > blockbuf_cap) {   /* put_bits of the header itself fails -> ZULTRA_ERROR_DST */
            fail = 1;
            break;
         }

         if (r == 0 && (((uint64_t)c0 + nbits) >> 3) <= (uint64_t)sub &&
             o0 + (size_t)(((uint64_t)c0 + nbits) >> 3) <= blockbuf_cap) {
            stream_bits(&s, (uint32_t)is_final, 1);
            stream_bits(&s, (uint32_t)(1 + dyn), 2);
            stream_append(&s, tmp, nbits);
         }
         else {
            /* stored fallback, pieces of at most 65535 bytes (libzultra.c:350-397) */
            int rem = sub, off = 0;
            while (rem && !fail) {
               int piece = rem > 65535 ? 65535 : rem;
               int piece_final = (rem > 65535) ? 0 : is_final;
               int k;
               stream_bits(&s, (uint32_t)piece_final, 1);
               stream_bits(&s, 0, 2);
               stream_pad(&s);
               if ((s.pos - block_base) + 4 + (size_t)piece > blockbuf_cap) {
                  fail = 1;
                  break;
               }
               stream_byte(&s, (uint8_t)(piece & 0xff));
               stream_byte(&s, (uint8_t)(piece >> 8));
               stream_byte(&s, (uint8_t)((piece & 0xff) ^ 0xff));
               stream_byte(&s, (uint8_t)((piece >> 8) ^ 0xff));
               for (k = 0; k < piece; k++) stream_byte(&s, in[done + (size_t)at + (size_t)off + (size_t)k]);
               off += piece;
               rem -= piece;
            }
         }
         at += sub;
      }

      done += nblk;
      prev = (nblk > ZO_HISTORY) ? ZO_HISTORY : (int)nblk;   /* libzultra.c:406-408 */
   }

   free(first_win);
   free(tmp);
   free(match);
   if (fail) return (size_t)-1;

   stream_pad(&s);   /* libzultra.c:414-417 */

   /* footer — frame.c:509-547 */
   if (flags & 2) {
      int k;
      for (k = 0; k < 4; k++) stream_byte(&s, (uint8_t)(sum >> (8 * k)));
      for (k = 0; k < 4; k++) stream_byte(&s, (uint8_t)((uint64_t)n >> (8 * k)));
   }
   else if (flags & 1) {
      int k;
      for (k = 3; k >= 0; k--) stream_byte(&s, (uint8_t)(sum >> (8 * k)));
   }
   if (s.overflow) return (size_t)-1;
   return s.pos;
}

size_t zo_memory_compress(const uint8_t *in, size_t n, uint8_t *out, size_t cap, unsigned flags,
                          unsigned max_block) {
   return zo_memory_compress_dict(in, n, out, cap, flags, max_block, NULL, 0);
}
