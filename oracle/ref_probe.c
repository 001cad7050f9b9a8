/*
 * ref_probe.c — TEST INFRASTRUCTURE ONLY. Our own glue, compiled together with the reference's sources
 * (never copied; see oracle/Makefile `ref`) into oracle/_ref/libzultra_ref.so.
 *
 * The reference keeps its intermediates (match[], split offsets, per-sub-block costs and bits) inside
 * `zultra_compressor_t` (src/private.h:64-99). This file includes the reference's own headers to read
 * them, so that the restatement in zultra_oracle.c — and through it the HIP kernels — can be pinned
 * stage by stage, not only on final bytes. Each probe drives exactly the call sequence of
 * zultra_stream_compress (src/libzultra.c:287-343).
 */
#include <stdlib.h>
#include <string.h>
#include "libzultra.h"
#include "private.h"
#include "matchfinder.h"
#include "blockdeflate.h"

typedef struct {
   zultra_stream_t strm;
   int max_block;
   int prev, n;           /* window currently analysed */
   const unsigned char *win;
} ref_probe_t;

ref_probe_t *ref_probe_create(int nMaxBlockSize) {
   ref_probe_t *p = (ref_probe_t *)calloc(1, sizeof(ref_probe_t));
   if (!p) return NULL;
   if (zultra_stream_init(&p->strm, 0, (unsigned int)nMaxBlockSize) != ZULTRA_OK) {
      free(p);
      return NULL;
   }
   p->max_block = (int)p->strm.state->max_block_size;
   return p;
}

void ref_probe_destroy(ref_probe_t *p) {
   if (p) {
      zultra_stream_end(&p->strm);
      free(p);
   }
}

/* libzultra.c:287-293: suffix array + interval tree, skip history, find all matches of the block.
 * win = prev history bytes followed by n block bytes (prev <= 32768, n <= max block). The window pointer
 * must stay valid for later probes. */
int ref_probe_analyse(ref_probe_t *p, const unsigned char *win, int prev, int n) {
   zultra_compressor_t *c = p->strm.state;
   if (prev < 0 || prev > HISTORY_SIZE || n <= 0 || n > p->max_block) return -1;
   if (zultra_build_suffix_array(c, win, prev + n)) return -2;
   if (prev) zultra_skip_matches(c, 0, prev);
   zultra_find_all_matches(c, prev, prev + n);
   p->win = win;
   p->prev = prev;
   p->n = n;
   return 0;
}

/* Copy the match rows of the block positions: out[(i*8+m)*2+0]=length, +1=offset, i relative to block. */
void ref_probe_get_matches(ref_probe_t *p, unsigned short *out) {
   zultra_compressor_t *c = p->strm.state;
   const zultra_match_t *m = c->match + ((size_t)p->prev << MATCHES_PER_OFFSET_SHIFT);
   size_t i, cnt = (size_t)p->n * NMATCHES_PER_OFFSET;
   for (i = 0; i < cnt; i++) {
      out[2 * i] = m[i].length;
      out[2 * i + 1] = m[i].offset;
   }
}

/* libzultra.c:303: block splitter. Returns the number of entries written (last one = prev+n). */
int ref_probe_split(ref_probe_t *p, int *pSplitOffset /* [MAX_SPLITS] */) {
   return zultra_block_split(p->strm.state, p->win, p->prev, p->n, MAX_SPLITS, pSplitOffset);
}

/* libzultra.c:317-324: static / dynamic cost of one sub-block and the resulting type decision. */
int ref_probe_costs(ref_probe_t *p, int nStart, int nSize, int *pStatic, int *pDynamic) {
   zultra_compressor_t *c = p->strm.state;
   int s = 0, d = 0;
   if (zultra_block_prepare_cost_evaluation(c, p->win, nStart, nSize) < 0 ||
       zultra_block_evaluate_static_cost(&c->literalsEncoder, &c->offsetEncoder, &s) < 0 ||
       zultra_huffman_encoder_estimate_dynamic_codelens(&c->literalsEncoder) < 0 ||
       zultra_huffman_encoder_estimate_dynamic_codelens(&c->offsetEncoder) < 0 ||
       zultra_block_evaluate_dynamic_cost(&c->literalsEncoder, &c->offsetEncoder, &d) < 0)
      return -1;
   *pStatic = s;
   *pDynamic = d;
   return (s <= d) ? 0 : 1;   /* nIsDynamic */
}

/* libzultra.c:343: encode one sub-block body (no BFINAL/BTYPE bits) from bit phase 0 into out.
 * Returns 0 and the exact bit count, or -1 when zultra_block_deflate fails. Also copies out the final
 * parse (best_match) when pBest != NULL: pBest[(i-nStart)*2+0]=length, +1=offset. */
int ref_probe_deflate(ref_probe_t *p, int nStart, int nSize, int nIsDynamic,
                      unsigned char *out, int nOutCap, long long *pBits, unsigned short *pBest) {
   zultra_compressor_t *c = p->strm.state;
   zultra_bitwriter_t bw;
   int r, i;

   zultra_bitwriter_init(&bw, out, 0, nOutCap);
   r = zultra_block_deflate(c, &bw, p->win, nStart, nSize, nIsDynamic);
   if (r < 0) return -1;
   *pBits = (long long)bw.nOutOffset * 8 + bw.nEncBitCount;
   if (zultra_bitwriter_flush_bits(&bw) < 0) return -1;
   if (pBest) {
      for (i = 0; i < nSize; i++) {
         pBest[2 * i] = c->best_match[nStart + i].length;
         pBest[2 * i + 1] = c->best_match[nStart + i].offset;
      }
   }
   return 0;
}

/* Final code lengths of the last ref_probe_deflate call (288 lit/len + 32 dist). */
void ref_probe_get_codelens(ref_probe_t *p, int *pLit /*[288]*/, int *pDist /*[32]*/) {
   zultra_compressor_t *c = p->strm.state;
   memcpy(pLit, c->literalsEncoder.nCodeLength, 288 * sizeof(int));
   memcpy(pDist, c->offsetEncoder.nCodeLength, 32 * sizeof(int));
}

/* Whole-stream entry with a preset dictionary (libzultra.c:177 + 601), for dictionary fixtures. */
size_t ref_probe_memory_compress_dict(const unsigned char *in, size_t nIn, unsigned char *out, size_t nOutCap,
                                      unsigned int nFlags, unsigned int nMaxBlockSize,
                                      const void *pDict, int nDictSize) {
   zultra_stream_t strm;
   zultra_status_t st;
   memset(&strm, 0, sizeof(strm));
   if (zultra_stream_init(&strm, nFlags, nMaxBlockSize)) return (size_t)-1;
   if (pDict && nDictSize > 0 && zultra_stream_set_dictionary(&strm, pDict, nDictSize) != ZULTRA_OK) {
      zultra_stream_end(&strm);
      return (size_t)-1;
   }
   strm.next_in = in;
   strm.avail_in = nIn;
   strm.next_out = out;
   strm.avail_out = nOutCap;
   st = zultra_stream_compress(&strm, ZULTRA_FINALIZE);
   zultra_stream_end(&strm);
   if (st != ZULTRA_STREAM_END) return (size_t)-1;
   return nOutCap - strm.avail_out;
}
