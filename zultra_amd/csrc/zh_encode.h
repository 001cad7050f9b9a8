// zh_encode.h — stage 3 of the hot path: one wave turns one sub-block into its deflate bit string.
//
// Replaces, per sub-block, the calls at reference src/libzultra.c:317-324 (static vs dynamic decision:
// zultra_block_prepare_cost_evaluation, zultra_block_evaluate_static_cost, 2x estimate_dynamic_codelens,
// zultra_block_evaluate_dynamic_cost) and :343 zultra_block_deflate (src/blockdeflate.c:827-997):
// tentative Huffman codes from the greedy parse, four backward optimal-parse passes each followed by a
// histogram of the chosen parse and a rebuild of the codes, literalisation of cheap matches, the RLE-friendly
// alternative tables, the 20-way search over code-length RLE masks, header and token emission.
//
// Mapping onto a CDNA4 wave:
//  * optimal parse: the recurrence is serial in the position, but the match candidates of positions i, i-1, i-2
//    only read cost[>= i+1]; so each step prices THREE positions, one per 16-lane DPP row. Candidates collapse to
//    one per length (cheapest eligible distance, found by a prefix-min when a tile of 63 positions is staged), so a
//    row's 16 lanes cover lengths 3..18 in one go; one row-wise DPP min picks each row's winner with the reference's
//    tie rule in the key (cost, slot, longest-first); the three literal-vs-match decisions chain on the scalar unit.
//    Costs live in a 512-entry LDS ring; match rows arrive with coalesced loads issued one tile ahead.
//  * every forward walk over the chosen parse (histogram, literalisation, emission) follows the token chain on
//    the scalar unit (v_readlane over a 64-position tile held in one VGPR) and lets the token lanes work in
//    parallel; emission gets its bit offsets from a wave prefix sum and ORs 48-bit token codes into an LDS
//    staging window that is flushed to HBM in whole dwords.
//  * the 20 RLE-mask candidates are evaluated by 20 lanes at once, each with a private 19-symbol encoder.
#pragma once
#include <zh_platform.h>
#include "zh_common.h"
#include "zh_huffman.h"
#include "zh_split.h"

#define ZH_OBUF_WORDS 128   // LDS staging window for token bits: 64 tokens x 48 bits = 96 dwords + carry

struct zh_enc_ws_t {
   int32_t lit_freq[ZH_NLIT], dist_freq[ZH_NDIST];
   uint8_t lit_len[ZH_NLIT], dist_len[ZH_NDIST];
   uint16_t lit_code[ZH_NLIT], dist_code[ZH_NDIST];
   int32_t alt_lit_freq[ZH_NLIT], alt_dist_freq[ZH_NDIST];
   uint8_t alt_lit_len[ZH_NLIT], alt_dist_len[ZH_NDIST];
   uint16_t alt_lit_code[ZH_NLIT], alt_dist_code[ZH_NDIST];
   uint8_t lens[ZH_NLIT + ZH_NDIST];
   uint8_t keep[ZH_NLIT];
   uint8_t lencost[256];             // price of length e+3 under the current codes (blockdeflate.c:216-219)
   uint8_t distcost[ZH_NDIST];       // price of a distance symbol incl. extra bits (blockdeflate.c:127-136)
   int32_t cost[512];                // ring of cost[i & 511] (the reference's cost[] at blockdeflate.c:255)
   uint32_t tile[64 * ZH_NMATCH];    // staged match rows: len(9) | dist symbol(5)<<9 | offset<<16
   uint32_t best_tile[64];
   uint32_t npos[64];                // per staged position: number of long slots | longest short length << 4
   uint8_t bo[64 * 40];              // per staged position and length k-3: cheapest distance price << 3 | slot
   uint32_t obuf[ZH_OBUF_WORDS];
   zh_huff_scratch_t sc;
   zh_cl_t cl;
   zh_cl_t cl_work[20];
   int32_t tmp;
};

// ---- price tables of the current codes ------------------------------------------------------------------
__device__ inline void zh_refresh_prices_wave(zh_enc_ws_t *ws) {
   const int lane = (int)zh_lane();
   for (int e = lane; e < 256; e += 64) {
      int idx = zh_len_idx((uint32_t)e + 3);
      ws->lencost[e] = (uint8_t)(ws->lit_len[257 + idx] + zh_lenidx_xbits(idx));
   }
   if (lane < ZH_NDIST) ws->distcost[lane] = (uint8_t)(ws->dist_len[lane] + zh_dist_xbits(lane));
   zh_sync();
}

// ---- backward optimal parse (blockdeflate.c:254-323) ----------------------------------------------------
// cost[i] = min(literal(i) + cost[i+1], min over match slots m and lengths k of len(k) + dist(m) + cost[i+k]).
//
// Two observations shape the kernel (a lone wave issues ~1 instruction per 4-5 clocks, so the instruction count per
// position is what matters, not LDS latency):
//  1. Per length k only the cheapest distance among the slots that reach k can win, and with the reference's order
//     (slot ascending, then length descending, strict improvement) the winner is the lexicographic minimum of
//     (cost, slot, -k). So the candidates of a position collapse to one per length: while a tile of 63 positions is
//     staged (off the serial path) every position gets a byte table bo[k-3] = (cheapest distance price << 3 | slot)
//     for k = 3..39, built with a prefix-min over its slots. Slots stored with length >= 40 are only ever tried at
//     their full (end-clamped) length (blockdeflate.c:286-297); they keep their own entries ("long slots").
//  2. The match part of positions i, i-1, i-2 only needs cost[>= i+1] (k >= 3), so three positions are priced per
//     step, one per 16-lane DPP row (row 3 idles): lane s of a row prices k = 3+s (and 19+s, 35+s when the position
//     reaches that far), one row-wise DPP min yields each row's winner, and the three literal-vs-match decisions
//     chain on the scalar unit.
#define ZH_DP_TILE 63   // positions staged per tile (21 triples)
#define ZH_DP_BO 40     // bytes of per-length table per position (k = 3..39 -> 37 used)

// Global loads of one tile (8 match-row dwords per lane + the lane's byte), issued together so that their latency
// overlaps the 21 steps of the previous tile. Lane l owns slot (l & 7) of positions base + q*8 + (l >> 3).
struct zh_dp_regs_t {
   uint32_t raw[ZH_NMATCH];
   uint32_t byte;
};
__device__ __forceinline__ void zh_dp_fetch(zh_dp_regs_t &r, const uint8_t *win, const uint32_t *rows, uint32_t prev, int64_t base,
                                            int64_t lo, uint32_t lane) {
#pragma unroll
   for (uint32_t q = 0; q < ZH_NMATCH; q++) {
      const uint32_t idx = q * 64 + lane;
      const int64_t pos = base + (idx >> 3);
      const bool ok = idx < ZH_DP_TILE * ZH_NMATCH && pos >= lo;
      const int64_t cp = ok ? pos : lo;                       // clamped address: the load is always legal
      const uint32_t v = rows[(uint64_t)(cp - prev) * ZH_NMATCH + (idx & 7)];
      r.raw[q] = ok ? v : 0;
   }
   const int64_t bp = base + lane;
   const bool okb = lane < ZH_DP_TILE && bp >= lo;
   const uint32_t bv = win[okb ? bp : lo];
   r.byte = okb ? bv : 0x100u;
}

__device__ inline void zh_optimal_parse_wave(zh_enc_ws_t *ws, const uint8_t *win, const uint32_t *rows /* row r = pos - prev */,
                                             uint32_t prev, uint32_t start, uint32_t end, uint32_t *best_out /* index = pos - prev */,
                                             uint64_t *loop_clocks = nullptr) {
   const uint32_t lane = zh_lane();
   const uint32_t row = lane >> 4, s = lane & 15;
   uint64_t loop_acc = 0;
   if (end <= start) return;
   zh_refresh_prices_wave(ws);
   // this lane's length prices: k = 3+s, 19+s, 35+s
   const uint32_t lc0 = ws->lencost[s], lc1 = ws->lencost[16 + s], lc2 = ws->lencost[32 + (s & 7)];

   int32_t cost_next = 0;   // cost[top+1], wave-uniform
   if (lane == 0) ws->cost[end & 511] = 0;
   zh_sync();

   zh_dp_regs_t regs;
   {
      const int64_t hi0 = (int64_t)end - 1, base0 = hi0 - (ZH_DP_TILE - 1);
      zh_dp_fetch(regs, win, rows, prev, base0, base0 > (int64_t)start ? base0 : (int64_t)start, lane);
   }

   for (int64_t hi = (int64_t)end - 1; hi >= (int64_t)start; hi -= ZH_DP_TILE) {
      const int64_t base64 = hi - (ZH_DP_TILE - 1);                     // tile index a <-> window position base + a
      const int32_t base = (int32_t)base64;
      const int32_t alo = base64 >= (int64_t)start ? 0 : (int32_t)((int64_t)start - base64);   // first tile index in range
      const int32_t rb = (int32_t)end - base;                           // room of tile index a = rb - a

      // ---- stage the tile: per-slot entries, per-length tables, per-position summaries ---------------------------
      const uint32_t m8 = lane & 7;
#pragma unroll
      for (uint32_t q = 0; q < ZH_NMATCH; q++) {
         const uint32_t a = q * 8 + (lane >> 3);
         const uint32_t raw = regs.raw[q];
         const uint32_t len = raw & 0xffffu, off = raw >> 16;
         const bool valid = len >= ZH_MIN_MATCH;
         const bool is_long = len >= ZH_LEAVE_ALONE;
         const uint32_t oc = valid ? (uint32_t)ws->distcost[zh_dist_sym(off)] : 31u;
         // inclusive prefix-min over the SHORT slots 0..m of this position of (price << 3 | slot): DPP row shifts
         uint32_t pm = (valid && !is_long) ? ((oc << 3) | m8) : 0xFFu;
         {
            const uint32_t o1 = zh_row_shr<1>(pm);
            if (m8 >= 1) pm = min(pm, o1);
            const uint32_t o2 = zh_row_shr<2>(pm);
            if (m8 >= 2) pm = min(pm, o2);
            const uint32_t o4 = zh_row_shr<4>(pm);
            if (m8 >= 4) pm = min(pm, o4);
         }
         const uint32_t nxt = zh_row_shl<1>(len);                            // length stored in slot m+1
         const uint32_t below = (m8 < 7 && nxt >= ZH_MIN_MATCH) ? nxt : 2u;  // lengths <= below belong to later slots
         const uint64_t longmask = zh_ballot(is_long);
         const uint64_t validmask = zh_ballot(valid);
         if (a < ZH_DP_TILE) {
            ws->tile[a * ZH_NMATCH + m8] = valid ? (len | (oc << 9) | (off << 16)) : 0;
            if (valid && !is_long)
               for (uint32_t k = below + 1; k <= len; k++) ws->bo[a * ZH_DP_BO + k - 3] = (uint8_t)pm;
            if (m8 == 0) {
               const uint32_t nlong = (uint32_t)__popc((uint32_t)((longmask >> (lane & 56u)) & 0xffu));     // long slots are a prefix
               const uint32_t nvalid = (uint32_t)__popc((uint32_t)((validmask >> (lane & 56u)) & 0xffu));
               ws->npos[a] = nlong | (nvalid << 4);
            }
         }
      }
      zh_sync();
      for (uint32_t a = lane; a < ZH_DP_TILE; a += 64) {
         const uint32_t info = ws->npos[a];
         const uint32_t nlong = info & 15u, nvalid = info >> 4;
         const uint32_t kmax = nvalid > nlong ? (ws->tile[a * ZH_NMATCH + nlong] & 511u) : 0u;   // longest short length
         ws->npos[a] = nlong | (kmax << 4);
      }
      const uint32_t litcost = regs.byte < 256 ? ws->lit_len[regs.byte] : 0;
      zh_sync();
      // start the next tile's loads now; they complete while this tile is priced
      if (base64 - 1 >= (int64_t)start) {
         const int64_t nbase = base64 - ZH_DP_TILE;
         zh_dp_fetch(regs, win, rows, prev, nbase, nbase > (int64_t)start ? nbase : (int64_t)start, lane);
      }

      // ---- price the tile, three positions per step ---------------------------------------------------------------
      const uint64_t t_loop = loop_clocks ? zh_clock() : 0;
      const uint32_t cb = (uint32_t)base + 3 + s;          // cost index of this lane's first length: (cb + a) & 511
      const bool row_ok = row < 3;
      for (int32_t t = ZH_DP_TILE - 1; t >= alo; t -= 3) {
         const int32_t a = t - (int32_t)row;
         const bool active = row_ok && a >= alo;
         uint32_t key = 0xFFFFFFFFu;
         uint32_t info = 0, kmax = 0;
         if (active) {
            info = ws->npos[a];
            kmax = min(info >> 4, (uint32_t)(rb - a));              // end clamp (blockdeflate.c:283-284)
            if (3 + s <= kmax) {
               const uint32_t b = ws->bo[(uint32_t)a * ZH_DP_BO + s];
               const uint32_t c = lc0 + (b >> 3) + (uint32_t)ws->cost[(cb + (uint32_t)a) & 511];
               key = (c << 9) | ((b & 7u) << 6) | (36u - s);        // 39 - k
            }
         }
         // rarely needed parts: lengths 19..39, and slots stored with length >= 40
         if (zh_ballot(kmax > 18u || (info & 15u) != 0)) {
            if (active) {
               const uint32_t room = (uint32_t)(rb - a);
               if (19 + s <= kmax) {
                  const uint32_t b = ws->bo[(uint32_t)a * ZH_DP_BO + 16 + s];
                  const uint32_t c = lc1 + (b >> 3) + (uint32_t)ws->cost[(cb + 16 + (uint32_t)a) & 511];
                  key = min(key, (c << 9) | ((b & 7u) << 6) | (20u - s));
               }
               if (s < 5 && 35 + s <= kmax) {
                  const uint32_t b = ws->bo[(uint32_t)a * ZH_DP_BO + 32 + s];
                  const uint32_t c = lc2 + (b >> 3) + (uint32_t)ws->cost[(cb + 32 + (uint32_t)a) & 511];
                  key = min(key, (c << 9) | ((b & 7u) << 6) | (4u - s));
               }
               if (s < (info & 15u)) {                              // long slot s: full (clamped) length only
                  const uint32_t e = ws->tile[(uint32_t)a * ZH_NMATCH + s];
                  const uint32_t mlen = min(e & 511u, room);
                  uint32_t enc = mlen - ZH_MIN_MATCH;               // wraps below 3, then saturates (:289, :216-219)
                  if (enc > 255) enc = 255;
                  const uint32_t c = (uint32_t)ws->lencost[enc] + ((e >> 9) & 31u) +
                                     (uint32_t)ws->cost[((uint32_t)base + (uint32_t)a + mlen) & 511];
                  key = min(key, (c << 9) | (s << 6));
               }
            }
         }
         const uint32_t rkey = zh_row_min(key);   // every lane of a row now holds that row's best match candidate
         // literal first; a match must be strictly cheaper (:292,:307). An absent candidate (all ones) prices at 2^23-1.
         const int32_t m0 = (int32_t)(zh_readlane(rkey, 0) >> 9), m1 = (int32_t)(zh_readlane(rkey, 16) >> 9),
                       m2 = (int32_t)(zh_readlane(rkey, 32) >> 9);
         const int32_t l0 = (int32_t)zh_readlane(litcost, t) + cost_next;
         const int32_t c0 = min(l0, m0);
         const int32_t l1 = (int32_t)zh_readlane(litcost, (t - 1) & 63) + c0;
         const int32_t c1 = min(l1, m1);
         const int32_t l2 = (int32_t)zh_readlane(litcost, (t - 2) & 63) + c1;
         const int32_t c2 = min(l2, m2);
         if (active && s == 0) {
            const int32_t myc = row == 0 ? c0 : (row == 1 ? c1 : c2);
            const int32_t myl = row == 0 ? l0 : (row == 1 ? l1 : l2);
            ws->cost[((uint32_t)base + (uint32_t)a) & 511] = myc;
            ws->best_tile[a] = (myc < myl) ? rkey : 0xFFFFFFFFu;   // decoded when the tile is flushed
         }
         cost_next = (t - 2 >= alo) ? c2 : ((t - 1 >= alo) ? c1 : c0);
         zh_ballot(true);   // orders the LDS cost writes before the next step's reads (free in lock-step on the GPU)
      }
      if (loop_clocks) loop_acc += zh_clock() - t_loop;
      zh_sync();
      // flush: decode the winning (slot, length) of each position and store the parse
      if ((int32_t)lane < ZH_DP_TILE && (int32_t)lane >= alo) {
         const uint32_t kk = ws->best_tile[lane];
         uint32_t pick = 0;
         if (kk != 0xFFFFFFFFu) {
            const uint32_t m = (kk >> 6) & 7u;
            const uint32_t e = ws->tile[lane * ZH_NMATCH + m];
            const uint32_t len = (m < (ws->npos[lane] & 15u)) ? min(e & 511u, (uint32_t)(rb - (int32_t)lane)) : (39u - (kk & 63u));
            pick = len | (e & 0xffff0000u);
         }
         best_out[(uint32_t)(base + (int32_t)lane) - prev] = pick;
      }
      zh_sync();
   }
   if (loop_clocks && lane == 0) *loop_clocks = loop_acc;
}

// ---- histogram of the chosen parse (blockdeflate.c:371-400) ---------------------------------------------
__device__ inline void zh_parse_histogram_wave(zh_enc_ws_t *ws, const uint8_t *win, uint32_t prev, uint32_t start, uint32_t end,
                                               const uint32_t *best) {
   const uint32_t lane = zh_lane();
   for (uint32_t s2 = lane; s2 < ZH_NLIT; s2 += 64) ws->lit_freq[s2] = 0;
   if (lane < ZH_NDIST) ws->dist_freq[lane] = 0;
   zh_sync();
   uint32_t carry = 0;
   for (uint32_t base = start; base < end; base += 64) {
      const uint32_t limit = min(64u, end - base);
      const uint32_t pos = base + lane;
      uint32_t b = 0, byte = 0;
      if (pos < end) {
         b = best[pos - prev];
         byte = win[pos];
      }
      const uint32_t len = b & 0xffffu;
      uint64_t mask = zh_chain_mask(len, carry, limit);
      if ((mask >> lane) & 1ull) {
         if (len >= ZH_MIN_MATCH) {
            atomicAdd(&ws->lit_freq[257 + zh_len_idx(len)], 1);
            atomicAdd(&ws->dist_freq[zh_dist_sym(b >> 16)], 1);
         }
         else
            atomicAdd(&ws->lit_freq[byte], 1);
      }
   }
   zh_sync();
   if (lane == 0) ws->lit_freq[ZH_EOB] += 1;
   zh_sync();
}

// ---- matches that are cheaper as literals (blockdeflate.c:410-458) --------------------------------------
__device__ inline void zh_literalize_wave(zh_enc_ws_t *ws, const uint8_t *win, uint32_t prev, uint32_t start, uint32_t end,
                                          uint32_t *best) {
   const uint32_t lane = zh_lane();
   zh_refresh_prices_wave(ws);
   uint32_t carry = 0;
   for (uint32_t base = start; base < end; base += 64) {
      const uint32_t limit = min(64u, end - base);
      const uint32_t pos = base + lane;
      uint32_t b = (pos < end) ? best[pos - prev] : 0;
      const uint32_t len = b & 0xffffu;
      uint64_t mask = zh_chain_mask(len, carry, limit);
      if (((mask >> lane) & 1ull) && len >= ZH_MIN_MATCH) {
         const uint32_t off = b >> 16;
         if (off >= 1 && off <= ZH_MAX_DIST) {
            const uint32_t mcost = (uint32_t)ws->lencost[min(len - ZH_MIN_MATCH, 255u)] + (uint32_t)ws->distcost[zh_dist_sym(off)];
            uint32_t lcost = 0, j = 0;
            bool usable = true;
            for (; j < len && lcost < mcost; j++) {
               uint32_t l = ws->lit_len[win[pos + j]];
               if (l == 0) {
                  usable = false;   // a byte without a code keeps the match (:436-440)
                  break;
               }
               lcost += l;
            }
            if (usable && lcost < mcost)
               for (j = 0; j < len; j++) best[pos - prev + j] &= 0xffff0000u;   // length := 0 (:449-451)
         }
      }
   }
   __threadfence_block();
   zh_sync();
}

// ---- bit output ----------------------------------------------------------------------------------------
// Serial writer used by lane 0 for the block header; hands its partial dword over to the token emitter.
struct zh_bitw_t {
   uint32_t *out;        // dword-aligned slot in HBM
   uint32_t cap_bits;
   uint64_t acc;
   uint32_t nacc;
   uint32_t nbits;       // total so far
   __device__ __forceinline__ void put(uint32_t v, uint32_t n) {
      acc |= (uint64_t)v << nacc;
      nacc += n;
      nbits += n;
      if (nacc >= 32) {
         if (nbits - nacc + 32 <= cap_bits) out[(nbits - nacc) >> 5] = (uint32_t)acc;
         acc >>= 32;
         nacc -= 32;
      }
   }
};
struct zh_cl_write_sink {
   const zh_cl_t *h;
   zh_bitw_t *w;
   __device__ __forceinline__ void put(int sym, int xval, int xbits) {
      w->put(h->code[sym], h->len[sym]);
      if (xbits) w->put((uint32_t)xval, (uint32_t)xbits);
   }
};

// Token emission (blockdeflate.c:471-507). bitpos = bits already in the slot; obuf[0] holds the partial dword.
// Returns the final bit count (keeps counting past the capacity; nothing is stored past it).
__device__ inline uint32_t zh_emit_tokens_wave(zh_enc_ws_t *ws, const uint8_t *win, uint32_t prev, uint32_t start, uint32_t end,
                                               const uint32_t *best, uint32_t *out, uint32_t cap_bits, uint32_t bitpos) {
   const uint32_t lane = zh_lane();
   uint32_t carry = 0;
   for (uint32_t base = start; base < end; base += 64) {
      const uint32_t limit = min(64u, end - base);
      const uint32_t pos = base + lane;
      uint32_t b = 0, byte = 0;
      if (pos < end) {
         b = best[pos - prev];
         byte = win[pos];
      }
      const uint32_t len = b & 0xffffu;
      uint64_t mask = zh_chain_mask(len, carry, limit);
      uint64_t code = 0;
      uint32_t nb = 0;
      if ((mask >> lane) & 1ull) {
         if (len >= ZH_MIN_MATCH) {
            const uint32_t off = b >> 16;
            const int li = zh_len_idx(len), ds = zh_dist_sym(off);
            const uint32_t lx = (uint32_t)zh_lenidx_xbits(li), dx = (uint32_t)zh_dist_xbits(ds);
            code = ws->lit_code[257 + li];
            nb = ws->lit_len[257 + li];
            code |= (uint64_t)(len - zh_lenidx_base(li)) << nb;
            nb += lx;
            code |= (uint64_t)ws->dist_code[ds] << nb;
            nb += ws->dist_len[ds];
            code |= (uint64_t)(off - zh_dist_base(ds)) << nb;
            nb += dx;
         }
         else {
            code = ws->lit_code[byte];
            nb = ws->lit_len[byte];
         }
      }
      const uint32_t offs = zh_wave_excl_sum(nb);
      const uint32_t total = zh_wave_sum(nb);
      const uint32_t word0 = bitpos >> 5;
      if (nb) {
         // a token is at most 15+5+15+13 = 48 bits; shifted by < 32 it spans at most 3 dwords of the window
         const uint32_t bp = bitpos + offs;
         const uint32_t w = (bp >> 5) - word0, sh = bp & 31;
         const uint64_t lo = code << sh;
         atomicOr(&ws->obuf[w], (uint32_t)lo);
         atomicOr(&ws->obuf[w + 1], (uint32_t)(lo >> 32));
         if (sh) atomicOr(&ws->obuf[w + 2], (uint32_t)(code >> (64 - sh)));
      }
      zh_sync();
      const uint32_t newpos = bitpos + total;
      const uint32_t nfull = (newpos >> 5) - word0;
      for (uint32_t k = lane; k < nfull; k += 64)
         if (((word0 + k + 1) << 5) <= cap_bits) out[word0 + k] = ws->obuf[k];
      zh_sync();
      const uint32_t partial = ws->obuf[nfull];
      zh_sync();
      for (uint32_t k = lane; k < ZH_OBUF_WORDS; k += 64) ws->obuf[k] = (k == 0) ? partial : 0;
      zh_sync();
      bitpos = newpos;
   }
   // end-of-block symbol, then flush the partial dword
   if (lane == 0) {
      uint64_t acc = ws->obuf[0] | ((uint64_t)ws->lit_code[ZH_EOB] << (bitpos & 31));
      uint32_t nacc = (bitpos & 31) + ws->lit_len[ZH_EOB];
      uint32_t w = bitpos >> 5;
      bitpos += ws->lit_len[ZH_EOB];
      if (((w + 1) << 5) <= cap_bits) out[w] = (uint32_t)acc;
      if (nacc > 32 && ((w + 2) << 5) <= cap_bits) out[w + 1] = (uint32_t)(acc >> 32);
      ws->tmp = (int32_t)bitpos;
   }
   zh_sync();
   uint32_t r = (uint32_t)ws->tmp;
   zh_sync();
   return r;
}

// ---- the kernel ------------------------------------------------------------------------------------------
// sub-block work item produced by zh_plan_subblocks
struct zh_work_t {
   uint32_t block, start, size;   // start = absolute window offset
   uint32_t tok0, tok1;           // greedy token range
   uint64_t out_off;              // byte offset of the bit slot inside the batch payload (multiple of 4)
   uint32_t out_cap;              // slot capacity in bytes (multiple of 4)
   uint32_t index;                // position of this sub-block in stream order
};

__global__ void __launch_bounds__(64)
zh_encode(const uint8_t *__restrict__ data, const zh_block_t *__restrict__ blocks, const zh_match_t *__restrict__ match,
          uint64_t match_stride, const uint16_t *__restrict__ tok_info, uint64_t tok_stride, const zh_work_t *__restrict__ work,
          uint32_t *best_all, uint64_t best_stride, uint8_t *payload, zh_subblock_t *results, uint64_t *prof) {
   __shared__ zh_enc_ws_t ws;
   const zh_work_t wk = work[blockIdx.x];
   const zh_block_t blk = blocks[wk.block];
   const uint8_t *win = data + blk.win_off;
   const uint32_t prev = blk.prev;
   const uint32_t *rows = (const uint32_t *)(match + (uint64_t)wk.block * match_stride);
   const uint16_t *ti = tok_info + (uint64_t)wk.block * tok_stride;
   uint32_t *best = best_all + (uint64_t)wk.block * best_stride;
   uint32_t *out = (uint32_t *)(payload + wk.out_off);
   const uint32_t cap_bits = wk.out_cap * 8;
   const uint32_t start = wk.start, end = wk.start + wk.size;
   const uint32_t lane = zh_lane();
   // optional phase profile: 16 shader-clock stamps per sub-block (prof == NULL in normal runs)
#define ZH_STAMP(k)                                                       \
   do {                                                                   \
      if (prof && lane == 0) prof[(uint64_t)wk.index * 16 + (k)] = zh_clock(); \
   } while (0)
   ZH_STAMP(0);

   // ---- libzultra.c:317-324: greedy histogram, static price, dynamic price -------------------------------
   for (uint32_t s = lane; s < ZH_NLIT; s += 64) ws.lit_freq[s] = 0;
   if (lane < ZH_NDIST) ws.dist_freq[lane] = 0;
   for (uint32_t k = lane; k < ZH_OBUF_WORDS; k += 64) ws.obuf[k] = 0;
   zh_sync();
   zh_token_histogram_wave(ti, wk.tok0, wk.tok1, ws.lit_freq, ws.dist_freq);
   if (lane == 0) ws.lit_freq[ZH_EOB] += 1;
   zh_sync();

   uint32_t sc_part = 0;   // blockdeflate.c:538-566
   for (uint32_t s = lane; s < 257 + 29; s += 64) {
      int xb = (s >= 257) ? zh_lenidx_xbits((int)s - 257) : 0;
      sc_part += (uint32_t)(ws.lit_freq[s] * (zh_static_lit_len((int)s) + xb));
   }
   if (lane < ZH_NDIST) sc_part += (uint32_t)(ws.dist_freq[lane] * (5 + zh_dist_xbits((int)lane)));
   const int static_cost = (int)zh_wave_sum(sc_part) + 3;
   const int dynamic_cost = zh_dynamic_cost_wave(ws.lit_freq, ws.dist_freq, ws.lit_len, ws.dist_len, ws.lens, &ws.cl, &ws.tmp,
                                                 &ws.sc, true);
   const uint32_t is_dynamic = (static_cost <= dynamic_cost) ? 0u : 1u;
   ZH_STAMP(1);

   uint32_t failed = 0;
   uint32_t bitpos = 0;

   if (!is_dynamic) {
      // ---- blockdeflate.c:836-858 -----------------------------------------------------------------------
      for (uint32_t s = lane; s < ZH_NLIT; s += 64) ws.lit_len[s] = (uint8_t)zh_static_lit_len((int)s);
      if (lane < ZH_NDIST) ws.dist_len[lane] = 5;
      zh_sync();
      zh_huff_static_codes_wave(ws.lit_len, ws.lit_code, ZH_NLIT, &ws.sc);
      zh_huff_static_codes_wave(ws.dist_len, ws.dist_code, ZH_NDIST, &ws.sc);
      zh_optimal_parse_wave(&ws, win, rows, prev, start, end, best);
   }
   else {
      // ---- blockdeflate.c:859-920: the greedy histogram is still in place -----------------------------------
      if (zh_huff_build_wave(ws.lit_freq, ws.lit_len, ws.lit_code, ZH_NLIT, 15, &ws.sc) < 0) failed = 1;
      if (zh_huff_build_wave(ws.dist_freq, ws.dist_len, ws.dist_code, ZH_NDIST, 15, &ws.sc) < 0) failed = 1;
      ZH_STAMP(2);
      for (int pass = 0; pass <= 3; pass++) {
         for (uint32_t s = lane; s < ZH_NLIT; s += 64)
            if (!ws.lit_len[s]) ws.lit_len[s] = 9;
         if (lane < ZH_NDIST && !ws.dist_len[lane]) ws.dist_len[lane] = 6;
         zh_sync();
         zh_optimal_parse_wave(&ws, win, rows, prev, start, end, best, (prof && pass == 3) ? &prof[(uint64_t)wk.index * 16 + 15] : nullptr);
         __threadfence_block();
         zh_sync();
         ZH_STAMP(3 + 2 * pass);
         zh_parse_histogram_wave(&ws, win, prev, start, end, best);
         if (pass == 3 && lane == 0) {
            int used = 0;
            for (int s = 0; used < 2 && s < ZH_NDIST - 2; s++)
               if (ws.dist_freq[s]) used++;
            if (used == 0)
               ws.dist_freq[0] = ws.dist_freq[1] = 1;
            else if (used == 1) {
               if (ws.dist_freq[0])
                  ws.dist_freq[1] = 1;
               else
                  ws.dist_freq[0] = 1;
            }
         }
         zh_sync();
         if (zh_huff_build_wave(ws.lit_freq, ws.lit_len, ws.lit_code, ZH_NLIT, 15, &ws.sc) < 0) failed = 1;
         if (zh_huff_build_wave(ws.dist_freq, ws.dist_len, ws.dist_code, ZH_NDIST, 15, &ws.sc) < 0) failed = 1;
         ZH_STAMP(4 + 2 * pass);
      }

      zh_literalize_wave(&ws, win, prev, start, end, best);   // histograms stay as they were (:923)
      ZH_STAMP(11);

      // ---- blockdeflate.c:925-945: RLE-friendlier alternative ------------------------------------------------
      {
         const int cur_cost = zh_dynamic_cost_wave(ws.lit_freq, ws.dist_freq, ws.lit_len, ws.dist_len, ws.lens, &ws.cl, &ws.tmp,
                                                   &ws.sc, false);
         for (uint32_t s = lane; s < ZH_NLIT; s += 64) ws.alt_lit_freq[s] = ws.lit_freq[s];
         if (lane < ZH_NDIST) ws.alt_dist_freq[lane] = ws.dist_freq[lane];
         zh_sync();
         if (lane == 0) {
            zh_smooth_for_rle_lane(ZH_NLIT, ws.alt_lit_freq, ws.keep);
            zh_smooth_for_rle_lane(ZH_NDIST, ws.alt_dist_freq, ws.keep);
         }
         zh_sync();
         if (zh_huff_build_wave(ws.alt_lit_freq, ws.alt_lit_len, ws.alt_lit_code, ZH_NLIT, 15, &ws.sc) < 0) failed = 1;
         if (zh_huff_build_wave(ws.alt_dist_freq, ws.alt_dist_len, ws.alt_dist_code, ZH_NDIST, 15, &ws.sc) < 0) failed = 1;
         const int alt_cost = zh_dynamic_cost_wave(ws.alt_lit_freq, ws.alt_dist_freq, ws.alt_lit_len, ws.alt_dist_len, ws.lens,
                                                   &ws.cl, &ws.tmp, &ws.sc, false);
         if (alt_cost < cur_cost) {
            for (uint32_t s = lane; s < ZH_NLIT; s += 64) {
               ws.lit_len[s] = ws.alt_lit_len[s];
               ws.lit_code[s] = ws.alt_lit_code[s];
            }
            if (lane < ZH_NDIST) {
               ws.dist_len[lane] = ws.alt_dist_len[lane];
               ws.dist_code[lane] = ws.alt_dist_code[lane];
            }
         }
         zh_sync();
      }

      ZH_STAMP(12);
      // ---- blockdeflate.c:947-992: header ---------------------------------------------------------------------
      const int nlit = zh_defined_count(ws.lit_len, ZH_NLIT, 257);
      const int ndist = zh_defined_count(ws.dist_len, ZH_NDIST, 1);
      for (int s = (int)lane; s < nlit; s += 64) ws.lens[s] = ws.lit_len[s];
      if ((int)lane < ndist) ws.lens[nlit + (int)lane] = ws.dist_len[lane];
      zh_sync();

      uint32_t mkey = 0xFFFFFFFFu;
      if (lane < 20) {
         const unsigned mask = lane < 8 ? lane : 9 + 2 * (lane - 8);   // 0..7, 9, 11, ..., 31 (:959)
         zh_cl_t *h = &ws.cl_work[lane];
         zh_cl_reset(h);
         zh_cl_count_sink cs{h};
         zh_cl_tokenize(ws.lens, nlit + ndist, mask, cs);
         if (zh_cl_build_lane(h, 7) < 0)
            mkey = 0xFFFFFFFEu;
         else {
            zh_cl_size_sink ss{h, 0};
            zh_cl_tokenize(ws.lens, nlit + ndist, mask, ss);
            mkey = ((uint32_t)ss.bits << 6) | (63u - lane);   // cheapest; among equals the last tried (:966)
         }
      }
      zh_sync();
      const uint32_t mbest = zh_wave_min(mkey);
      const uint32_t anybad = zh_wave_sum(mkey == 0xFFFFFFFEu ? 1u : 0u);
      if (anybad) failed = 1;
      const uint32_t best_lane = 63u - (mbest & 63u);
      const unsigned best_mask = best_lane < 8 ? best_lane : 9 + 2 * (best_lane - 8);

      if (lane == 0) {
         const zh_cl_t *h = &ws.cl_work[best_lane < 20 ? best_lane : 0];
         const int ncl = zh_cl_raw_table_size(h);
         if (nlit > 286 || ndist > 30 || ncl > ZH_NCL) ws.tmp = -1;
         else {
            zh_bitw_t w{out, cap_bits, 0, 0, 0};
            w.put((uint32_t)(nlit - 257), 5);
            w.put((uint32_t)(ndist - 1), 5);
            w.put((uint32_t)(ncl - 4), 4);
            for (int k = 0; k < ncl; k++) w.put(h->len[zh_cl_order(k)], 3);
            zh_cl_write_sink sink{h, &w};
            zh_cl_tokenize(ws.lens, nlit + ndist, best_mask, sink);
            ws.obuf[0] = (uint32_t)w.acc;   // partial dword continues in the token emitter
            ws.tmp = (int32_t)w.nbits;
         }
      }
      zh_sync();
      if (ws.tmp < 0)
         failed = 1;
      else
         bitpos = (uint32_t)ws.tmp;
      zh_sync();
   }

   __threadfence_block();
   zh_sync();
   ZH_STAMP(13);
   uint32_t nbits = 0;
   if (!failed) nbits = zh_emit_tokens_wave(&ws, win, prev, start, end, best, out, cap_bits, bitpos);
   if (nbits > cap_bits) failed = 1;   // outgrew the slot: the stitcher stores the sub-block instead
   ZH_STAMP(14);

   if (lane == 0) {
      zh_subblock_t r;
      r.block = wk.block;
      r.start = wk.start - prev;
      r.size = wk.size;
      r.is_dynamic = is_dynamic;
      r.static_cost = static_cost;
      r.dynamic_cost = dynamic_cost;
      r.failed = failed;
      r.reserved = 0;
      r.nbits = nbits;
      r.bits_off = wk.out_off;
      results[wk.index] = r;
   }
}

// ---- work-list planning: one thread per max-block turns the token boundaries into sub-block work items ----
// (libzultra.c:309-314: nBlockSize = nSplitOffset[k] - (nInStart + prev)). Sub-block k of block b gets the
// payload slot starting at (block's slot base) + (offset of the sub-block in the block) + 64*k, so slots
// never overlap and every slot can hold size+8 bytes.
__global__ void zh_plan_subblocks(const zh_block_t *__restrict__ blocks, uint32_t nblocks, const uint32_t *__restrict__ tok_pos,
                                  uint64_t tok_stride, const uint32_t *__restrict__ ntok, const uint32_t *__restrict__ split_tok,
                                  const uint32_t *__restrict__ split_cnt, const uint32_t *__restrict__ sub_base /* exclusive scan of split_cnt */,
                                  uint64_t slot_stride, zh_work_t *work) {
   const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
   if (b >= nblocks) return;
   const zh_block_t blk = blocks[b];
   const uint32_t *tp = tok_pos + (uint64_t)b * tok_stride;
   const uint32_t *st = split_tok + (uint64_t)b * (ZH_MAX_SPLITS + 1);
   const uint32_t cnt = split_cnt[b];
   const uint32_t nt = ntok[b];
   for (uint32_t k = 0; k < cnt; k++) {
      const uint32_t t0 = st[k], t1 = st[k + 1];
      const uint32_t p0 = (t0 < nt) ? tp[t0] : blk.prev + blk.n;
      const uint32_t p1 = (t1 < nt) ? tp[t1] : blk.prev + blk.n;
      zh_work_t w;
      w.block = b;
      w.start = p0;
      w.size = p1 - p0;
      w.tok0 = t0;
      w.tok1 = t1;
      const uint32_t rel = p0 - blk.prev;
      w.out_off = (uint64_t)b * slot_stride + ((rel + 3u) & ~3u) + 64u * k;
      w.out_cap = ((p1 - p0) + 8u + 3u) & ~3u;
      w.index = sub_base[b] + k;
      work[sub_base[b] + k] = w;
   }
}
