// zh_split.h — stage 2 of the hot path: greedy tokenisation and the block splitter, one wave per max-block.
//
// Replaces zultra_block_split (reference src/blockdeflate.c:634-813, called at src/libzultra.c:303) and the
// greedy histogram it and the cost evaluation lean on (blockdeflate.c:333-361, :519-527).
//
// Observation that shapes the kernels: every greedy walk the reference makes inside one max-block — the
// splitter's scan at each recursion level, the "left part" histograms, the per-sub-block initial entropy —
// starts on a token boundary of the walk that started at the block start, because recursion ranges begin
// at checkpoints and checkpoints are token ends. So there is exactly ONE greedy token chain per max-block.
//   zh_tokenize_spans  materialises it once (token position + packed symbols), 64 positions per step: the match
//                lengths of a tile sit in one VGPR, the chain is followed on the scalar unit with
//                v_readlane, and token lanes compact their record with a ballot prefix.
//   zh_split     then works on token ranges: histograms are lane-parallel LDS atomics over tokens, the
//                reference's checkpoints fall every 256 tokens, and only the drift test and the
//                Moffat-Katajainen merges are serial.
#pragma once
#include <zh_platform.h>
#include "zh_common.h"
#include "zh_huffman.h"

// token info: literal/length symbol (9 bits) | distance symbol (5 bits) << 9
#define ZH_TOK_SYM(info) ((info) & 511u)
#define ZH_TOK_DSYM(info) (((info) >> 9) & 31u)

// Follow the chain through a 64-position tile. `len` holds, per lane, the step length of that position
// (>=3: match, else literal). `carry` = offset of the first token start relative to the tile (may be >= 64).
// Returns the mask of token starts among the first `limit` positions and updates carry for the next tile.
// (Round 5: a run of literals is one loop iteration, not one per literal — a ballot gives the tile's match positions; standing on a match the step is what it was
// (mark, read its length, jump), standing on a literal everything up to the next match is marked at once. Rounds 1-4 took a dependent readlane-compare-add round per
// token: on inputs that are mostly literals — the first kilobytes of every small file — 64 rounds per tile, 6600 cycles; the forward walks of zh_parse_lanes were
// 29 % of that kernel on configuration 5, tools/lp_profile.py.)
__device__ inline uint64_t zh_chain_mask(uint32_t len, uint32_t &carry, uint32_t limit) {
   const uint64_t below_limit = limit >= 64u ? ~0ull : ((1ull << limit) - 1ull);
   const uint64_t matches = zh_ballot(len >= ZH_MIN_MATCH) & below_limit;
   uint64_t mask = 0;
   uint32_t p = carry;
   while (p < limit) {   // (p < 64)
      if ((matches >> p) & 1ull) {
         mask |= 1ull << p;
         p += zh_readlane(len, (int)p);
         continue;
      }
      const uint64_t ahead = matches & (~0ull << p);
      const uint32_t q = ahead ? (uint32_t)zh_ctz64(ahead) : limit;   // the next match (or the limit): p .. q - 1 are literals
      mask |= (q >= 64u ? ~0ull : ((1ull << q) - 1ull)) & (~0ull << p);
      p = q;
   }
   carry = p - limit;   // only meaningful when limit == 64 (a full tile) or at the very end
   return mask;
}

// first barrier at or after block-relative position r, limited to rend (returns rend if there is none before it)
__device__ inline uint32_t zh_first_barrier(const uint64_t *bar, uint32_t r, uint32_t rend) {
   if (r >= rend) return rend;
   uint32_t w = r >> 6;
   const uint32_t wend = (rend + 63) >> 6;
   uint64_t m = bar[w] & (~0ull << (r & 63));
   while (!m) {
      if (++w >= wend) return rend;
      m = bar[w];
   }
   const uint32_t q = w * 64 + (uint32_t)zh_ctz64(m);
   return q < rend ? q : rend;
}

// ---------------------------------------------------------------------------------------------------------
// Barriers and the token chain, in chunks of ZH_TOK_CHUNK positions (a max-block of the reference's default size, 1 MiB,
// would otherwise be one wave following one chain: 40 ms per 100 MB).
//   zh_barriers        one wave per chunk: bit r of the barrier bitmap (zh_parse.h) is set when every match that starts at a
//                      block position < r ends at or before r, i.e. when the running maximum of (r' + max(longest length at
//                      r', 1)) over r' < r is <= r. A chunk computes it from its own positions and reports its maximum;
//   zh_barriers_fix    what earlier chunks reach into a chunk (at most 257 positions) clears the bits below it;
//   zh_tokenize_spans  every parse has a token boundary at a barrier, so the chain restarts exactly at the first barrier at
//                      or after each chunk start: one wave per span, tokens staged at the span's position offset;
//   zh_tokens_compact  one workgroup per max-block moves the spans' tokens together (in place: they only move down).
#ifndef ZH_TOK_CHUNK
#define ZH_TOK_CHUNK 16384u   // a multiple of 64 (the emulator build uses 1024, so that its small test windows span several chunks)
#endif
// A batch of a few max-blocks (one call on one block: what counts is the longest chain of dependent steps, not the number of waves) takes chunks of an eighth of
// that — `chunk` is a kernel argument, cpb the chunks per max-block that go with it (round 5: one 39 KB block, zh_barriers 0.09 -> 0.03 ms, zh_tokenize_spans 0.18 -> 0.05).
#define ZH_TOK_CHUNK_SMALL (ZH_TOK_CHUNK / 8u >= 512u ? ZH_TOK_CHUNK / 8u : 512u)
static_assert(ZH_TOK_CHUNK % 64u == 0 && ZH_TOK_CHUNK_SMALL % 64u == 0, "chunks are whole words of the barrier bitmap");

__global__ void __launch_bounds__(64)
zh_barriers(const zh_block_t *__restrict__ blocks, const uint32_t *__restrict__ longest, uint64_t longest_stride, uint64_t *bars, uint64_t bar_stride,
            uint32_t *chunkmax, uint32_t cpb, uint32_t chunk, uint32_t *slot0, uint64_t slot0_stride /* slot 0 of every row once more, 4 bytes per block position and
            dense: this kernel reads the rows in position order — a word out of every 16 bytes — and is the one place where that copy can be
            written coalesced (the frontier owns entries of the order: its stores to such a plane scattered); zh_tokenize_spans and
            zh_list_huge read it instead of the rows. It lives in the parse-entry array, which nothing else touches before the first parse pass */) {
   const uint32_t b = blockIdx.x / cpb, c = blockIdx.x - b * cpb;
   const zh_block_t blk = blocks[b];
   const uint32_t n = blk.n, lo = c * chunk;
   if (lo >= n) return;
   const uint32_t hi = min(n, lo + chunk);
   const uint32_t *rows = longest + (uint64_t)b * longest_stride;   // per position r its longest match: slot 0 of its row, rows[4 r] (matchfinder.c:221; first plane, zh_common.h)
   uint64_t *bar = bars + (uint64_t)b * bar_stride;
   const uint32_t lane = zh_lane();
   uint32_t reach_before = 0;   // running maximum over the chunk's earlier tiles
   uint32_t pm[4];              // the rows of four tiles are in flight while one is processed
#pragma unroll
   for (uint32_t u = 0; u < 4; u++) {
      const uint32_t r = lo + u * 64 + lane;
      pm[u] = r < hi ? rows[4u * r] : 0u;
   }
   for (uint32_t base4 = lo; base4 < hi; base4 += 256) {
#pragma unroll
      for (uint32_t u = 0; u < 4; u++) {
         const uint32_t base = base4 + u * 64;
         if (base >= hi) break;
         const uint32_t r = base + lane;
         const uint32_t len = pm[u] & 0xffffu;
         if (r < hi) slot0[(uint64_t)b * slot0_stride + r] = pm[u];
         pm[u] = r + 256 < hi ? rows[4u * (r + 256)] : 0u;
         const uint32_t incl = zh_wave_incl_max(r < hi ? r + max(len, 1u) : 0u);
         const uint32_t up = zh_wave_shr1(incl, 0u);   // (lane 0's is not used)
         const uint32_t excl = lane ? max(reach_before, up) : reach_before;
         const uint64_t bm = zh_ballot(r < hi && excl <= r);
         if (lane == 0) bar[base >> 6] = bm;
         reach_before = max(reach_before, zh_readlane(incl, 63));
      }
   }
   if (lane == 0) chunkmax[blockIdx.x] = reach_before;
}

__global__ void zh_barriers_fix(const zh_block_t *__restrict__ blocks, uint32_t nblocks, uint64_t *bars, uint64_t bar_stride, const uint32_t *__restrict__ chunkmax,
                                uint32_t cpb, uint32_t chunk) {
   const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
   if (b >= nblocks) return;
   const uint32_t n = blocks[b].n;
   uint64_t *bar = bars + (uint64_t)b * bar_stride;
   uint32_t reach = 0;
   for (uint32_t c = 1; c * chunk < n; c++) {
      reach = max(reach, chunkmax[b * cpb + c - 1]);
      const uint32_t lo = c * chunk;   // positions lo .. reach-1 have a match crossing them (reach <= lo + 257)
      for (uint32_t w = lo >> 6; w * 64 < min(reach, n); w++) {
         const uint32_t keep_from = reach - w * 64;   // bits below it are cleared
         bar[w] &= keep_from >= 64 ? 0ull : (~0ull << keep_from);
      }
   }
}

__global__ void __launch_bounds__(64)
zh_tokenize_spans(const uint8_t *__restrict__ data, const zh_block_t *__restrict__ blocks, const uint32_t *__restrict__ longest, uint64_t longest_stride,
                  uint32_t *tok_pos, uint16_t *tok_info, uint64_t tok_stride, const uint64_t *__restrict__ bars, uint64_t bar_stride, uint32_t *spanstart,
                  uint32_t *spancnt, uint32_t cpb, uint32_t chunk) {
   const uint32_t b = blockIdx.x / cpb, c = blockIdx.x - b * cpb;
   const zh_block_t blk = blocks[b];
   const uint32_t n = blk.n, lo = c * chunk;
   const uint32_t lane = zh_lane();
   const uint64_t *bar = bars + (uint64_t)b * bar_stride;
   // the span of this chunk: from the first barrier at or after its start to the first barrier at or after its end
   const uint32_t s0 = c == 0 ? 0u : zh_first_barrier(bar, lo, n);
   const uint32_t s1 = (lo >= n || lo + chunk >= n) ? n : zh_first_barrier(bar, lo + chunk, n);
   if (lo >= n || s0 >= s1) {
      if (lane == 0) {
         spanstart[blockIdx.x] = min(s0, n);
         spancnt[blockIdx.x] = 0;
      }
      return;
   }
   const uint8_t *win = data + blk.win_off;
   const uint32_t *rows = longest + (uint64_t)b * longest_stride;   // (the dense copy of slot 0 written by zh_barriers: one word per block position)
   uint32_t *tp = tok_pos + (uint64_t)b * tok_stride + s0;    // staged at the span's position offset (tokens <= positions)
   uint16_t *ti = tok_info + (uint64_t)b * tok_stride + s0;
   uint32_t ntok = 0, carry = 0;
   uint32_t pm[4], pb[4];   // the rows of four tiles are in flight while one is processed
#pragma unroll
   for (uint32_t u = 0; u < 4; u++) {
      const uint32_t r = s0 + u * 64 + lane;
      pm[u] = 0;
      pb[u] = 0;
      if (r < s1) {
         pm[u] = rows[r];
         pb[u] = win[blk.prev + r];
      }
   }
   for (uint32_t base4 = s0; base4 < s1; base4 += 256) {
#pragma unroll
      for (uint32_t u = 0; u < 4; u++) {
         const uint32_t base = base4 + u * 64;
         if (base >= s1) break;
         const uint32_t limit = min(64u, s1 - base);
         const uint32_t r = base + lane;
         const uint32_t m0 = pm[u], byte = pb[u];
         pm[u] = 0;
         pb[u] = 0;
         if (r + 256 < s1) {
            pm[u] = rows[r + 256];
            pb[u] = win[blk.prev + r + 256];
         }
         const uint32_t len = m0 & 0xffffu;
         const uint64_t mask = zh_chain_mask(len, carry, limit);
         if ((mask >> lane) & 1ull) {
            const uint32_t idx = ntok + (uint32_t)zh_popc64(mask & ((1ull << lane) - 1));
            const uint32_t info = (len >= ZH_MIN_MATCH) ? ((257u + (uint32_t)zh_len_idx(len)) | ((uint32_t)zh_dist_sym(m0 >> 16) << 9)) : byte;
            tp[idx] = blk.prev + r;
            ti[idx] = (uint16_t)info;
         }
         ntok += (uint32_t)zh_popc64(mask);
      }
   }
   if (lane == 0) {
      spanstart[blockIdx.x] = s0;
      spancnt[blockIdx.x] = ntok;
   }
}

#define ZH_COMPACT_THREADS 256
__global__ void __launch_bounds__(ZH_COMPACT_THREADS)
zh_tokens_compact(const zh_block_t *__restrict__ blocks, uint32_t *tok_pos, uint16_t *tok_info, uint64_t tok_stride, const uint32_t *__restrict__ spanstart,
                  const uint32_t *__restrict__ spancnt, uint32_t cpb, uint32_t chunk, uint32_t *ntok_out) {
   const uint32_t b = blockIdx.x, tid = threadIdx.x;
   const uint32_t n = blocks[b].n;
   uint32_t *tp = tok_pos + (uint64_t)b * tok_stride;
   uint16_t *ti = tok_info + (uint64_t)b * tok_stride;
   uint32_t total = 0;
   for (uint32_t c = 0; c * chunk < n; c++) {
      const uint32_t src = spanstart[b * cpb + c], m = spancnt[b * cpb + c];
      if (m && src != total) {
         // moving down in place: a step writes below what it has just read, and never above what later steps read
         for (uint32_t off = 0; off < m; off += 8 * ZH_COMPACT_THREADS) {   // eight elements per thread in flight
            uint32_t v[8], w[8];
#pragma unroll
            for (uint32_t u = 0; u < 8; u++) {
               const uint32_t k = off + u * ZH_COMPACT_THREADS + tid;
               v[u] = 0;
               w[u] = 0;
               if (k < m) {
                  v[u] = tp[src + k];
                  w[u] = ti[src + k];
               }
            }
            __syncthreads();
#pragma unroll
            for (uint32_t u = 0; u < 8; u++) {
               const uint32_t k = off + u * ZH_COMPACT_THREADS + tid;
               if (k < m) {
                  tp[total + k] = v[u];
                  ti[total + k] = (uint16_t)w[u];
               }
            }
            __syncthreads();
         }
      }
      total += m;
   }
   if (tid == 0) ntok_out[b] = total;
}

// ---------------------------------------------------------------------------------------------------------
// zh_split: one workgroup of ZH_SPLIT_WAVES waves per max-block
// ---------------------------------------------------------------------------------------------------------
// The reference's search (blockdeflate.c:659-773) walks the checkpoints of a range one after another, but nothing it
// decides at a checkpoint feeds the next one except cumulative histograms: whether checkpoint j triggers an evaluation
// depends on the 18-bin statistics before and inside its interval, and an evaluation prices "tokens before the previous
// checkpoint" against "the rest". So: the interval statistics are gathered by all waves, every wave then replays the cheap
// trigger scan, the triggered evaluations (four Huffman length builds and two table costs each — what the kernel's time
// goes into) are pulled by the waves from a counter, and the reference's selection rule runs over the gains in order.
// The histogram and the price of the whole range are needed only once a checkpoint triggers; the price is then computed by
// one wave next to the evaluations (measured before: all eight waves pricing it took 33-73 % of the kernel's wave-cycles).
#ifndef ZH_SPLIT_WAVES
#define ZH_SPLIT_WAVES 8
#endif
#define ZH_SPLIT_THREADS (64 * ZH_SPLIT_WAVES)
#define ZH_SPLIT_MAXCP 256   // checkpoints per chunk (a 64 KiB max-block has at most 256)

struct zh_split_wave_ws_t {   // private to one wave
   int32_t left_lit[ZH_NLIT], left_dist[ZH_NDIST];
   int32_t cur_lit[ZH_NLIT], cur_dist[ZH_NDIST];
   uint8_t lit_len[ZH_NLIT], dist_len[ZH_NDIST];
   uint8_t lens[ZH_NLIT + ZH_NDIST];
   zh_huff_scratch_t sc;
   zh_cl_t cl;
   int32_t tmp;
};
struct zh_split_shared_t {
   int32_t tot_lit[ZH_NLIT], tot_dist[ZH_NDIST];
   uint32_t fresh[ZH_SPLIT_MAXCP][9];   // 18 bin counts of 16 bits per interval (an interval has at most 512 tokens)
   int32_t gain[ZH_SPLIT_MAXCP];   // per triggered checkpoint: price of the two halves
   int32_t total_cost;             // price of the whole range (wave 0)
   uint32_t next_eval;             // the next triggered checkpoint to hand to a wave
};

// blockdeflate.c:577-618 for the histogram in (lit, dist): unlimited lengths, body bits, header bits, +3.
// All lanes of one wave call; returns the same value in every lane.
__device__ __forceinline__ int zh_dynamic_cost_wave(const int32_t *lit, const int32_t *dist, uint8_t *lit_len, uint8_t *dist_len,
                                           uint8_t *lens, zh_cl_t *cl, int32_t *tmp, zh_huff_scratch_t *sc,
                                           bool compute_lengths) {
   const int lane = (int)zh_lane();
   if (compute_lengths) {
      zh_huff_lengths_wave(lit, lit_len, ZH_NLIT, sc);
      zh_huff_lengths_wave(dist, dist_len, ZH_NDIST, sc);
   }
   uint32_t body = 0;
   for (int s = lane; s < 257 + 29; s += 64) {
      int xb = (s >= 257) ? zh_lenidx_xbits(s - 257) : 0;
      body += (uint32_t)(lit[s] * ((int)lit_len[s] + xb));
   }
   if (lane < ZH_NDIST) body += (uint32_t)(dist[lane] * ((int)dist_len[lane] + zh_dist_xbits(lane)));
   body = zh_wave_sum(body);

   const int nlit = zh_defined_count(lit_len, ZH_NLIT, 257);
   const int ndist = zh_defined_count(dist_len, ZH_NDIST, 1);
   for (int s = lane; s < nlit; s += 64) lens[s] = lit_len[s];
   if (lane < ndist) lens[nlit + lane] = dist_len[lane];
   zh_wave_sync();
   const int r = (int)body + zh_table_cost_wave(lens, nlit + ndist, cl, sc->keys) + 3;   // the sort scratch (keys, sorted: 576 words) is free here
   return r;
}

// histogram of the greedy tokens [t0, t1) added into (lit, dist). All lanes of one wave call.
__device__ inline void zh_token_histogram_wave(const uint16_t *ti, uint32_t t0, uint32_t t1, int32_t *lit, int32_t *dist) {
   for (uint32_t t = t0 + zh_lane(); t < t1; t += 64) {
      uint32_t info = ti[t];
      uint32_t s = ZH_TOK_SYM(info);
      atomicAdd(&lit[s], 1);
      if (s > 256) atomicAdd(&dist[ZH_TOK_DSYM(info)], 1);
   }
   zh_wave_sync();
}

// Search the best split of token range [t0, t1) (blockdeflate.c:659-773). Returns the token index of the
// split boundary, or 0xFFFFFFFF. All threads of the workgroup call; uniform result.
template <int WAVES>
__device__ inline uint32_t zh_split_search_wg(zh_split_shared_t *sh, zh_split_wave_ws_t *ws, const uint32_t *tp, const uint16_t *ti, uint32_t t0,
                                              uint32_t t1, uint32_t start_pos, uint32_t end_pos) {
   const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

   // checkpoints: the first boundary c0 with >= 256 tokens and >= 512 bytes since the range start, then every 256 tokens (:705)
   uint32_t c0 = t0 + 256;
   if (c0 > t1) return 0xFFFFFFFFu;
   while (c0 < t1 && tp[c0] - start_pos < 512) c0++;
   {
      const uint32_t pc = (c0 < t1) ? tp[c0] : end_pos;
      if (pc - start_pos < 512) return 0xFFFFFFFFu;   // ran out of tokens before 512 bytes
   }
   const uint32_t ncp = (t1 - c0) / 256 + 1;

   uint32_t seen = 0;      // lane < 18: tokens of this bin before the current interval (:709-719)
   uint32_t nseen = 0;
   uint32_t best = 0xFFFFFFFFu;
   int best_gain = 0;
   uint32_t left_upto = 0xFFFFFFFFu;   // this wave's left histogram covers tokens [t0, left_upto); all ones = not started
   bool have_total = false;            // histogram and price of the whole range: only needed once a checkpoint triggers

   for (uint32_t j0 = 0; j0 < ncp; j0 += ZH_SPLIT_MAXCP) {
      const uint32_t nj = min((uint32_t)ZH_SPLIT_MAXCP, ncp - j0);
      __syncthreads();   // the previous chunk's statistics have been read
      // 18-bin statistics of every interval of the chunk: interval j = tokens [c(j-1), c(j)), c(-1) = t0, c(j) = c0 + 256 j
      for (uint32_t k = tid; k < nj * 9; k += (64u * WAVES)) sh->fresh[k / 9][k % 9] = 0;
      for (uint32_t k = tid; k < nj; k += (64u * WAVES)) sh->gain[k] = 0;
      if (tid == 0) sh->next_eval = 0;
      __syncthreads();
      for (uint32_t jj = wave; jj < nj; jj += (uint32_t)WAVES) {
         const uint32_t j = j0 + jj;
         const uint32_t lo = j == 0 ? t0 : c0 + 256 * (j - 1), hi = c0 + 256 * j;
         for (uint32_t t = lo + lane; t < hi; t += 64) {
            const uint32_t s = ZH_TOK_SYM(ti[t]);
            const uint32_t bin = (s < 256) ? (((s >> 4) & 0xc) | (s & 3)) : (s >= 263 ? 17u : 16u);   // :689-699 (len >= 9 <=> symbol >= 263)
            atomicAdd(&sh->fresh[jj][bin >> 1], 1u << (16u * (bin & 1u)));
         }
      }
      __syncthreads();
      // which checkpoints trigger an evaluation (every wave replays this; it is a few instructions per checkpoint)
      uint64_t trig[ZH_SPLIT_MAXCP / 64] = {0, 0, 0, 0};
      for (uint32_t jj = 0; jj < nj; jj++) {
         const uint32_t j = j0 + jj;
         const uint32_t nfresh = j == 0 ? c0 - t0 : 256u;
         const uint32_t fr = lane < 18 ? (sh->fresh[jj][lane >> 1] >> (16u * (lane & 1u))) & 0xffffu : 0u;
         if (nseen) {
            uint32_t d = 0;
            if (lane < 18) {
               const uint32_t expected = seen * nfresh;   // uint32 wrap is part of the behaviour (:710-711)
               const uint32_t actual = fr * nseen;
               d = expected > actual ? expected - actual : actual - expected;
            }
            const uint32_t drift = zh_wave_sum(d);
            if ((drift / nfresh) >= (nseen * 45 / 100)) trig[jj >> 6] |= 1ull << (jj & 63);   // a previous checkpoint exists: nseen != 0
         }
         seen += fr;
         nseen += nfresh;
      }
      if (!(trig[0] | trig[1] | trig[2] | trig[3])) continue;   // nothing to evaluate (the common case on homogeneous data)

      if (!have_total) {
         // histogram of the whole range (all waves), its price (wave 0 alone, while the others start on the evaluations)
         have_total = true;
         for (uint32_t s = tid; s < ZH_NLIT; s += (64u * WAVES)) sh->tot_lit[s] = 0;
         if (tid < ZH_NDIST) sh->tot_dist[tid] = 0;
         __syncthreads();
         for (uint32_t t = t0 + tid; t < t1; t += (64u * WAVES)) {
            const uint32_t info = ti[t];
            const uint32_t s = ZH_TOK_SYM(info);
            atomicAdd(&sh->tot_lit[s], 1);
            if (s > 256) atomicAdd(&sh->tot_dist[ZH_TOK_DSYM(info)], 1);
         }
         __syncthreads();
         if (tid == 0) sh->tot_lit[ZH_EOB] += 1;
         __syncthreads();
         if (wave == 0) {
            const int tc = zh_dynamic_cost_wave(sh->tot_lit, sh->tot_dist, ws->lit_len, ws->dist_len, ws->lens, &ws->cl, &ws->tmp, &ws->sc, true);
            if (lane == 0) sh->total_cost = tc;
         }
      }
      // the triggered evaluations: left = tokens before the previous checkpoint, right = the rest (:732-750). The waves pull
      // HALF evaluations (one side of one checkpoint: two Huffman length builds and a table cost) from a counter — a search
      // triggers only two or three evaluations on average, so this halves its latency — in ascending checkpoint order, so a
      // wave's left histogram grows from one item to the next instead of being recounted from the range start.
      uint32_t scan_jj = 0, scan_cnt = 0;   // triggered checkpoints below scan_jj: scan_cnt
      uint32_t last_k = 0xFFFFFFFFu, last_jj = nj;
      for (;;) {
         uint32_t want = 0;
         if (lane == 0) want = atomicAdd(&sh->next_eval, 1u);
         want = zh_readfirstlane(want);
         const uint32_t side = want & 1u;
         want >>= 1;
         uint32_t jj = nj;
         if (want == last_k)
            jj = last_jj;
         else
            while (scan_jj < nj) {
               const bool set = (trig[scan_jj >> 6] >> (scan_jj & 63)) & 1ull;
               const uint32_t at = scan_jj++;
               if (set && scan_cnt++ == want) {
                  jj = at;
                  break;
               }
            }
         if (jj >= nj) break;
         last_k = want;
         last_jj = jj;
         const uint32_t cp = c0 + 256 * (j0 + jj - 1);
         if (left_upto == 0xFFFFFFFFu) {
            for (uint32_t s = lane; s < ZH_NLIT; s += 64) ws->left_lit[s] = 0;
            if (lane < ZH_NDIST) ws->left_dist[lane] = 0;
            left_upto = t0;
         }
         else if (lane == 0)
            ws->left_lit[ZH_EOB] = 0;   // the end-of-block symbol of the previous evaluation
         zh_wave_sync();
         zh_token_histogram_wave(ti, left_upto, cp, ws->left_lit, ws->left_dist);
         left_upto = cp;
         int cost;
         if (side) {
            for (uint32_t s = lane; s < ZH_NLIT; s += 64) ws->cur_lit[s] = sh->tot_lit[s] - ws->left_lit[s];
            if (lane < ZH_NDIST) ws->cur_dist[lane] = sh->tot_dist[lane] - ws->left_dist[lane];
            zh_wave_sync();
            if (lane == 0) {
               ws->left_lit[ZH_EOB] = 1;   // (reset before the histogram grows again)
               ws->cur_lit[ZH_EOB] = 1;
            }
            zh_wave_sync();
            cost = zh_dynamic_cost_wave(ws->cur_lit, ws->cur_dist, ws->lit_len, ws->dist_len, ws->lens, &ws->cl, &ws->tmp, &ws->sc, true);
         }
         else {
            if (lane == 0) ws->left_lit[ZH_EOB] = 1;
            zh_wave_sync();
            cost = zh_dynamic_cost_wave(ws->left_lit, ws->left_dist, ws->lit_len, ws->dist_len, ws->lens, &ws->cl, &ws->tmp, &ws->sc, true);
         }
         if (lane == 0) atomicAdd(&sh->gain[jj], cost);
      }
      __syncthreads();
      // the reference keeps the first non-negative gain, then only strictly larger ones (:751-757)
      const int total_cost = sh->total_cost;
      for (uint32_t jj = 0; jj < nj; jj++) {
         if (!((trig[jj >> 6] >> (jj & 63)) & 1ull)) continue;
         const int gain = total_cost - sh->gain[jj];
         if (gain >= 0 && (best == 0xFFFFFFFFu || best_gain < gain)) {
            best = c0 + 256 * (j0 + jj - 1);
            best_gain = gain;
         }
      }
   }
   __syncthreads();   // the caller may start the next search: shared state is free
   return best;
}

// Output: split_tok[b*65 + k] token boundaries (k = 0..count, first = 0, last = ntok), split_cnt[b] = number of sub-blocks.
// WAVES = 8 measured best for 64 KiB max-blocks (4: +70 %, 16: +35 %); large max-blocks are few and their searches trigger
// hundreds of evaluations, there 16 waves win
template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES)
zh_split(const zh_block_t *__restrict__ blocks, const uint32_t *__restrict__ tok_pos, const uint16_t *__restrict__ tok_info,
         uint64_t tok_stride, const uint32_t *__restrict__ ntok_in, uint32_t *split_tok, uint32_t *split_cnt) {
   __shared__ zh_split_shared_t sh;
   __shared__ zh_split_wave_ws_t wws[WAVES];
   zh_split_wave_ws_t *ws = &wws[threadIdx.x >> 6];
   const zh_block_t blk = blocks[blockIdx.x];
   const uint32_t *tp = tok_pos + (uint64_t)blockIdx.x * tok_stride;
   const uint16_t *ti = tok_info + (uint64_t)blockIdx.x * tok_stride;
   const uint32_t ntok = ntok_in[blockIdx.x];
   const uint32_t block_end = blk.prev + blk.n;
   uint32_t *out = split_tok + (uint64_t)blockIdx.x * (ZH_MAX_SPLITS + 1);

   // explicit recursion (depth <= 6): frames hold [t0,t1), the chosen boundary and a phase; identical in every thread
   uint32_t f_t0[8], f_t1[8], f_best[8];
   int f_phase[8];
   int sp = 0;
   uint32_t count = 0;   // interior splits recorded so far (max 63, :805)
   f_t0[0] = 0;
   f_t1[0] = ntok;
   f_phase[0] = 0;
   f_best[0] = 0;

   if (threadIdx.x == 0) out[0] = 0;

   while (sp >= 0) {
      const uint32_t t0 = f_t0[sp], t1 = f_t1[sp];
      if (f_phase[sp] == 0) {
         const uint32_t p0 = (t0 < ntok) ? tp[t0] : block_end;
         const uint32_t p1 = (t1 < ntok) ? tp[t1] : block_end;
         if (count >= ZH_MAX_SPLITS - 1 || sp >= 6 || (p1 - p0) < 8192) {   // :643-647
            sp--;
            continue;
         }
         const uint32_t b = zh_split_search_wg<WAVES>(&sh, ws, tp, ti, t0, t1, p0, p1);
         if (b == 0xFFFFFFFFu) {
            sp--;
            continue;
         }
         f_best[sp] = b;
         f_phase[sp] = 1;
         f_t0[sp + 1] = t0;
         f_t1[sp + 1] = b;
         f_phase[sp + 1] = 0;
         sp++;
      }
      else if (f_phase[sp] == 1) {
         if (count < ZH_MAX_SPLITS - 1) {
            count++;
            if (threadIdx.x == 0) out[count] = f_best[sp];
         }
         f_phase[sp] = 2;
         f_t0[sp + 1] = f_best[sp];
         f_t1[sp + 1] = t1;
         f_phase[sp + 1] = 0;
         sp++;
      }
      else
         sp--;
   }
   if (threadIdx.x == 0) {
      out[count + 1] = ntok;
      split_cnt[blockIdx.x] = count + 1;
   }
}
