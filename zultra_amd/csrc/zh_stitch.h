// zh_stitch.h — stream assembly: where every sub-block lands in the deflate stream, and the kernels that put it there.
//
// Replaces the framing half of the reference's per-sub-block loop (src/libzultra.c:327-398): BFINAL/BTYPE bits, the
// compressed-or-stored decision — which depends on the running bit phase because the reference compares whole
// flushed bytes (:345-347) — stored pieces of <= 65535 bytes, and the bit carry across max-blocks (:427-434); plus
// the CRC-32 of src/frame.c:324-354 computed per max-block on the device and combined on the host.
//
//   zh_stitch_plan   (host, serial, ~48 B per sub-block): the phase-dependent decisions -> destination bit offsets.
//   zh_stitch        (device, one workgroup per sub-block): funnel-shifts the phase-0 bit string of a compressed
//                    sub-block to its destination (interior dwords plain stores, the two boundary dwords atomic OR),
//                    or writes the stored pieces from the raw window bytes.
//   zh_crc32_blocks  (device, one workgroup per max-block): slice CRCs by table lookup from LDS, folded with a
//                    "shift by one slice" operator; the host combines the per-block values (GF(2) operator for a whole
//                    max-block, applied once per block).
#pragma once
#include "zh_parse.h"   // (zh_run_is_void)
#include <stdint.h>

#include "zh_common.h"

typedef struct zh_stitch_item_s {
   uint64_t dst_bit;    // bit offset of the sub-block's first header bit in the stream buffer
   uint32_t stored;     // 1: stored fallback
   uint32_t is_final;   // BFINAL of the (last piece of the) sub-block
} zh_stitch_item_t;

// One sub-block's step of the reference's loop (libzultra.c:327-398): `bit` = where its first header bit goes, block_base = the byte the
// max-block's writer started at (the reference's offset restarts at 0 per max-block, pending bits carry). Advances `bit` past the sub-block,
// says whether it is stored; -1 where the reference fails with ZULTRA_ERROR_DST (its per-max-block buffer of 1 + bs + 5*(bs/65535+1) bytes
// overflows, libzultra.c:115,382). Shared by the host planner and the device scan below: one statement of the rule.
ZH_HD int zh_stitch_step(uint64_t *bit_io, uint64_t block_base, uint32_t size, uint64_t nbits, uint32_t failed, uint64_t blockbuf_cap, uint32_t *stored_out) {
   uint64_t bit = *bit_io;
   const uint32_t nacc = (uint32_t)(bit & 7);
   const uint32_t c0 = (nacc + 3) & 7;
   const uint64_t o0 = ((bit >> 3) - block_base) + ((nacc + 3) >> 3);
   if (o0 > blockbuf_cap) return -1;
   const uint64_t body_bytes = ((uint64_t)c0 + nbits) >> 3;
   if (!failed && body_bytes <= size && o0 + body_bytes <= blockbuf_cap) {
      *stored_out = 0;
      bit += 3 + nbits;
   }
   else {
      *stored_out = 1;
      uint32_t rem = size;
      while (rem) {
         const uint32_t piece = rem > 65535 ? 65535 : rem;
         bit = (bit + 3 + 7) & ~7ull;   // header bits, pad to a byte
         if (((bit >> 3) - block_base) + 4 + piece > blockbuf_cap) return -1;
         bit += 32 + 8ull * piece;
         rem -= piece;
      }
   }
   *bit_io = bit;
   return 0;
}
ZH_HD uint64_t zh_stitch_blockbuf_cap(uint32_t max_block_size) { return 1 + (uint64_t)max_block_size + 5 * ((uint64_t)max_block_size / 65535 + 1); }

// Serial planner (host): the reference statement of the rule, used by the host stitcher (zultra_hip_stitch) and by the tests that hold the
// device scan to it. phase = pending bits (0..7) before the first sub-block. Returns 0, or -1 where the reference fails with ZULTRA_ERROR_DST.
// *end_bit = bit offset after the last sub-block (relative to the same origin as dst_bit; origin = start of the byte
// that holds the pending bits).
static inline int zh_stitch_plan(uint32_t phase, const zh_subblock_t *subs, uint32_t nsubs, uint32_t max_block_size, int final_block,
                                 zh_stitch_item_t *items, uint64_t *end_bit) {
   const uint64_t blockbuf_cap = zh_stitch_blockbuf_cap(max_block_size);
   uint64_t bit = phase;
   uint64_t block_base = 0;
   uint32_t cur_block = 0xFFFFFFFFu;
   for (uint32_t k = 0; k < nsubs; k++) {
      const zh_subblock_t *sb = &subs[k];
      if (sb->block != cur_block) {
         cur_block = sb->block;
         block_base = bit >> 3;
      }
      const int last_of_block = (k + 1 == nsubs) || (subs[k + 1].block != sb->block);
      const uint32_t is_final = ((int)sb->block == final_block && last_of_block) ? 1u : 0u;
      if (items) {
         items[k].dst_bit = bit;
         items[k].is_final = is_final;
      }
      uint32_t stored = 0;
      if (zh_stitch_step(&bit, block_base, sb->size, sb->nbits, sb->failed, blockbuf_cap, &stored) != 0) return -1;
      if (items) items[k].stored = stored;
   }
   *end_bit = bit;
   return 0;
}

#if defined(__HIPCC__) || defined(ZH_EMU)
#include <zh_platform.h>

#define ZH_STITCH_THREADS 256

// what zh_stitch_scan (below) reports about a batch
struct zh_scan_out_t {
   uint64_t end_bit;          // after the last sub-block, from the origin of dst_bit (the byte that holds the pending bits)
   uint32_t failed;           // != 0: the reference would fail with ZULTRA_ERROR_DST
   uint32_t nsubs;
   uint64_t table_end[8];     // end_bit for each of the eight start phases of the batch (start bit = the phase)
   uint32_t table_failed;     // bit p: start phase p overflows a max-block buffer
   uint32_t pad;
};


__device__ __forceinline__ void zh_or_bits(uint32_t *out, uint64_t bit, uint32_t value, uint32_t nbits) {
   // nbits <= 16: at most two dwords
   const uint64_t v = (uint64_t)value << (bit & 31);
   atomicOr(&out[bit >> 5], (uint32_t)v);
   if (((bit & 31) + nbits) > 32) atomicOr(&out[(bit >> 5) + 1], (uint32_t)(v >> 32));
}

// out must be zero-filled. One workgroup per sub-block. Nothing is written when the scan found that the reference would fail, or that the stream
// would not fit the buffer (the host reports either after the one synchronisation of a stitch).
__device__ __forceinline__ void zh_stitch_one(uint32_t s, const zh_subblock_t *__restrict__ subs, const zh_stitch_item_t *__restrict__ items, const zh_block_t *__restrict__ blocks,
                                              const uint8_t *__restrict__ data, const uint8_t *__restrict__ payload, uint32_t *out) {
   const zh_subblock_t sb = subs[s];
   const zh_stitch_item_t it = items[s];
   const uint32_t tid = threadIdx.x;

   if (!it.stored) {
      if (tid == 0) zh_or_bits(out, it.dst_bit, it.is_final | ((1u + sb.is_dynamic) << 1), 3);   // BFINAL, BTYPE (:329-332)
      const uint32_t *src = (const uint32_t *)(payload + sb.bits_off);
      const uint64_t d0 = it.dst_bit + 3;
      const uint64_t nbits = sb.nbits;
      if (nbits == 0) return;
      const uint64_t nsrc = (nbits + 31) >> 5;                 // source dwords that hold valid bits (zero padded)
      const uint64_t w0 = d0 >> 5, w1 = (d0 + nbits - 1) >> 5;  // first / last destination dword
      const uint32_t r = (uint32_t)((32 - (d0 & 31)) & 31);     // source bit offset of destination dword w0+1, mod 32
      for (uint64_t w = w0 + tid; w <= w1; w += ZH_STITCH_THREADS) {
         // destination dword w holds source bits [32w - d0, 32w - d0 + 32)
         const int64_t sbit = (int64_t)(w << 5) - (int64_t)d0;
         uint32_t v;
         if (sbit < 0) {
            v = src[0] << (uint32_t)(-sbit);                    // first dword: low source bits move up
         }
         else {
            const uint64_t j = (uint64_t)sbit >> 5;
            const uint32_t sh = (uint32_t)sbit & 31;
            const uint32_t lo = j < nsrc ? src[j] : 0;
            const uint32_t hi = (j + 1) < nsrc ? src[j + 1] : 0;
            v = sh ? ((lo >> sh) | (hi << (32 - sh))) : lo;
         }
         if (w == w0 || w == w1)
            atomicOr(&out[w], v);    // shared with the neighbouring sub-blocks
         else
            out[w] = v;
      }
      (void)r;
   }
   else {
      // stored pieces (libzultra.c:350-397)
      const zh_block_t blk = blocks[sb.block];
      const uint8_t *raw = data + blk.win_off + blk.prev + sb.start;
      uint8_t *out8 = (uint8_t *)out;
      uint64_t bit = it.dst_bit;
      uint32_t rem = sb.size;
      while (rem) {
         const uint32_t piece = rem > 65535 ? 65535 : rem;
         const uint32_t fin = (rem > 65535) ? 0 : it.is_final;
         if (tid == 0) zh_or_bits(out, bit, fin, 3);            // BFINAL + BTYPE 00
         const uint64_t byte0 = (bit + 3 + 7) >> 3;             // LEN starts on the next byte boundary
         if (tid == 0) {
            // the four LEN/NLEN bytes may straddle dwords shared with neighbours only through byte0's dword: use OR
            const uint32_t hdr[4] = {piece & 0xffu, piece >> 8, (piece & 0xffu) ^ 0xffu, (piece >> 8) ^ 0xffu};
            for (int q = 0; q < 4; q++) atomicOr(&out[(byte0 + q) >> 2], hdr[q] << (8 * ((byte0 + q) & 3)));
         }
         const uint64_t body = byte0 + 4;
         // raw bytes: boundary dwords by OR, interior bytes directly
         for (uint32_t k = tid; k < piece; k += ZH_STITCH_THREADS) {
            const uint64_t o = body + k;
            if ((o >> 2) == (body >> 2) || (o >> 2) == ((body + piece - 1) >> 2))
               atomicOr(&out[o >> 2], (uint32_t)raw[k] << (8 * (o & 3)));
            else
               out8[o] = raw[k];
         }
         raw += piece;
         rem -= piece;
         bit = (body + piece) << 3;
      }
   }
}

// One workgroup per sub-block, striding: the grid may be sized before the host knows the count (a stitch enqueued with its batch, zh_device.hip) —
// the scan leaves it in scan->nsubs.
__global__ void __launch_bounds__(ZH_STITCH_THREADS)
zh_stitch(const zh_subblock_t *__restrict__ subs, const zh_stitch_item_t *__restrict__ items, const zh_block_t *__restrict__ blocks,
          const uint8_t *__restrict__ data, const uint8_t *__restrict__ payload, uint32_t *out, const zh_scan_out_t *__restrict__ scan, uint64_t stream_cap) {
   if (scan->failed || ((scan->end_bit + 7) >> 3) + 8 > stream_cap) return;
   const uint32_t nsubs = scan->nsubs;
   for (uint32_t s = blockIdx.x; s < nsubs; s += gridDim.x) zh_stitch_one(s, subs, items, blocks, data, payload, out);
}

// ---- sub-block descriptors of a batch in stream order ---------------------------------------------------------------------------------
// Every run of a batch (zh_device.hip) writes its descriptors at its worst-case offset, with sub-block, max-block and payload coordinates of
// its own; how many it wrote is in its counters. One small kernel behind the last run lays them end to end in batch coordinates — what the
// host summed up after reading the counts back in rounds 1-4 — and leaves the total.
struct zh_runs_t {
   uint32_t nruns;
   uint32_t b0[8];         // first max-block of run k
   uint32_t cnt_stride;    // words between the runs' counter blocks
   uint32_t nsubs_field;   // index of the sub-block count inside a counter block (ZH_CNT_NSUBS)
   uint64_t max_subs;      // descriptors of run k start at results[b0[k] * max_subs]
   uint64_t slot_stride;   // payload bytes per max-block
};
#define ZH_COMPACT_RESULTS_THREADS 256
__global__ void __launch_bounds__(ZH_COMPACT_RESULTS_THREADS)
zh_compact_results(zh_runs_t R, const zh_subblock_t *__restrict__ results, const uint32_t *__restrict__ counters, zh_subblock_t *out, uint32_t *nsubs_out /* [0] total, [1 + k] of run k */) {
   // a void run (zh_parse.h, zh_run_is_void: enqueued without chain kernels or overflow forms it turned out to need) left no descriptors: the total says so — a stitch that
   // went out with the batch then writes nothing (zh_stitch_scan), and the host, which reads the same counters, runs the batch again
   for (uint32_t k = 0; k < R.nruns; k++) {
      if (zh_run_is_void(counters + (size_t)k * R.cnt_stride)) {
         if (blockIdx.x == 0 && threadIdx.x == 0) nsubs_out[0] = 0xFFFFFFFFu;
         return;
      }
   }
   uint32_t base = 0;
   for (uint32_t k = 0; k < R.nruns; k++) {
      const uint32_t ns = counters[(size_t)k * R.cnt_stride + R.nsubs_field];
      const zh_subblock_t *src = results + (uint64_t)R.b0[k] * R.max_subs;
      for (uint32_t i = blockIdx.x * ZH_COMPACT_RESULTS_THREADS + threadIdx.x; i < ns; i += gridDim.x * ZH_COMPACT_RESULTS_THREADS) {
         zh_subblock_t r = src[i];
         r.block += R.b0[k];
         r.bits_off += (uint64_t)R.b0[k] * R.slot_stride;
         out[base + i] = r;
      }
      if (blockIdx.x == 0 && threadIdx.x == 0) nsubs_out[1 + k] = ns;
      base += ns;
   }
   if (blockIdx.x == 0 && threadIdx.x == 0) nsubs_out[0] = base;
}

// ---- zh_stitch_scan: the phase-dependent decisions on the device (libzultra.c:327-398, 414-436) ---------------------------------------------
// Whether a sub-block is stored, and where its bits go, depends on the bit phase the stream has reached: the reference compares whole flushed
// bytes (:345-347). Across max-blocks the only state is that phase (the writer's offset restarts per max-block), so a run of max-blocks is a map
// phase -> bits it adds, a table of eight entries, and tables compose: the transfer-table scan SURVEY.md 8(f1) asks for. One workgroup:
//   0  the first sub-block of every max-block (the descriptors are in stream order, every max-block has at least one);
//   1  thread t walks its chunk of consecutive max-blocks once for each of the eight start phases -> table T[t][p] (32-bit: a chunk is far below 512 MB);
//   2  the tables are walked in two levels (the waves' 64 chunks for each start phase, then the waves with the true phase, then the chunks of
//      every wave with its true start): every chunk's true start bit. For the eight possible start phases of the BATCH the same walk gives the
//      batch's own table — what a rank sends its neighbours when a stream is cut over several devices (zultra_amd/sharded.py);
//   3  thread t walks its chunk again from its true start bit and writes the items: destination bit, stored or not, BFINAL.
// files != 0: every max-block is a stream of its own (files mode): it starts on a byte boundary at phase 0, its last sub-block carries BFINAL;
// file_off[b] = its first byte, file_off[nblocks] = the end.
// Rounds 1-4 planned this serially on the host (zh_stitch_plan above, kept as the statement of the rule the tests hold this kernel to), between a
// read-back of the descriptors and an upload of the items.
#define ZH_SCAN_THREADS 1024
__global__ void __launch_bounds__(ZH_SCAN_THREADS)
zh_stitch_scan(const zh_subblock_t *__restrict__ subs, const uint32_t *__restrict__ nsubs_p, uint32_t nblocks, uint32_t phase, uint32_t max_block_size, int final_block, int files,
               uint32_t *blk_start /* scratch, nblocks + 1 */, zh_stitch_item_t *items, uint64_t *file_off /* files: nblocks + 1 */, zh_scan_out_t *out) {
   __shared__ uint32_t T[ZH_SCAN_THREADS][8];      // per chunk and start phase: bits added
   __shared__ uint32_t Tfail[ZH_SCAN_THREADS];     // ... bit p: the walk from phase p overflows
   __shared__ uint64_t Wv[ZH_SCAN_THREADS / 64][8];                                // the same per wave of chunks
   __shared__ uint32_t Wfail[ZH_SCAN_THREADS / 64];
   __shared__ uint64_t wave_start[ZH_SCAN_THREADS / 64 + 1];
   __shared__ uint64_t chunk_start[ZH_SCAN_THREADS];
   __shared__ uint32_t s_failed, s_table_failed;
   const uint32_t tid = threadIdx.x;
   const uint32_t nsubs = *nsubs_p;
   if (nsubs == 0xFFFFFFFFu) {   // (a void run in the batch, zh_compact_results: nothing to assemble — zh_stitch leaves at `failed`)
      if (tid == 0) {
         out->end_bit = 0;
         out->failed = 1;
         out->nsubs = nsubs;
         out->table_failed = 0xffu;
      }
      return;
   }
   const uint64_t cap = zh_stitch_blockbuf_cap(max_block_size);
   if (tid == 0) s_failed = s_table_failed = 0;
   if (tid < ZH_SCAN_THREADS / 64) Wfail[tid] = 0;
   // ---- 0: where every max-block's sub-blocks start
   for (uint32_t k = tid; k < nsubs; k += ZH_SCAN_THREADS)
      if (k == 0 || subs[k].block != subs[k - 1].block) blk_start[subs[k].block] = k;
   if (tid == 0) blk_start[nblocks] = nsubs;
   __threadfence_block();
   __syncthreads();
   const uint32_t per = (nblocks + ZH_SCAN_THREADS - 1) / ZH_SCAN_THREADS;
   const uint32_t B0 = min(nblocks, tid * per), B1 = min(nblocks, B0 + per);
   // ---- 1: the chunk's table
   {
      uint64_t bit[8], base[8];
      uint32_t fail = 0;
#pragma unroll
      for (uint32_t p = 0; p < 8; p++) bit[p] = p;
      for (uint32_t b = B0; b < B1; b++) {
         const uint32_t k0 = blk_start[b], k1 = blk_start[b + 1];
#pragma unroll
         for (uint32_t p = 0; p < 8; p++) {
            if (files) bit[p] = (bit[p] + 7) & ~7ull;
            base[p] = bit[p] >> 3;
         }
         for (uint32_t k = k0; k < k1; k++) {
            const uint32_t size = subs[k].size, failed = subs[k].failed;
            const uint64_t nbits = subs[k].nbits;
#pragma unroll
            for (uint32_t p = 0; p < 8; p++) {
               uint32_t stored;
               if (zh_stitch_step(&bit[p], base[p], size, nbits, failed, cap, &stored) != 0) fail |= 1u << p;
            }
         }
      }
#pragma unroll
      for (uint32_t p = 0; p < 8; p++) T[tid][p] = (uint32_t)(bit[p] - p);
      Tfail[tid] = fail;
   }
   __syncthreads();
   // ---- 2a: per wave of chunks and start phase
   if (tid < (ZH_SCAN_THREADS / 64) * 8) {
      const uint32_t w = tid >> 3, p = tid & 7;
      uint64_t bit = p;
      uint32_t fail = 0;
      for (uint32_t c = w * 64; c < w * 64 + 64; c++) {
         const uint32_t q = files ? 0u : (uint32_t)(bit & 7);
         if (files) bit = (bit + 7) & ~7ull;
         fail |= (Tfail[c] >> q) & 1u;
         bit += T[c][q];
      }
      Wv[w][p] = bit - p;
      if (fail) atomicOr(&Wfail[w], 1u << p);
   }
   __syncthreads();
   // ---- 2b: the waves with the true phase; and the batch's own table
   if (tid == 0) {
      uint64_t bit = phase;
      for (uint32_t w = 0; w < ZH_SCAN_THREADS / 64; w++) {
         wave_start[w] = bit;
         const uint32_t q = files ? 0u : (uint32_t)(bit & 7);
         if (files) bit = (bit + 7) & ~7ull;
         bit += Wv[w][q];
      }
      wave_start[ZH_SCAN_THREADS / 64] = bit;
   }
   if (tid >= 64 && tid < 72) {
      const uint32_t p = tid - 64;
      uint64_t bit = p;
      uint32_t fail = 0;
      for (uint32_t w = 0; w < ZH_SCAN_THREADS / 64; w++) {
         const uint32_t q = files ? 0u : (uint32_t)(bit & 7);
         if (files) bit = (bit + 7) & ~7ull;
         fail |= (Wfail[w] >> q) & 1u;
         bit += Wv[w][q];
      }
      out->table_end[p] = bit;
      if (fail) atomicOr(&s_table_failed, 1u << p);
   }
   __syncthreads();
   // ---- 2c: the chunks of every wave from the wave's true start
   if ((tid & 63u) == 0) {
      const uint32_t w = tid >> 6;
      uint64_t bit = wave_start[w];
      for (uint32_t c = w * 64; c < w * 64 + 64; c++) {
         chunk_start[c] = bit;
         const uint32_t q = files ? 0u : (uint32_t)(bit & 7);
         if (files) bit = (bit + 7) & ~7ull;
         bit += T[c][q];
      }
   }
   __syncthreads();
   // ---- 3: the items
   {
      uint64_t bit = chunk_start[tid];
      uint32_t fail = 0;
      for (uint32_t b = B0; b < B1; b++) {
         const uint32_t k0 = blk_start[b], k1 = blk_start[b + 1];
         if (files) {
            bit = (bit + 7) & ~7ull;
            file_off[b] = bit >> 3;
         }
         const uint64_t base = bit >> 3;
         for (uint32_t k = k0; k < k1; k++) {
            zh_stitch_item_t it;
            it.dst_bit = bit;
            it.is_final = (k + 1 == k1 && (files || (int)b == final_block)) ? 1u : 0u;
            it.stored = 0;
            if (zh_stitch_step(&bit, base, subs[k].size, subs[k].nbits, subs[k].failed, cap, &it.stored) != 0) fail = 1;
            items[k] = it;
         }
      }
      if (fail) atomicOr(&s_failed, 1u);
   }
   __syncthreads();
   if (tid == 0) {
      uint64_t end = wave_start[ZH_SCAN_THREADS / 64];
      if (files) {
         end = (end + 7) & ~7ull;
         file_off[nblocks] = end >> 3;
      }
      out->end_bit = end;
      out->failed = s_failed;
      out->table_failed = s_table_failed;
      out->nsubs = nsubs;
   }
}

// ---- CRC-32 per max-block ---------------------------------------------------------------------------------------
#define ZH_CRC_THREADS 256
#define ZH_CRC_SLICE 256   // bytes per thread-slice

// tables: [0..255] byte table of the reflected polynomial 0xEDB88320; [256..1279] four tables of the operator
// "append ZH_CRC_SLICE zero bytes" applied to a 32-bit state, one per state byte.
// Also the two sums of Adler-32 (src/frame.c:74-138) per max-block, for zlib framing: A = sum of the bytes and
// Bw = sum of (n - i) * byte[i], both mod 65521 — appending the block to a running (a, b) is then
// b += n * a + Bw, a += A (zultra_adler32_append).
#define ZH_ADLER_MOD 65521u
__global__ void __launch_bounds__(ZH_CRC_THREADS)
zh_crc32_blocks(const uint8_t *__restrict__ data, const zh_block_t *__restrict__ blocks, const uint32_t *__restrict__ tables,
                uint32_t *crc_out, uint32_t *adler_out /* 2 per block */) {
   __shared__ uint32_t T[256 + 1024];
   __shared__ uint32_t part[ZH_CRC_THREADS];
   __shared__ uint32_t asum[2];
   const zh_block_t blk = blocks[blockIdx.x];
   const uint8_t *p = data + blk.win_off + blk.prev;
   const uint32_t n = blk.n, tid = threadIdx.x;
   for (uint32_t k = tid; k < 256 + 1024; k += ZH_CRC_THREADS) T[k] = tables[k];
   if (tid < 2) asum[tid] = 0;
   __syncthreads();

   // slices are aligned to the END of the block: slice 0 is the short one, all later slices are full
   const uint32_t nslices = (n + ZH_CRC_SLICE - 1) / ZH_CRC_SLICE;
   const uint32_t first = n - (nslices - 1) * ZH_CRC_SLICE;   // 1..ZH_CRC_SLICE bytes
   uint32_t total = 0;                                        // raw CRC state of the bytes folded so far (no pre/post inversion)
   for (uint32_t s0 = 0; s0 < nslices; s0 += ZH_CRC_THREADS) {
      const uint32_t sl = s0 + tid;
      uint32_t c = 0;
      if (sl < nslices) {
         const uint32_t beg = sl == 0 ? 0 : first + (sl - 1) * ZH_CRC_SLICE;
         const uint32_t len = sl == 0 ? first : ZH_CRC_SLICE;
         uint32_t s1 = 0, s2 = 0;   // slice sums: bytes, and (len - k) * byte[k]  (< 2^24 for 256-byte slices)
         for (uint32_t k = 0; k < len; k++) {
            const uint32_t d = p[beg + k];
            c = (c >> 8) ^ T[(c ^ d) & 0xff];
            s1 += d;
            s2 += (len - k) * d;
         }
         // weight of byte k of this slice inside the block = n - (beg + k) = (n - beg - len) + (len - k)
         const uint64_t after = (uint64_t)(n - beg - len) % ZH_ADLER_MOD;
         atomicAdd(&asum[0], s1 % ZH_ADLER_MOD);
         atomicAdd(&asum[1], (uint32_t)((after * (s1 % ZH_ADLER_MOD) + s2) % ZH_ADLER_MOD));
      }
      part[tid] = c;
      __syncthreads();
      if (tid == 0) {
         const uint32_t cnt = min((uint32_t)ZH_CRC_THREADS, nslices - s0);
         for (uint32_t k = 0; k < cnt; k++) {
            // state after appending a full slice = shift(total) ^ state(slice); the very first slice starts from 0
            if (s0 + k > 0)
               total = T[256 + (total & 0xff)] ^ T[512 + ((total >> 8) & 0xff)] ^ T[768 + ((total >> 16) & 0xff)] ^ T[1024 + (total >> 24)];
            total ^= part[k];
         }
      }
      __syncthreads();
      if (tid < 2) asum[tid] %= ZH_ADLER_MOD;   // at most 256 addends below 65521 per round: no overflow
      __syncthreads();
   }
   if (tid < 2) adler_out[2 * blockIdx.x + tid] = asum[tid];
   // crc_out holds the linear part: CRC(data) with zero initial state and no final inversion; the affine terms
   // (initial 0xFFFFFFFF, final XOR) are applied by the host combine.
   if (tid == 0) crc_out[blockIdx.x] = total;
}

// The same for SMALL inputs (files mode: a few KiB each), several to a workgroup: the 5 KB of tables are staged once for 256 / spg inputs instead of once per input
// (a 4 KiB input gave sixteen of the 256 threads a slice, and the table load was most of the kernel: 0.8 ms per 32 768 inputs at the tail of every files batch), and
// every thread has a slice. spg = slices per input, a power of two >= ceil(largest input / ZH_CRC_SLICE), at most ZH_CRC_THREADS; thread t works on input
// blockIdx.x * (ZH_CRC_THREADS / spg) + t / spg, slice t % spg.
__global__ void __launch_bounds__(ZH_CRC_THREADS)
zh_crc32_small(const uint8_t *__restrict__ data, const zh_block_t *__restrict__ blocks, uint32_t nblocks, const uint32_t *__restrict__ tables,
               uint32_t *crc_out, uint32_t *adler_out /* 2 per block */, uint32_t spg) {
   __shared__ uint32_t T[256 + 1024];
   __shared__ uint32_t part[ZH_CRC_THREADS];
   __shared__ uint32_t asum[ZH_CRC_THREADS][2];   // (per input of the workgroup: the first ZH_CRC_THREADS / spg rows)
   const uint32_t tid = threadIdx.x, per = ZH_CRC_THREADS / spg, g = tid / spg, sl = tid % spg;
   const uint32_t b = blockIdx.x * per + g;
   for (uint32_t k = tid; k < 256 + 1024; k += ZH_CRC_THREADS) T[k] = tables[k];
   if (tid < per) asum[tid][0] = asum[tid][1] = 0;
   __syncthreads();
   uint32_t n = 0, nslices = 0, c = 0;
   if (b < nblocks) {
      const zh_block_t blk = blocks[b];
      const uint8_t *p = data + blk.win_off + blk.prev;
      n = blk.n;
      nslices = (n + ZH_CRC_SLICE - 1) / ZH_CRC_SLICE;   // <= spg
      if (sl < nslices) {
         const uint32_t first = n - (nslices - 1) * ZH_CRC_SLICE;
         const uint32_t beg = sl == 0 ? 0 : first + (sl - 1) * ZH_CRC_SLICE;
         const uint32_t len = sl == 0 ? first : ZH_CRC_SLICE;
         uint32_t s1 = 0, s2 = 0;
         for (uint32_t k = 0; k < len; k++) {
            const uint32_t d = p[beg + k];
            c = (c >> 8) ^ T[(c ^ d) & 0xff];
            s1 += d;
            s2 += (len - k) * d;
         }
         const uint64_t after = (uint64_t)(n - beg - len) % ZH_ADLER_MOD;
         atomicAdd(&asum[g][0], s1 % ZH_ADLER_MOD);   // (at most 256 addends below 65521: no overflow)
         atomicAdd(&asum[g][1], (uint32_t)((after * (s1 % ZH_ADLER_MOD) + s2) % ZH_ADLER_MOD));
      }
   }
   part[tid] = c;
   __syncthreads();
   if (b < nblocks && sl == 0) {
      uint32_t total = 0;
      for (uint32_t k = 0; k < nslices; k++) {
         if (k > 0) total = T[256 + (total & 0xff)] ^ T[512 + ((total >> 8) & 0xff)] ^ T[768 + ((total >> 16) & 0xff)] ^ T[1024 + (total >> 24)];
         total ^= part[g * spg + k];
      }
      crc_out[b] = total;
      adler_out[2 * b] = asum[g][0] % ZH_ADLER_MOD;
      adler_out[2 * b + 1] = asum[g][1] % ZH_ADLER_MOD;
   }
}
#endif
