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
#include <stdint.h>

#include "zh_common.h"

typedef struct zh_stitch_item_s {
   uint64_t dst_bit;    // bit offset of the sub-block's first header bit in the stream buffer
   uint32_t stored;     // 1: stored fallback
   uint32_t is_final;   // BFINAL of the (last piece of the) sub-block
} zh_stitch_item_t;

// Serial planner. phase = pending bits (0..7) before the first sub-block. Returns 0, or -1 where the reference fails
// with ZULTRA_ERROR_DST (its per-max-block buffer of 1 + bs + 5*(bs/65535+1) bytes overflows, libzultra.c:115,382).
// *end_bit = bit offset after the last sub-block (relative to the same origin as dst_bit; origin = start of the byte
// that holds the pending bits).
static inline int zh_stitch_plan(uint32_t phase, const zh_subblock_t *subs, uint32_t nsubs, uint32_t max_block_size, int final_block,
                                 zh_stitch_item_t *items, uint64_t *end_bit) {
   const uint64_t blockbuf_cap = 1 + (uint64_t)max_block_size + 5 * ((uint64_t)max_block_size / 65535 + 1);
   uint64_t bit = phase;
   uint64_t block_base = 0;
   uint32_t cur_block = 0xFFFFFFFFu;
   for (uint32_t k = 0; k < nsubs; k++) {
      const zh_subblock_t *sb = &subs[k];
      if (sb->block != cur_block) {
         cur_block = sb->block;
         block_base = bit >> 3;   // the reference's writer offset restarts at 0 per max-block, pending bits carry
      }
      const int last_of_block = (k + 1 == nsubs) || (subs[k + 1].block != sb->block);
      const uint32_t is_final = ((int)sb->block == final_block && last_of_block) ? 1u : 0u;
      const uint32_t nacc = (uint32_t)(bit & 7);
      const uint32_t c0 = (nacc + 3) & 7;
      const uint64_t o0 = ((bit >> 3) - block_base) + ((nacc + 3) >> 3);
      if (o0 > blockbuf_cap) return -1;
      const uint64_t body_bytes = ((uint64_t)c0 + sb->nbits) >> 3;
      if (items) {
         items[k].dst_bit = bit;
         items[k].is_final = is_final;
      }
      if (!sb->failed && body_bytes <= sb->size && o0 + body_bytes <= blockbuf_cap) {
         if (items) items[k].stored = 0;
         bit += 3 + sb->nbits;
      }
      else {
         if (items) items[k].stored = 1;
         uint32_t rem = sb->size;
         while (rem) {
            const uint32_t piece = rem > 65535 ? 65535 : rem;
            bit = (bit + 3 + 7) & ~7ull;   // header bits, pad to a byte
            if (((bit >> 3) - block_base) + 4 + piece > blockbuf_cap) return -1;
            bit += 32 + 8ull * piece;
            rem -= piece;
         }
      }
   }
   *end_bit = bit;
   return 0;
}

#if defined(__HIPCC__) || defined(ZH_EMU)
#include <zh_platform.h>

#define ZH_STITCH_THREADS 256

__device__ __forceinline__ void zh_or_bits(uint32_t *out, uint64_t bit, uint32_t value, uint32_t nbits) {
   // nbits <= 16: at most two dwords
   const uint64_t v = (uint64_t)value << (bit & 31);
   atomicOr(&out[bit >> 5], (uint32_t)v);
   if (((bit & 31) + nbits) > 32) atomicOr(&out[(bit >> 5) + 1], (uint32_t)(v >> 32));
}

// out must be zero-filled. One workgroup per sub-block.
__global__ void __launch_bounds__(ZH_STITCH_THREADS)
zh_stitch(const zh_subblock_t *__restrict__ subs, const zh_stitch_item_t *__restrict__ items, const zh_block_t *__restrict__ blocks,
          const uint8_t *__restrict__ data, const uint8_t *__restrict__ payload, uint32_t *out) {
   const zh_subblock_t sb = subs[blockIdx.x];
   const zh_stitch_item_t it = items[blockIdx.x];
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
#endif
