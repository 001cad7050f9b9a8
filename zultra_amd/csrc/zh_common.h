// zh_common.h — constants, descriptors and RFC 1951 symbol arithmetic shared by the kernels and the host
// layer of the MI355X deflate block compressor.
//
// Vocabulary (follows the reference): a *max-block* is up to nMaxBlockSize input bytes
// compressed as one unit (libzultra.c:269-403); its *window* is the preceding <=32 KiB of raw input
// (history) followed by the block; the splitter cuts a max-block into <=64 *sub-blocks*, each of which
// becomes one deflate block (static or dynamic Huffman, or stored if it does not shrink).
#pragma once
#include <stdint.h>

#define ZH_MIN_MATCH 3          // format.h:37
#define ZH_MAX_MATCH 258        // format.h:38
#define ZH_MAX_DIST 32768       // format.h:40
#define ZH_HISTORY 32768        // format.h:41
#define ZH_NMATCH 8             // private.h:49 NMATCHES_PER_OFFSET
#define ZH_LEAVE_ALONE 40       // private.h:52
#define ZH_MAX_SPLITS 64        // private.h:56
#define ZH_NLIT 288             // format.h:44
#define ZH_NDIST 32             // format.h:49
#define ZH_NCL 19               // format.h:43
#define ZH_EOB 256
#define ZH_MIN_BLOCK 32768      // libzultra.c:89
#define ZH_MAX_BLOCK 2097152    // libzultra.c:91

// One max-block of a batch: window = data + win_off, `prev` history bytes then `n` block bytes.
// This is the argument tuple of the five calls at libzultra.c:287-343.
typedef struct zh_block_s {
   uint64_t win_off;
   uint32_t prev;
   uint32_t n;
} zh_block_t;

// Matchfinder segment (internal to the device layer). Match rows only depend on the 32 KiB before a position and the
// <= 258 bytes after it, so a max-block of any size is cut into segments that fit the LDS window: `prev` bytes of history,
// `n` positions that get rows, `tail` bytes of look-ahead (0 at the end of the max-block, where the reference clamps the
// match length too). A max-block of <= 64 KiB is one segment.
typedef struct zh_seg_s {
   uint64_t win_off;   // segment window = data + win_off
   uint32_t prev, n, tail;
   uint32_t block;     // max-block it belongs to (index into the run's block list)
   uint64_t row_off;   // its first row, in positions from the start of the max-block's rows
} zh_seg_t;
#define ZH_SEG_WINDOW 98304u                                   // LDS window of the matchfinder kernels
#define ZH_SEG_POSITIONS (ZH_SEG_WINDOW - ZH_HISTORY - ZH_MAX_MATCH)   // rows per segment of a large max-block

// Result of one sub-block (device -> host). `bits_off` is the byte offset of its phase-0 bitstream in the
// batch payload buffer. BFINAL/BTYPE, the stored fallback and the bit carry are applied by the stitcher
// (libzultra.c:327-398), which needs exactly these fields.
typedef struct zh_subblock_s {
   uint32_t block;         // index of the max-block in the batch
   uint32_t start;         // offset of the sub-block inside the max-block
   uint32_t size;          // input bytes
   uint32_t is_dynamic;    // 1 -> BTYPE 2, 0 -> BTYPE 1  (libzultra.c:323,332)
   int32_t static_cost;    // blockdeflate.c:538
   int32_t dynamic_cost;   // blockdeflate.c:577
   uint32_t failed;        // zultra_block_deflate would have returned -1, or the bits outgrew the slot
   uint32_t reserved;
   uint64_t nbits;         // exact bit count of the body
   uint64_t bits_off;
} zh_subblock_t;

typedef struct zh_match_s {
   uint16_t length;   // private.h:59-62
   uint16_t offset;
} zh_match_t;

// Match rows in HBM (per max-block, match_stride = 8 x max-block size entries): two planes of 16 bytes per position —
// slots 0..3 at [position], slots 4..7 at [max-block size + position]. Rows are longest first and zero padded, and only about
// one position in twenty has four or more matches: the matchfinder writes the second plane only then, and every reader
// fetches it only when slot 3 holds a match. (One 32-byte row per position cost every reader a second 16-byte fetch of
// zeros per position and pass.)
#define ZH_ROW_HI_OFF(match_stride) ((match_stride) / ZH_NMATCH)   /* second plane, in 16-byte units from the block's first row */

#if defined(__HIPCC__) || defined(ZH_EMU)
#define ZH_HD __host__ __device__ __forceinline__
#else
#define ZH_HD static inline
#endif

// ---- RFC 1951 §3.2.5 symbol arithmetic (replaces the lookup tables at blockdeflate.c:45-85) -------------

// distance 1..32768 -> distance code 0..29
ZH_HD int zh_dist_sym(uint32_t d) {
   uint32_t v = d - 1;
   if (v < 4) return (int)v;
   int nb = 31 - __builtin_clz(v);
   return 2 * nb + (int)((v >> (nb - 1)) & 1u);
}
ZH_HD int zh_dist_xbits(int sym) { return sym < 4 ? 0 : (sym >> 1) - 1; }
ZH_HD uint32_t zh_dist_base(int sym) {
   if (sym < 4) return (uint32_t)sym + 1;
   int xb = (sym >> 1) - 1;
   return ((2u + (uint32_t)(sym & 1)) << xb) + 1;
}
// match length 3..258 -> index of the length code (0..28, i.e. symbol 257+idx)
ZH_HD int zh_len_idx(uint32_t len) {
   uint32_t v = len - 3;
   if (v < 8) return (int)v;
   if (v >= 255) return 28;
   int nb = 31 - __builtin_clz(v);
   return 4 * (nb - 1) + (int)((v >> (nb - 2)) & 3u);
}
ZH_HD int zh_lenidx_xbits(int idx) { return (idx < 8 || idx == 28) ? 0 : (idx >> 2) - 1; }
ZH_HD uint32_t zh_lenidx_base(int idx) {   // base length of the code
   if (idx < 8) return (uint32_t)idx + 3;
   if (idx == 28) return 258;
   int xb = (idx >> 2) - 1;
   return ((4u + (uint32_t)(idx & 3)) << xb) + 3;
}
ZH_HD int zh_static_lit_len(int s) { return s < 144 ? 8 : (s < 256 ? 9 : (s < 280 ? 7 : 8)); }
