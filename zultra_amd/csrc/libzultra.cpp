// libzultra.cpp — the drop-in libzultra API (include/libzultra.h) on top of the MI355X device layer.
//
// Host C++ mirror of reference src/libzultra.c (stream state machine, :200-514; in-memory entry, :576-619),
// src/frame.c (gzip / zlib / raw framing and checksums) and src/dictionary.c. The per-max-block section of the
// reference's loop (:287-403) is replaced by batched calls into include/zultra_hip.h; what depends on the running
// bit phase (3 header bits, stored fallback, bit carry, :327-398 and :414-436) is zultra_hip_stitch below.
// There is no CPU implementation of the hot path in this library: without a HIP device, stream init fails.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "../../include/libzultra.h"
#include "../../include/zultra_hip.h"

// ================================================================================================================
// Framing and checksums (reference src/frame.c)
// ================================================================================================================

static uint32_t g_crc_tab[8][256];
static std::once_flag g_crc_once;

static void crc_init() {
   for (uint32_t i = 0; i < 256; i++) {
      uint32_t c = i;
      for (int k = 0; k < 8; k++) c = (c >> 1) ^ ((c & 1) ? 0xEDB88320u : 0);
      g_crc_tab[0][i] = c;
   }
   for (uint32_t i = 0; i < 256; i++)
      for (int t = 1; t < 8; t++) g_crc_tab[t][i] = (g_crc_tab[t - 1][i] >> 8) ^ g_crc_tab[0][g_crc_tab[t - 1][i] & 0xff];
}

// CRC-32 (reflected 0xEDB88320), slicing-by-8; same function of the data as frame.c:324-354.
static uint32_t crc32_update(uint32_t crc, const uint8_t *p, size_t n) {
   std::call_once(g_crc_once, crc_init);
   crc = ~crc;
   while (n && ((uintptr_t)p & 7)) {
      crc = (crc >> 8) ^ g_crc_tab[0][(crc ^ *p++) & 0xff];
      n--;
   }
   while (n >= 8) {
      uint64_t v;
      memcpy(&v, p, 8);
      v ^= crc;
      crc = g_crc_tab[7][v & 0xff] ^ g_crc_tab[6][(v >> 8) & 0xff] ^ g_crc_tab[5][(v >> 16) & 0xff] ^ g_crc_tab[4][(v >> 24) & 0xff] ^
            g_crc_tab[3][(v >> 32) & 0xff] ^ g_crc_tab[2][(v >> 40) & 0xff] ^ g_crc_tab[1][(v >> 48) & 0xff] ^ g_crc_tab[0][v >> 56];
      p += 8;
      n -= 8;
   }
   while (n--) crc = (crc >> 8) ^ g_crc_tab[0][(crc ^ *p++) & 0xff];
   return ~crc;
}

// ---- folding device-computed per-block CRCs into the running gzip checksum -----------------------------------------
// With R(s, D) the raw register after feeding D from state s, R(s, D) = Z_n(s) ^ R(0, D) (n = |D|, Z_n = "feed n zero
// bytes", linear over GF(2)); the device returns R(0, block). Z_n is kept as a 32x32 bit matrix per length.
namespace {
struct CrcShiftOp {
   uint32_t col[32];   // image of each state bit
   uint32_t apply(uint32_t v) const {
      uint32_t r = 0;
      for (int b = 0; v; v >>= 1, b++)
         if (v & 1) r ^= col[b];
      return r;
   }
};
static void op_square(CrcShiftOp &dst, const CrcShiftOp &src) {
   for (int b = 0; b < 32; b++) dst.col[b] = src.apply(src.col[b]);
}
static CrcShiftOp crc_shift_op(size_t nbytes) {
   std::call_once(g_crc_once, crc_init);
   CrcShiftOp pw, res, tmp;
   for (int b = 0; b < 32; b++) {           // one zero byte
      uint32_t v = 1u << b;
      pw.col[b] = (v >> 8) ^ g_crc_tab[0][v & 0xff];
      res.col[b] = 1u << b;                  // identity
   }
   for (size_t n = nbytes; n; n >>= 1) {
      if (n & 1) {
         for (int b = 0; b < 32; b++) tmp.col[b] = pw.apply(res.col[b]);
         res = tmp;
      }
      op_square(tmp, pw);
      pw = tmp;
   }
   return res;
}
static std::mutex g_op_mutex;
static size_t g_op_len[4] = {0, 0, 0, 0};
static CrcShiftOp g_op[4];
static int g_op_next = 0;
}   // namespace

extern "C" uint32_t zultra_crc32_append(uint32_t crc, uint32_t block_linear_crc, size_t block_len) {
   CrcShiftOp op;
   {
      std::lock_guard<std::mutex> lk(g_op_mutex);
      int hit = -1;
      for (int i = 0; i < 4; i++)
         if (g_op_len[i] == block_len && block_len) hit = i;
      if (hit < 0) {
         hit = g_op_next;
         g_op_next = (g_op_next + 1) & 3;
         g_op[hit] = crc_shift_op(block_len);
         g_op_len[hit] = block_len;
      }
      op = g_op[hit];
   }
   return ~(op.apply(~crc) ^ block_linear_crc);
}

extern "C" uint32_t zultra_crc32_append_many(uint32_t crc, const uint32_t *block_linear_crc, const uint32_t *block_len, uint32_t nblocks) {
   for (uint32_t b = 0; b < nblocks; b++) crc = zultra_crc32_append(crc, block_linear_crc[b], block_len[b]);
   return crc;
}

extern "C" uint32_t zultra_adler32_append(uint32_t adler, uint32_t block_sum, uint32_t block_weighted_sum, size_t block_len) {
   const uint64_t M = 65521;
   const uint64_t a = adler & 0xffff, b = (adler >> 16) & 0xffff;
   const uint64_t b2 = (b + (uint64_t)(block_len % M) * a + block_weighted_sum) % M;
   const uint64_t a2 = (a + block_sum) % M;
   return (uint32_t)((b2 << 16) | a2);
}

// Adler-32 (frame.c:74-138): sums modulo 65521, reduced every 5552 bytes.
static uint32_t adler32_update(uint32_t adler, const uint8_t *p, size_t n) {
   uint32_t a = adler & 0xffff, b = (adler >> 16) & 0xffff;
   while (n) {
      size_t chunk = n > 5552 ? 5552 : n;
      for (size_t k = 0; k < chunk; k++) {
         a += p[k];
         b += a;
      }
      a %= 65521u;
      b %= 65521u;
      p += chunk;
      n -= chunk;
   }
   return a | (b << 16);
}

extern "C" int zultra_frame_get_header_size(const unsigned int nFlags, const void *pDict, const int nDictSize) {
   if (nFlags & ZULTRA_FLAG_GZIP_FRAMING) return 10;
   if (nFlags & ZULTRA_FLAG_ZLIB_FRAMING) return (pDict && nDictSize) ? 6 : 2;
   return 0;
}

extern "C" int zultra_frame_encode_header(unsigned char *p, const int nMax, const unsigned int nFlags, const void *pDict,
                                          const int nDictSize) {
   if (nFlags & ZULTRA_FLAG_GZIP_FRAMING) {
      // RFC 1952: magic, CM=8, FLG=0, MTIME=0, XFL=2 (max compression), OS=255 (frame.c:390-401)
      static const unsigned char gz[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 2, 255};
      if (nMax < 10) return ZULTRA_ENCODE_ERR;
      memcpy(p, gz, 10);
      return 10;
   }
   if (nFlags & ZULTRA_FLAG_ZLIB_FRAMING) {
      // RFC 1950: CMF=0x78, FLG: level 3, FDICT if a dictionary is set, FCHECK (frame.c:412-433)
      const bool has_dict = pDict && nDictSize;
      if (nMax < (has_dict ? 6 : 2)) return ZULTRA_ENCODE_ERR;
      unsigned cmf = 0x78, flg = 0xc0 | (has_dict ? 0x20 : 0);
      flg |= (31 - ((cmf << 8 | flg) % 31)) & 0x1f;
      p[0] = (unsigned char)cmf;
      p[1] = (unsigned char)flg;
      if (!has_dict) return 2;
      uint32_t a = adler32_update(1, (const uint8_t *)pDict, (size_t)nDictSize);
      p[2] = (unsigned char)(a >> 24);
      p[3] = (unsigned char)(a >> 16);
      p[4] = (unsigned char)(a >> 8);
      p[5] = (unsigned char)a;
      return 6;
   }
   return 0;
}

extern "C" zultra_frame_checksum_t zultra_frame_init_checksum(const unsigned int nFlags) {
   if (nFlags & ZULTRA_FLAG_GZIP_FRAMING) return 0;
   if (nFlags & ZULTRA_FLAG_ZLIB_FRAMING) return 1;
   return 0;
}

extern "C" zultra_frame_checksum_t zultra_frame_update_checksum(zultra_frame_checksum_t sum, const void *pData, size_t n,
                                                                const unsigned int nFlags) {
   if (nFlags & ZULTRA_FLAG_GZIP_FRAMING) return crc32_update(sum, (const uint8_t *)pData, n);
   if (nFlags & ZULTRA_FLAG_ZLIB_FRAMING) return adler32_update(sum, (const uint8_t *)pData, n);
   return 0;
}

extern "C" int zultra_frame_get_footer_size(const unsigned int nFlags) {
   if (nFlags & ZULTRA_FLAG_GZIP_FRAMING) return 8;
   if (nFlags & ZULTRA_FLAG_ZLIB_FRAMING) return 4;
   return 0;
}

extern "C" int zultra_frame_encode_footer(unsigned char *p, const int nMax, const zultra_frame_checksum_t sum, long long nOriginalSize,
                                          const unsigned int nFlags) {
   if (nFlags & ZULTRA_FLAG_GZIP_FRAMING) {
      if (nMax < 8) return ZULTRA_ENCODE_ERR;
      for (int k = 0; k < 4; k++) p[k] = (unsigned char)(sum >> (8 * k));                                       // CRC32, LE
      for (int k = 0; k < 4; k++) p[4 + k] = (unsigned char)((unsigned long long)nOriginalSize >> (8 * k));   // ISIZE mod 2^32
      return 8;
   }
   if (nFlags & ZULTRA_FLAG_ZLIB_FRAMING) {
      if (nMax < 4) return ZULTRA_ENCODE_ERR;
      for (int k = 0; k < 4; k++) p[k] = (unsigned char)(sum >> (8 * (3 - k)));   // Adler-32, BE
      return 4;
   }
   return 0;
}

// ================================================================================================================
// Dictionary file helper (reference src/dictionary.c:49-104)
// ================================================================================================================

extern "C" zultra_status_t zultra_dictionary_load(const char *name, void **ppData, int *pSize) {
   unsigned char *buf = NULL;
   int n = 0;
   if (name) {
      FILE *f = fopen(name, "rb");
      if (!f) return ZULTRA_ERROR_DICTIONARY;
      buf = (unsigned char *)malloc(HISTORY_SIZE);
      if (!buf) {
         fclose(f);
         return ZULTRA_ERROR_MEMORY;
      }
      fseek(f, 0, SEEK_END);
      long long sz = ftello(f);
      if (sz > HISTORY_SIZE)
         fseek(f, -HISTORY_SIZE, SEEK_END);   // last 32 KiB of the file
      else
         fseek(f, 0, SEEK_SET);
      n = (int)fread(buf, 1, HISTORY_SIZE, f);
      if (n < 0) n = 0;
      fclose(f);
   }
   *ppData = buf;
   *pSize = n;
   return ZULTRA_OK;
}

extern "C" void zultra_dictionary_free(void **ppData) {
   if (ppData && *ppData) {
      free(*ppData);
      *ppData = NULL;
   }
}

// ================================================================================================================
// Stitcher (reference src/libzultra.c:327-398, 414-436; bit writer src/huffman/bitwriter.c:63-98)
// ================================================================================================================

namespace {
struct Sink {
   uint8_t *out;
   size_t cap, pos;
   uint32_t acc, nacc;
   bool overflow;
   bool dry;   // count only (out == NULL): used to plan bit offsets of later shards
   inline void byte(uint8_t b) {
      if (!dry) {
         if (pos < cap)
            out[pos] = b;
         else
            overflow = true;
      }
      pos++;
   }
   inline void bits(uint32_t v, uint32_t n) {   // n <= 16
      acc |= v << nacc;
      nacc += n;
      while (nacc >= 8) {
         byte((uint8_t)acc);
         acc >>= 8;
         nacc -= 8;
      }
   }
   inline void pad() {
      if (nacc) {
         byte((uint8_t)(acc & ((1u << nacc) - 1)));
         acc = 0;
         nacc = 0;
      }
   }
   // append nbits of a phase-0 bit string
   void append(const uint8_t *src, uint64_t nbits) {
      const uint64_t full = nbits >> 3;
      if (dry) {
         pos += full;
         if (nbits & 7) bits(0, (uint32_t)(nbits & 7));
         return;
      }
      if (nacc == 0) {
         size_t room = pos < cap ? cap - pos : 0;
         size_t k = full < room ? (size_t)full : room;
         memcpy(out + pos, src, k);
         if (k < full) overflow = true;
         pos += full;
      }
      else {
         // shift-merge, 8 source bytes per step
         uint64_t i = 0;
         const uint32_t sh = nacc;
         uint64_t carry = acc;
         if (pos + full + 8 <= cap) {
            for (; i + 8 <= full; i += 8) {
               uint64_t v;
               memcpy(&v, src + i, 8);
               uint64_t o = carry | (v << sh);
               memcpy(out + pos, &o, 8);
               pos += 8;
               carry = v >> (64 - sh);
            }
         }
         acc = (uint32_t)carry;
         for (; i < full; i++) bits(src[i], 8);
      }
      if (nbits & 7) bits(src[full] & ((1u << (nbits & 7)) - 1), (uint32_t)(nbits & 7));
   }
};
}   // namespace

extern "C" size_t zultra_hip_stitch(zultra_hip_bitstate_t *state, const zultra_hip_subblock_t *subs, uint32_t nsubs,
                                    const uint8_t *payload, const uint8_t *raw, const uint64_t *raw_off, uint32_t max_block_size,
                                    int final_block, uint8_t *out, size_t out_cap) {
   Sink s{out, out_cap, 0, state->acc, state->nacc, false, out == NULL};
   // capacity of the reference's per-max-block output buffer (libzultra.c:115): exceeding it is ZULTRA_ERROR_DST
   const size_t blockbuf_cap = 1 + (size_t)max_block_size + 5 * ((size_t)max_block_size / 65535 + 1);
   size_t block_base = 0;
   uint32_t cur_block = 0xFFFFFFFFu;

   for (uint32_t k = 0; k < nsubs; k++) {
      const zultra_hip_subblock_t &sb = subs[k];
      if (sb.block != cur_block) {
         cur_block = sb.block;
         block_base = s.pos;   // bitwriter offset restarts at 0 for every max-block, pending bits carry (:427-434)
      }
      const bool last_of_block = (k + 1 == nsubs) || (subs[k + 1].block != sb.block);
      const uint32_t is_final = ((int)sb.block == final_block && last_of_block) ? 1u : 0u;   // :328

      // where the three header bits leave the reference's writer (:329-337)
      const uint32_t c0 = (s.nacc + 3) & 7;
      const size_t o0 = (s.pos - block_base) + ((s.nacc + 3) >> 3);
      if (o0 > blockbuf_cap) return (size_t)-1;
      const uint64_t body_bytes = ((uint64_t)c0 + sb.nbits) >> 3;

      if (!sb.failed && body_bytes <= sb.size && o0 + body_bytes <= blockbuf_cap) {   // :345-347
         s.bits(is_final, 1);
         s.bits(1 + sb.is_dynamic, 2);
         s.append(payload + sb.bits_off, sb.nbits);
      }
      else {
         // stored, pieces of at most 65535 bytes (:350-397)
         const uint8_t *src = s.dry ? NULL : raw + raw_off[sb.block] + sb.start;
         uint32_t rem = sb.size;
         while (rem) {
            const uint32_t piece = rem > 65535 ? 65535 : rem;
            s.bits((rem > 65535) ? 0 : is_final, 1);
            s.bits(0, 2);
            s.pad();
            if ((s.pos - block_base) + 4 + piece > blockbuf_cap) return (size_t)-1;   // :382
            s.byte((uint8_t)(piece & 0xff));
            s.byte((uint8_t)(piece >> 8));
            s.byte((uint8_t)((piece & 0xff) ^ 0xff));
            s.byte((uint8_t)((piece >> 8) ^ 0xff));
            if (!s.dry) {
               if (s.pos + piece <= s.cap)
                  memcpy(s.out + s.pos, src, piece);
               else
                  s.overflow = true;
            }
            s.pos += piece;
            if (!s.dry) src += piece;
            rem -= piece;
         }
      }
   }
   state->acc = s.acc;
   state->nacc = s.nacc;
   return s.overflow ? (size_t)-1 : s.pos;
}

extern "C" size_t zultra_hip_stitch_finish(zultra_hip_bitstate_t *state, uint8_t *out, size_t out_cap) {
   if (!state->nacc) return 0;
   if (out_cap < 1) return (size_t)-1;
   out[0] = (uint8_t)(state->acc & ((1u << state->nacc) - 1));   // libzultra.c:414-417
   state->acc = 0;
   state->nacc = 0;
   return 1;
}

// ================================================================================================================
// Streaming API (reference src/libzultra.c:82-565)
// ================================================================================================================

enum {
   ST_HAS_DICTIONARY = 1,
   ST_HEADER_EMITTED = 2,
   ST_FINALIZED = 4,
   ST_FOOTER_EMITTED = 8,
   ST_STREAM_ENDED = 16,
};

static int g_device = -1;
static std::mutex g_ctx_mutex;
struct CachedCtx {
   zultra_hip_ctx_t *ctx;
   int device;
   uint32_t max_block, max_blocks;
};
static std::vector<CachedCtx> g_ctx_pool;   // contexts released by finished streams, reused by later ones

static int zh_pick_device() {
   if (g_device >= 0) return g_device;
   const char *e = getenv("ZULTRA_HIP_DEVICE");
   return e ? atoi(e) : 0;
}

extern "C" void zultra_set_device(int nDevice) { g_device = nDevice; }

// The devices zultra_memory_compress spreads an input over: zultra_set_devices, else ZULTRA_HIP_DEVICES="0,1,2,..." (a device may be
// named more than once: "0,0" = two contexts on device 0, the kernels of one batch next to the transfers of another), else none.
static std::vector<int> g_devices;
extern "C" int zultra_set_devices(const int *pDevices, int nDevices) {
   std::lock_guard<std::mutex> lk(g_ctx_mutex);
   g_devices.clear();
   const int have = zultra_hip_device_count();
   for (int i = 0; i < nDevices; i++) {
      if (pDevices[i] < 0 || pDevices[i] >= have) {
         g_devices.clear();
         return -1;
      }
      g_devices.push_back(pDevices[i]);
   }
   return nDevices;
}
static std::vector<int> zh_pick_devices() {
   {
      std::lock_guard<std::mutex> lk(g_ctx_mutex);
      if (!g_devices.empty()) return g_devices;
   }
   std::vector<int> v;
   const char *e = getenv("ZULTRA_HIP_DEVICES");
   if (e) {
      const int have = zultra_hip_device_count();
      for (const char *p = e; *p;) {
         char *end = NULL;
         const long d = strtol(p, &end, 10);
         if (end == p) break;
         if (d >= 0 && d < have) v.push_back((int)d);
         p = (*end == ',') ? end + 1 : end;
      }
   }
   return v;
}

static uint32_t clamp_block(uint32_t n) {
   if (!n) n = ZULTRA_DEFAULT_MAX_BLOCK_SIZE;
   if (n < 32768) n = 32768;
   if (n > 2097152) n = 2097152;
   return n;
}

static zultra_hip_ctx_t *ctx_acquire_on(int dev, uint32_t bs, uint32_t want_blocks) {
   std::lock_guard<std::mutex> lk(g_ctx_mutex);
   for (size_t i = 0; i < g_ctx_pool.size(); i++) {
      if (g_ctx_pool[i].device == dev && g_ctx_pool[i].max_block == bs && g_ctx_pool[i].max_blocks >= want_blocks) {
         zultra_hip_ctx_t *c = g_ctx_pool[i].ctx;
         g_ctx_pool.erase(g_ctx_pool.begin() + (long)i);
         return c;
      }
   }
   // a smaller cached context of the same geometry would only waste memory next to the new one
   for (size_t i = 0; i < g_ctx_pool.size();) {
      if (g_ctx_pool[i].device == dev && g_ctx_pool[i].max_block == bs) {
         zultra_hip_destroy(g_ctx_pool[i].ctx);
         g_ctx_pool.erase(g_ctx_pool.begin() + (long)i);
      }
      else
         i++;
   }
   return zultra_hip_create(dev, bs, want_blocks);
}
static zultra_hip_ctx_t *ctx_acquire(uint32_t bs, uint32_t want_blocks) { return ctx_acquire_on(zh_pick_device(), bs, want_blocks); }

// A finished stream's context goes back under what the CONTEXT says it is (device, block size, capacity) — not under what the
// stream asked for or what zultra_set_device says now.
static void ctx_release(zultra_hip_ctx_t *c) {
   if (!c) return;
   const char *e = getenv("ZULTRA_HIP_CACHE");
   if (e && atoi(e) == 0) {
      zultra_hip_destroy(c);
      return;
   }
   CachedCtx cc;
   cc.ctx = c;
   zultra_hip_ctx_info(c, &cc.device, &cc.max_block, &cc.max_blocks, NULL);
   std::lock_guard<std::mutex> lk(g_ctx_mutex);
   size_t same = 0, first_same = 0;   // at most two kept per device
   for (size_t i = g_ctx_pool.size(); i-- > 0;)
      if (g_ctx_pool[i].device == cc.device) {
         same++;
         first_same = i;
      }
   if (same >= 2) {
      zultra_hip_destroy(g_ctx_pool[first_same].ctx);
      g_ctx_pool.erase(g_ctx_pool.begin() + (long)first_same);
   }
   g_ctx_pool.push_back(cc);
}

extern "C" int zultra_release_cached_contexts(void) {
   std::lock_guard<std::mutex> lk(g_ctx_mutex);
   const int n = (int)g_ctx_pool.size();
   for (size_t i = 0; i < g_ctx_pool.size(); i++) zultra_hip_destroy(g_ctx_pool[i].ctx);
   g_ctx_pool.clear();
   return n;
}

struct _zultra_compressor_s {
   unsigned flags;
   uint32_t max_block;
   uint32_t batch_blocks;       // max-blocks per device batch
   uint64_t flush_bytes;        // staged full max-blocks of this many bytes are compressed when a call's input is used up (0: only when the staging area is full)
   const void *dict;
   int dict_size;
   unsigned state;

   zultra_hip_ctx_t *hip;
   zultra_hip_bitstate_t bitstate;

   uint8_t *in;                 // [32 KiB history][batch_blocks * max_block]
   size_t in_bytes;             // input bytes staged after the history area
   int prev;                    // valid history bytes (end-aligned in the history area)

   uint8_t *out;                // stitched bytes of the last batch
   size_t out_cap, out_pos, out_pending;

   unsigned char frame[16];
   size_t frame_pos, frame_pending;

   // per max-block of a batch, batch_blocks entries each, from the caller's allocator (libzultra.c:94-147: every buffer of a stream
   // comes from zalloc; what does not here is the device context — device memory and pinned staging — which belongs to the backend)
   zultra_hip_block_t *blocks;
   uint64_t *raw_off;
   uint32_t *crc;
   uint32_t *adler_parts;       // two per max-block
   bool host_stitch;            // ZULTRA_HIP_HOST_STITCH=1: stitch on the host (A/B checking)
};

static void *default_zalloc(void *, unsigned int items, unsigned int size) { return malloc((size_t)items * size); }
static void default_zfree(void *, void *p) { free(p); }

static zultra_status_t stream_init_sized(zultra_stream_t *s, unsigned flags, unsigned bs_in, uint32_t batch_blocks) {
   const uint32_t bs = clamp_block(bs_in);
   if (!s->zalloc) s->zalloc = default_zalloc;
   if (!s->zfree) s->zfree = default_zfree;
   s->adler = 0;
   s->state = NULL;

   zultra_compressor_t *c = (zultra_compressor_t *)s->zalloc(s->opaque, 1, (unsigned)sizeof(zultra_compressor_t));
   if (!c) return ZULTRA_ERROR_MEMORY;
   new (c) zultra_compressor_t();
   s->state = c;
   c->flags = flags;
   c->max_block = bs;
   c->dict = NULL;
   c->dict_size = 0;
   c->state = 0;
   c->bitstate.acc = c->bitstate.nacc = 0;
   c->in_bytes = 0;
   c->prev = 0;
   c->out_pos = c->out_pending = 0;
   c->frame_pos = c->frame_pending = 0;
   c->in = c->out = NULL;
   c->hip = NULL;
   c->blocks = NULL;
   c->raw_off = NULL;
   c->crc = c->adler_parts = NULL;
   {
      const char *e = getenv("ZULTRA_HIP_HOST_STITCH");
      c->host_stitch = e && atoi(e) != 0;
   }

   if (!batch_blocks) {
      const char *e = getenv("ZULTRA_HIP_BATCH_BLOCKS");
      batch_blocks = e ? (uint32_t)atoi(e) : (uint32_t)((64u << 20) / bs);
      if (batch_blocks < 2) batch_blocks = 2;
   }
   // device bytes per context (MI355X has 288 GB): sized from the layout the context will really allocate
   const uint64_t budget = 24ull << 30;
   while (batch_blocks > 1 && (uint64_t)zultra_hip_context_bytes_on(zh_pick_device(), bs, batch_blocks) > budget) batch_blocks = batch_blocks - (batch_blocks + 7) / 8;
   c->batch_blocks = batch_blocks;
   {
      const char *e = getenv("ZULTRA_HIP_FLUSH_BYTES");
      c->flush_bytes = e ? (uint64_t)strtoull(e, NULL, 10) : (4ull << 20);
   }
   c->blocks = (zultra_hip_block_t *)s->zalloc(s->opaque, batch_blocks, (unsigned)sizeof(zultra_hip_block_t));
   c->raw_off = (uint64_t *)s->zalloc(s->opaque, batch_blocks, (unsigned)sizeof(uint64_t));
   c->crc = (uint32_t *)s->zalloc(s->opaque, batch_blocks, (unsigned)sizeof(uint32_t));
   c->adler_parts = (uint32_t *)s->zalloc(s->opaque, 2 * batch_blocks, (unsigned)sizeof(uint32_t));
   if (!c->blocks || !c->raw_off || !c->crc || !c->adler_parts) {
      zultra_stream_end(s);
      return ZULTRA_ERROR_MEMORY;
   }

   c->hip = ctx_acquire(bs, batch_blocks);
   if (!c->hip) {
      zultra_stream_end(s);
      return ZULTRA_ERROR_MEMORY;   // no usable HIP device / device memory: there is no CPU path
   }
   const size_t in_size = (size_t)HISTORY_SIZE + (size_t)batch_blocks * bs;
   // worst case per max-block: all sub-blocks stored (libzultra.c:576-587)
   c->out_cap = (size_t)batch_blocks * ((size_t)bs + 6 * 64 + 16) + 16;
   // staging lives in pinned memory owned by the (pooled) device context: it survives this stream and is reused by the next
   c->in = (uint8_t *)zultra_hip_staging(c->hip, 0, in_size);
   c->out = (uint8_t *)zultra_hip_staging(c->hip, 1, c->out_cap);
   if (!c->in || !c->out) {
      zultra_stream_end(s);
      return ZULTRA_ERROR_MEMORY;
   }
   return ZULTRA_OK;
}

extern "C" zultra_status_t zultra_stream_init(zultra_stream_t *s, const unsigned int nFlags, unsigned int nMaxBlockSize) {
   return stream_init_sized(s, nFlags, nMaxBlockSize, 0);
}

extern "C" zultra_status_t zultra_stream_set_dictionary(zultra_stream_t *s, const void *pDict, const int nDictSize) {
   zultra_compressor_t *c = s->state;
   if (c && c->state == 0) {   // only before the first compress call (libzultra.c:180)
      c->dict = pDict;
      c->dict_size = nDictSize;
      c->state |= ST_HAS_DICTIONARY;
      return ZULTRA_OK;
   }
   return ZULTRA_ERROR_COMPRESSION;
}

extern "C" void zultra_stream_end(zultra_stream_t *s) {
   if (s->state && s->zfree) {
      zultra_compressor_t *c = s->state;
      ctx_release(c->hip);
      /* c->in / c->out belong to the device context */
      if (c->blocks) s->zfree(s->opaque, c->blocks);
      if (c->raw_off) s->zfree(s->opaque, c->raw_off);
      if (c->crc) s->zfree(s->opaque, c->crc);
      if (c->adler_parts) s->zfree(s->opaque, c->adler_parts);
      c->~zultra_compressor_t();
      s->zfree(s->opaque, c);
      s->state = NULL;
   }
}

static void drain_frame(zultra_stream_t *s, zultra_compressor_t *c) {
   if (c->frame_pending && s->avail_out) {
      size_t k = s->avail_out < c->frame_pending ? s->avail_out : c->frame_pending;
      memcpy(s->next_out, c->frame + c->frame_pos, k);
      c->frame_pos += k;
      c->frame_pending -= k;
      s->next_out += k;
      s->avail_out -= k;
      s->total_out += k;
   }
}

// Compress `count` max-blocks staged at c->in + HISTORY (the last one `last_n` bytes long).
static zultra_status_t compress_staged(zultra_stream_t *s, zultra_compressor_t *c, uint32_t count, uint32_t last_n, bool final_last) {
   const uint32_t bs = c->max_block;
   if (count > c->batch_blocks) return ZULTRA_ERROR_COMPRESSION;
   const uint64_t base = (uint64_t)HISTORY_SIZE - (uint64_t)c->prev;   // window of block 0 starts here
   size_t consumed = 0;
   for (uint32_t b = 0; b < count; b++) {
      const uint32_t n = (b + 1 == count) ? last_n : bs;
      const uint32_t prev = (b == 0) ? (uint32_t)c->prev : (uint32_t)HISTORY_SIZE;   // blocks are >= 32 KiB unless last
      c->blocks[b].win_off = (uint64_t)HISTORY_SIZE + (uint64_t)b * bs - prev - base;
      c->blocks[b].prev = prev;
      c->blocks[b].n = n;
      c->raw_off[b] = (uint64_t)b * bs;
      consumed += n;
   }
   const size_t data_size = (size_t)c->prev + consumed;
   // the phase this batch starts at is what the batches before it left (libzultra.c:327-398): its stitch goes out with its kernels
   if (!c->host_stitch) (void)zultra_hip_stitch_with_batch(c->hip, 1, c->bitstate.nacc, final_last ? (int)count - 1 : -1);
   int nsubs = zultra_hip_compress_blocks(c->hip, c->in + base, data_size, 0, c->blocks, count);
   if (nsubs <= 0) return ZULTRA_ERROR_COMPRESSION;

   // checksum once per max-block over its bytes (libzultra.c:279): both kinds come from the device, folded here
   if (c->flags & ZULTRA_FLAG_GZIP_FRAMING) {
      if (zultra_hip_block_crc32(c->hip, c->crc) != (int)count) return ZULTRA_ERROR_COMPRESSION;
      for (uint32_t b = 0; b < count; b++) s->adler = zultra_crc32_append(s->adler, c->crc[b], c->blocks[b].n);
   }
   else if (c->flags & ZULTRA_FLAG_ZLIB_FRAMING) {
      if (zultra_hip_block_adler32(c->hip, c->adler_parts) != (int)count) return ZULTRA_ERROR_COMPRESSION;
      for (uint32_t b = 0; b < count; b++) s->adler = zultra_adler32_append(s->adler, c->adler_parts[2 * b], c->adler_parts[2 * b + 1], c->blocks[b].n);
   }

   size_t w = 0;
   if (!c->host_stitch) {
      // device stitch: descriptors -> plan on the host, bits moved by a kernel, one D2H of the finished bytes
      uint64_t end_bit = 0;
      const uint32_t phase = c->bitstate.nacc;
      const uint32_t pending = c->bitstate.acc & ((1u << phase) - 1);
      int rc = zultra_hip_stitch_device(c->hip, &c->bitstate, final_last ? (int)count - 1 : -1, &end_bit);
      if (rc == -2) return ZULTRA_ERROR_DST;
      if (rc != 0) return ZULTRA_ERROR_COMPRESSION;
      const size_t total = (size_t)((end_bit + 7) >> 3);
      if (total > c->out_cap) return ZULTRA_ERROR_DST;
      if (zultra_hip_stream_read(c->hip, c->out, 0, total) != 0) return ZULTRA_ERROR_COMPRESSION;
      c->out[0] |= (uint8_t)pending;
      if (final_last) {
         w = total;   // the last partial byte is already zero padded (libzultra.c:414-417)
         c->bitstate.acc = c->bitstate.nacc = 0;
         c->state |= ST_FINALIZED;
      }
      else {
         w = (size_t)(end_bit >> 3);
         c->bitstate.acc = (end_bit & 7) ? c->out[w] : 0;
      }
   }
   else {
      uint32_t cnt = 0;
      const zultra_hip_subblock_t *subs = zultra_hip_subblocks(c->hip, &cnt);
      size_t psize = 0;
      const uint8_t *payload = zultra_hip_payload(c->hip, &psize);
      w = zultra_hip_stitch(&c->bitstate, subs, cnt, payload, c->in + HISTORY_SIZE, c->raw_off, bs,
                            final_last ? (int)count - 1 : -1, c->out, c->out_cap);
      if (w == (size_t)-1) return ZULTRA_ERROR_DST;
      if (final_last) {
         size_t f = zultra_hip_stitch_finish(&c->bitstate, c->out + w, c->out_cap - w);
         if (f == (size_t)-1) return ZULTRA_ERROR_DST;
         w += f;
         c->state |= ST_FINALIZED;
      }
   }
   c->out_pos = 0;
   c->out_pending = w;

   // slide: the last <= 32 KiB just compressed become the history of what follows (libzultra.c:406-412)
   size_t keep = consumed < (size_t)HISTORY_SIZE ? consumed : (size_t)HISTORY_SIZE;
   if (consumed < (size_t)HISTORY_SIZE) {
      // (only possible for a last, short block; kept for symmetry) old history stays in front
      size_t old = (size_t)c->prev;
      if (old + consumed > (size_t)HISTORY_SIZE) old = (size_t)HISTORY_SIZE - consumed;
      memmove(c->in + HISTORY_SIZE - keep - old, c->in + HISTORY_SIZE - old, old);
      memmove(c->in + HISTORY_SIZE - keep, c->in + HISTORY_SIZE, keep);
      c->prev = (int)(old + keep);
   }
   else {
      memmove(c->in, c->in + HISTORY_SIZE + consumed - HISTORY_SIZE, HISTORY_SIZE);
      c->prev = HISTORY_SIZE;
   }
   const size_t left = c->in_bytes - consumed;
   if (left) memmove(c->in + HISTORY_SIZE, c->in + HISTORY_SIZE + consumed, left);
   c->in_bytes = left;
   c->dict_size = 0;
   return ZULTRA_OK;
}

extern "C" zultra_status_t zultra_stream_compress(zultra_stream_t *s, const int nDoFinalize) {
   zultra_compressor_t *c = s->state;
   zultra_status_t err = ZULTRA_OK;
   if (!c) return ZULTRA_ERROR_COMPRESSION;
   if (c->state & ST_STREAM_ENDED) return ZULTRA_ERROR_COMPRESSION;
   const uint32_t bs = c->max_block;

   do {
      if (!(c->state & ST_HEADER_EMITTED)) {
         c->state |= ST_HEADER_EMITTED;
         int h = zultra_frame_encode_header(c->frame, 16, c->flags, c->dict, c->dict_size);
         if (h < 0)
            err = ZULTRA_ERROR_COMPRESSION;
         else {
            c->frame_pos = 0;
            c->frame_pending = (size_t)h;
         }
         s->adler = zultra_frame_init_checksum(c->flags);
      }
      if (!err) drain_frame(s, c);

      if (!err && !c->prev && c->dict_size && c->dict && c->in_bytes == 0) {
         // preset dictionary = pre-filled history of the first max-block (libzultra.c:250-253)
         int d = c->dict_size > HISTORY_SIZE ? HISTORY_SIZE : c->dict_size;
         memcpy(c->in + HISTORY_SIZE - d, (const uint8_t *)c->dict + (c->dict_size - d), (size_t)d);
         c->prev = d;
      }

      if (!err && !c->frame_pending && !c->out_pending) {
         const size_t cap = (size_t)c->batch_blocks * bs;
         size_t take = s->avail_in;
         if (take > cap - c->in_bytes) take = cap - c->in_bytes;
         memcpy(c->in + HISTORY_SIZE + c->in_bytes, s->next_in, take);
         s->next_in += take;
         s->avail_in -= take;
         s->total_in += take;
         c->in_bytes += take;

         // which staged max-blocks may be compressed now (libzultra.c:269): a full block needs more input behind
         // it, or ZULTRA_FINALIZE; a partial block needs ZULTRA_FINALIZE
         const uint32_t full = (uint32_t)(c->in_bytes / bs);
         const uint32_t rem = (uint32_t)(c->in_bytes % bs);
         uint32_t count = 0, last_n = bs;
         bool final_last = false;
         if (nDoFinalize) {
            count = full + (rem ? 1 : 0);
            if (rem) last_n = rem;
            final_last = (s->avail_in == 0);
         }
         else if (full) {
            // A full max-block may be compressed once input beyond it has been seen (libzultra.c:269). Nothing obliges us to do it
            // at once: blocks are kept until the staging area is full, so that a caller feeding small chunks (the reference's
            // CLI reads 16 KiB at a time) still gets device batches of many max-blocks instead of one launch sequence per block.
            // Neither may a caller that pipes the output on wait for it for long: once the caller has nothing more to give in this call and
            // `flush_bytes` of full blocks are staged (4 MiB unless ZULTRA_HIP_FLUSH_BYTES says otherwise; the reference publishes after every
            // max-block, libzultra.c:424-462), they go to the device. A caller that hands over tens of MiB at once still fills the whole
            // staging area (64 MiB) first: the big batches are for those who have the data.
            const uint32_t ready = (rem || s->avail_in) ? full : full - 1;
            if (c->in_bytes >= cap || (ready >= c->batch_blocks) || (c->flush_bytes && s->avail_in == 0 && (uint64_t)ready * bs >= c->flush_bytes)) count = ready;
         }
         if (count) err = compress_staged(s, c, count, last_n, final_last && count > 0);
      }

      if (!err && !c->frame_pending && c->out_pending && s->avail_out) {
         size_t k = s->avail_out < c->out_pending ? s->avail_out : c->out_pending;
         memcpy(s->next_out, c->out + c->out_pos, k);
         c->out_pos += k;
         c->out_pending -= k;
         s->next_out += k;
         s->avail_out -= k;
         s->total_out += k;
      }

      if (!err && !c->frame_pending && !c->out_pending && (c->state & ST_FINALIZED) && !(c->state & ST_FOOTER_EMITTED)) {
         int f = zultra_frame_encode_footer(c->frame, 16, s->adler, (long long)s->total_in, c->flags);
         if (f < 0)
            err = ZULTRA_ERROR_COMPRESSION;
         else {
            c->state = (c->state | ST_FOOTER_EMITTED) & ~(unsigned)ST_FINALIZED;
            c->frame_pos = 0;
            c->frame_pending = (size_t)f;
         }
      }
      if (!err) drain_frame(s, c);
   } while (!err && s->avail_in && s->avail_out);

   if (err) return err;
   if ((c->state & ST_FOOTER_EMITTED) && !(c->state & ST_STREAM_ENDED) && !c->frame_pending) {
      c->state |= ST_STREAM_ENDED;
      return ZULTRA_STREAM_END;
   }
   return ZULTRA_OK;
}

// ================================================================================================================
// In-memory API (reference src/libzultra.c:576-619)
// ================================================================================================================

extern "C" size_t zultra_memory_bound(size_t nInputSize, const unsigned int nFlags, unsigned int nMaxBlockSize) {
   const size_t bs = clamp_block(nMaxBlockSize);
   return (size_t)zultra_frame_get_header_size(nFlags, NULL, 0) + ((nInputSize + (bs - 1)) / bs) * (1 + 4 + 1) * 64 + nInputSize + 1 +
          (size_t)zultra_frame_get_footer_size(nFlags);
}

// ---- in-memory compression over several device contexts ("lanes") ---------------------------------------------------------------
// The reference's zultra_memory_compress (libzultra.c:601-619) is one stream; here the whole input is at hand, and max-blocks only
// need the 32 KiB in front of them: the input is cut into contiguous *jobs* of max-blocks, job j goes to lane j % lanes — a lane = a
// device context on one of the devices of ZULTRA_HIP_DEVICES / zultra_set_devices, with a host thread of its own that stages the job's
// bytes into the context's pinned memory and runs the batch (upload + kernels). What depends on the bit phase the stream has reached —
// BFINAL / BTYPE bits, stored fallback, bit carry (libzultra.c:327-398,414-436) — is the stitch, and that is done by the calling
// thread job by job in stream order, on the job's own device, as soon as the job's batch is done; the stitched bytes go to pOut
// while the lanes are at their next jobs. Same bytes as the one-stream path: both cut max-blocks at multiples of the block size.
namespace {
struct MemJob {
   size_t block0, nblocks;   // max-blocks [block0, block0 + nblocks) of the input
   int state;                // 0 waiting, 1 batch done, -1 failed, 2 stitched (the lane's context is free again)
};
struct MemLanes {
   const unsigned char *in;
   size_t n_in;
   uint32_t bs;
   size_t total_blocks;
   const void *dict;
   int dict_size;
   std::vector<MemJob> jobs;
   std::vector<zultra_hip_ctx_t *> ctx;
   std::mutex m;
   std::condition_variable cv;
   bool abort;
};

// input bytes -> pinned staging: the copy is what a lane's first job waits for, so large ones are split over a few threads
static void staged_copy(uint8_t *dst, const uint8_t *src, size_t n) {
   const size_t piece = 16u << 20;
   if (n < 2 * piece) {
      memcpy(dst, src, n);
      return;
   }
   const size_t nt = n / piece < 4 ? n / piece : 4;
   std::vector<std::thread> th;
   size_t started = 1;   // pieces 1 .. started - 1 have a thread; nothing thrown here may cross the C ABI (extern "C" callers end in std::terminate)
   try {
      th.reserve(nt);
      for (; started < nt; started++) th.emplace_back([=] { memcpy(dst + n * started / nt, src + n * started / nt, n * (started + 1) / nt - n * started / nt); });
   } catch (...) {
   }
   memcpy(dst, src, n / nt);
   if (started < nt) memcpy(dst + n * started / nt, src + n * started / nt, n - n * started / nt);   // (no thread for the rest: copied here)
   for (auto &t : th) t.join();
}

static void mem_lane_body(MemLanes *M, size_t lane);
static void mem_lane_thread(MemLanes *M, size_t lane) {
   try {
      mem_lane_body(M, lane);
   } catch (...) {   // (std::bad_alloc from the block list: the batch fails, the process does not)
      {
         std::lock_guard<std::mutex> lk(M->m);
         for (size_t j = lane; j < M->jobs.size(); j += M->ctx.size())
            if (M->jobs[j].state == 0) M->jobs[j].state = -1;
         M->abort = true;
      }
      M->cv.notify_all();
   }
}
static void mem_lane_body(MemLanes *M, size_t lane) {
   zultra_hip_ctx_t *c = M->ctx[lane];
   std::vector<zultra_hip_block_t> blocks;
   for (size_t j = lane; j < M->jobs.size(); j += M->ctx.size()) {
      MemJob &J = M->jobs[j];
      if (j >= M->ctx.size()) {   // the lane's previous job must have left the context (stitched and read out)
         std::unique_lock<std::mutex> lk(M->m);
         M->cv.wait(lk, [&] { return M->abort || M->jobs[j - M->ctx.size()].state == 2; });
      }
      {
         std::lock_guard<std::mutex> lk(M->m);
         if (M->abort) return;
      }
      const size_t first = J.block0 * M->bs, last = first + J.nblocks * M->bs < M->n_in ? first + J.nblocks * M->bs : M->n_in;
      // history of the job's first max-block: the 32 KiB of input in front of it, or the preset dictionary (libzultra.c:250-253)
      size_t hist = first < (size_t)HISTORY_SIZE ? first : (size_t)HISTORY_SIZE;
      const uint8_t *hsrc = M->in + first - hist;
      if (J.block0 == 0 && M->dict && M->dict_size > 0) {
         hist = M->dict_size > HISTORY_SIZE ? (size_t)HISTORY_SIZE : (size_t)M->dict_size;
         hsrc = (const uint8_t *)M->dict + (M->dict_size - (int)hist);
      }
      // The job's windows are contiguous in the caller's buffer — unless its history is the preset dictionary — and are handed over
      // as they are: the device layer stages and uploads them run by run (zultra_hip_compress_blocks, data_on_device == 2).
      const bool in_place = hsrc + hist == M->in + first;
      uint8_t *stage = in_place ? (uint8_t *)(M->in + first - hist) : (uint8_t *)zultra_hip_staging(c, 0, (size_t)HISTORY_SIZE + J.nblocks * M->bs);
      int ok = stage != NULL;
      if (ok) {
         if (!in_place) {
            memcpy(stage, hsrc, hist);
            staged_copy(stage + hist, M->in + first, last - first);
         }
         blocks.resize(J.nblocks);
         for (size_t b = 0; b < J.nblocks; b++) {
            const uint32_t prev = b == 0 ? (uint32_t)hist : (uint32_t)HISTORY_SIZE;
            const size_t at = first + b * M->bs;
            blocks[b].win_off = hist + b * M->bs - prev;
            blocks[b].prev = prev;
            blocks[b].n = (uint32_t)(at + M->bs <= M->n_in ? M->bs : M->n_in - at);
         }
         // the stream's first job starts at phase 0: its stitch goes out with its kernels (the later jobs' phases are known when the jobs before them are stitched)
         if (j == 0) (void)zultra_hip_stitch_with_batch(c, 1, 0, M->jobs.size() == 1 ? (int)J.nblocks - 1 : -1);
         ok = zultra_hip_compress_blocks(c, stage, hist + (last - first), in_place ? 2 : 0, blocks.data(), (uint32_t)J.nblocks) > 0;
      }
      {
         std::lock_guard<std::mutex> lk(M->m);
         J.state = ok ? 1 : -1;
         if (!ok) M->abort = true;
      }
      M->cv.notify_all();
      if (!ok) return;
   }
}
}   // namespace

static size_t memory_compress_lanes(const unsigned char *pIn, size_t nIn, unsigned char *pOut, size_t nOutCap, const unsigned int nFlags, uint32_t bs,
                                    const void *pDict, int nDictSize, const std::vector<int> &devices) {
   MemLanes M;
   M.in = pIn;
   M.n_in = nIn;
   M.bs = bs;
   M.total_blocks = (nIn + bs - 1) / bs;
   M.dict = pDict;
   M.dict_size = nDictSize;
   M.abort = false;
   // jobs: as few as there are lanes when the input fits that way, else batches of the size a one-stream call would use
   size_t per = (M.total_blocks + devices.size() - 1) / devices.size();
   const uint64_t budget = 24ull << 30;
   if (per > 8192) per = 8192;
   for (int dv : devices)   // (the listed devices need not be alike: a job must fit the smallest budget)
      while (per > 1 && (uint64_t)zultra_hip_context_bytes_on(dv, bs, (uint32_t)per) > budget) per -= (per + 7) / 8;
   std::vector<std::thread> threads;
   size_t lanes = 0;
   try {
      for (size_t b = 0; b < M.total_blocks; b += per) M.jobs.push_back(MemJob{b, M.total_blocks - b < per ? M.total_blocks - b : per, 0});
      const size_t want = M.jobs.size() < devices.size() ? M.jobs.size() : devices.size();
      // a lane whose context cannot be had (a second large context on a device listed twice, a busy or smaller device) is left out: the
      // jobs go round the lanes that exist; none at all is the only failure
      for (size_t l = 0; l < want; l++) {
         zultra_hip_ctx_t *c = ctx_acquire_on(devices[l], bs, (uint32_t)per);
         if (c) M.ctx.push_back(c);
      }
      lanes = M.ctx.size();
      if (!lanes) return (size_t)-1;
      // (one lane with one job — a call on a few max-blocks, where 50 us of thread start-up and wake-ups count — runs on the calling thread)
      if (lanes == 1 && M.jobs.size() == 1)
         mem_lane_thread(&M, 0);
      else {
         threads.reserve(lanes);
         for (size_t l = 0; l < lanes; l++) threads.emplace_back(mem_lane_thread, &M, l);
      }
   } catch (...) {   // std::bad_alloc / std::system_error must not cross the C ABI
      {
         std::lock_guard<std::mutex> lk(M.m);
         M.abort = true;
      }
      M.cv.notify_all();
      for (auto &t : threads) t.join();
      for (auto *k : M.ctx) ctx_release(k);
      return (size_t)-1;
   }

   size_t w = 0;
   bool fail = false;
   {
      unsigned char hdr[16];
      const int h = zultra_frame_encode_header(hdr, 16, nFlags, pDict, nDictSize);
      if (h < 0 || (size_t)h > nOutCap)
         fail = true;
      else {
         memcpy(pOut, hdr, (size_t)h);
         w = (size_t)h;
      }
   }
   zultra_frame_checksum_t sum = zultra_frame_init_checksum(nFlags);
   zultra_hip_bitstate_t bit;
   bit.acc = bit.nacc = 0;
   std::vector<uint32_t> parts;
   try {
      parts.reserve(2 * per);
   } catch (...) {
      fail = true;
   }
   for (size_t j = 0; j < M.jobs.size() && !fail; j++) {
      MemJob &J = M.jobs[j];
      {
         std::unique_lock<std::mutex> lk(M.m);
         M.cv.wait(lk, [&] { return J.state != 0 || M.abort; });
         if (J.state != 1) fail = true;
      }
      zultra_hip_ctx_t *c = M.ctx[j % lanes];
      if (!fail) {
         // checksum once per max-block over its bytes (libzultra.c:279): computed on the device next to the compression, folded here
         const size_t first = J.block0 * bs;
         if (nFlags & ZULTRA_FLAG_GZIP_FRAMING) {
            parts.resize(J.nblocks);
            fail = zultra_hip_block_crc32(c, parts.data()) != (int)J.nblocks;
            for (size_t b = 0; b < J.nblocks && !fail; b++) {
               const size_t at = first + b * bs;
               sum = zultra_crc32_append(sum, parts[b], at + bs <= nIn ? bs : nIn - at);
            }
         }
         else if (nFlags & ZULTRA_FLAG_ZLIB_FRAMING) {
            parts.resize(2 * J.nblocks);
            fail = zultra_hip_block_adler32(c, parts.data()) != (int)J.nblocks;
            for (size_t b = 0; b < J.nblocks && !fail; b++) {
               const size_t at = first + b * bs;
               sum = zultra_adler32_append(sum, parts[2 * b], parts[2 * b + 1], at + bs <= nIn ? bs : nIn - at);
            }
         }
      }
      if (!fail) {
         const bool last = j + 1 == M.jobs.size();
         uint64_t end_bit = 0;
         const uint32_t phase = bit.nacc;
         const uint32_t pending = bit.acc & ((1u << phase) - 1);
         const int rc = zultra_hip_stitch_device(c, &bit, last ? (int)J.nblocks - 1 : -1, &end_bit);
         const size_t total = (size_t)((end_bit + 7) >> 3);
         uint8_t *stage = rc == 0 ? (uint8_t *)zultra_hip_staging(c, 1, total + 16) : NULL;
         if (rc != 0 || !stage || zultra_hip_stream_read(c, stage, 0, total) != 0)
            fail = true;
         else {
            stage[0] |= (uint8_t)pending;
            // whole bytes go out; a trailing partial byte stays pending for the next job (the last job's is zero padded, libzultra.c:414-417)
            const size_t whole = last ? total : (size_t)(end_bit >> 3);
            if (w + whole > nOutCap)
               fail = true;
            else {
               memcpy(pOut + w, stage, whole);
               w += whole;
               bit.acc = (!last && (end_bit & 7)) ? stage[whole] : 0;
               if (last) bit.nacc = 0;
            }
         }
      }
      {
         std::lock_guard<std::mutex> lk(M.m);
         J.state = 2;
         if (fail) M.abort = true;
      }
      M.cv.notify_all();
      if (fail) break;
   }
   if (fail) {
      {
         std::lock_guard<std::mutex> lk(M.m);
         M.abort = true;
      }
      M.cv.notify_all();
   }
   for (auto &t : threads) t.join();
   for (auto *k : M.ctx) ctx_release(k);
   if (fail) return (size_t)-1;
   unsigned char ftr[16];
   const int f = zultra_frame_encode_footer(ftr, 16, sum, (long long)nIn, nFlags);
   if (f < 0 || w + (size_t)f > nOutCap) return (size_t)-1;
   memcpy(pOut + w, ftr, (size_t)f);
   return w + (size_t)f;
}

extern "C" size_t zultra_memory_compress_dict(const unsigned char *pIn, size_t nIn, unsigned char *pOut, size_t nOutCap,
                                              const unsigned int nFlags, unsigned int nMaxBlockSize, const void *pDict, int nDictSize) {
   zultra_stream_t strm;
   memset(&strm, 0, sizeof(strm));
   const uint32_t bs = clamp_block(nMaxBlockSize);
   // The whole input is at hand: shards of max-blocks over the devices asked for (ZULTRA_HIP_DEVICES / zultra_set_devices; "0,0" = two
   // contexts on device 0), at least two max-blocks per lane — or one lane on the one device, its batch staged and uploaded run by run.
   // (ZULTRA_HIP_MEMORY_LANES=0: through the stream API instead, as the reference does, libzultra.c:601-619.)
   {
      std::vector<int> devs = zh_pick_devices();
      const char *e = getenv("ZULTRA_HIP_MEMORY_LANES");
      const int lanes = e ? atoi(e) : 1;
      if (devs.empty())
         for (int l = 0; l < lanes; l++) devs.push_back(zh_pick_device());
      while (devs.size() > 1 && (nIn + bs - 1) / bs < 2 * devs.size()) devs.pop_back();
      if (!devs.empty() && pIn && nIn) return memory_compress_lanes(pIn, nIn, pOut, nOutCap, nFlags, bs, pDict, nDictSize, devs);
   }
   // the whole input is at hand: size the device batch to it
   uint64_t want = (nIn + bs - 1) / bs;
   if (want < 1) want = 1;
   if (want > 8192) want = 8192;
   if (stream_init_sized(&strm, nFlags, nMaxBlockSize, (uint32_t)want) != ZULTRA_OK) return (size_t)-1;
   if (pDict && nDictSize > 0 && zultra_stream_set_dictionary(&strm, pDict, nDictSize) != ZULTRA_OK) {
      zultra_stream_end(&strm);
      return (size_t)-1;
   }
   strm.next_in = pIn;
   strm.avail_in = nIn;
   strm.next_out = pOut;
   strm.avail_out = nOutCap;
   zultra_status_t st = zultra_stream_compress(&strm, ZULTRA_FINALIZE);
   zultra_stream_end(&strm);
   if (st != ZULTRA_STREAM_END) return (size_t)-1;
   return nOutCap - strm.avail_out;
}

extern "C" size_t zultra_memory_compress(const unsigned char *pIn, size_t nIn, unsigned char *pOut, size_t nOutCap,
                                         const unsigned int nFlags, unsigned int nMaxBlockSize) {
   return zultra_memory_compress_dict(pIn, nIn, pOut, nOutCap, nFlags, nMaxBlockSize, NULL, 0);
}
