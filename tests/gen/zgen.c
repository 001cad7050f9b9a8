/* tests/gen/zgen.c — TEST / BENCH INFRASTRUCTURE: deterministic bulk generators for the BASELINE.json configurations that
 * are too large for numpy. Own PRNG (splitmix64), no libc rand(); every unit (segment, file) is seeded by
 * its global index, so any rank can generate any range of the one global stream independently and in parallel.
 *
 *   zgen_mixed       configuration 4: the "synthetic mixed-entropy corpus" — 1 MiB segments cycling through the parameter
 *                    grid of the reference's self-test (tool/zultra.c:529-534: 12 alphabet sizes x match probabilities
 *                    0, 0.1, ..., 0.9, 0.995), built the way its generate_compressible_data builds a buffer
 *                    (tool/zultra.c:425-463: literal runs of 0..127 bytes, matches of 3..1026 bytes at an offset uniform in
 *                    the history), plus every 16th segment incompressible noise (stored fallback) and every 16th a single
 *                    repeated byte. "History" is the last 128 KiB of the segment: the self-test's buffers are at most
 *                    4 x HISTORY_SIZE bytes (tool/zultra.c:533), so that is the range its offsets are uniform in.
 *   zgen_json_files  configuration 5: N independent JSON-like inputs of `file_size` bytes each (records
 *                    {"id":..,"user":"..","ts":..,"tags":[..],"v":..}\n from a 64-name list, truncated at file_size).
 */
#include <stdint.h>
#include <string.h>

#define SEG (1u << 20)

static inline uint64_t splitmix64(uint64_t *s) {
   uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
   z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
   z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
   return z ^ (z >> 31);
}

/* a stream of 32-bit draws, two per splitmix64 step */
typedef struct {
   uint64_t s, cur;
   int have;
} rng_t;
static inline uint32_t draw(rng_t *r) {
   if (r->have) {
      r->have = 0;
      return (uint32_t)(r->cur >> 32);
   }
   r->cur = splitmix64(&r->s);
   r->have = 1;
   return (uint32_t)r->cur;
}

static const int kAlphabets[12] = {1, 2, 3, 15, 30, 56, 96, 137, 178, 191, 255, 256};           /* tool/zultra.c:534 */
static const int kProb1023[11] = {0, 102, 204, 306, 409, 511, 613, 716, 818, 920, 1017};         /* (int)(p * 1023.0f) */

static void gen_selftest_like(uint8_t *buf, size_t n, rng_t *r, int nlit, int prob) {
   size_t i = 0;
   if (!n) return;
   buf[i++] = (uint8_t)(draw(r) % (uint32_t)nlit);
   while (i < n) {
      if ((int)(draw(r) & 1023u) >= prob) {
         size_t cnt = draw(r) & 127u;
         if (cnt > n - i) cnt = n - i;
         while (cnt--) buf[i++] = (uint8_t)(draw(r) % (uint32_t)nlit);
      }
      else {
         size_t len = 3 + (draw(r) & 1023u), hist = i < 131072 ? i : 131072, off;
         if (len > n - i) len = n - i;
         if (len > hist) len = hist;
         off = len < hist ? 1 + draw(r) % (uint32_t)(hist - len) : hist;   /* 1 .. hist-len (the reference's 0 copies garbage) */
         while (len--) {
            buf[i] = buf[i - off];
            i++;
         }
      }
   }
}

void zgen_mixed_segment(uint8_t *out, uint64_t k, uint64_t seed) {
   rng_t r;
   r.s = seed ^ (k * 0xD1B54A32D192ED03ull);
   r.have = 0;
   (void)splitmix64(&r.s);
   const unsigned sel = (unsigned)(k & 15u);
   if (sel == 14) {
      for (size_t i = 0; i < SEG; i += 8) {
         const uint64_t v = splitmix64(&r.s);
         memcpy(out + i, &v, 8);
      }
   }
   else if (sel == 15)
      memset(out, (int)(draw(&r) & 255u), SEG);
   else
      gen_selftest_like(out, SEG, &r, kAlphabets[k % 12], kProb1023[(k * 7) % 11]);
}

void zgen_mixed(uint8_t *out, uint64_t first_segment, uint64_t nsegments, uint64_t seed) {
#pragma omp parallel for schedule(dynamic, 1)
   for (int64_t k = 0; k < (int64_t)nsegments; k++) zgen_mixed_segment(out + (size_t)k * SEG, first_segment + (uint64_t)k, seed);
}

/* ---- JSON-like small files ------------------------------------------------------------------------------------------- */
static const char *kNames[64] = {
   "ana", "tobias", "lee", "marta", "ishaan", "chen", "olu", "freya", "sam", "noor", "pavel", "yuki", "diego", "amara", "jon", "lucia",
   "kofi", "mei", "rafael", "sven", "aisha", "tomas", "ines", "bao", "greta", "malik", "sofia", "ravi", "elena", "hugo", "zara", "omar",
   "nina", "felix", "priya", "ivan", "leila", "max", "hana", "arjun", "clara", "mateo", "anya", "david", "fatima", "oscar", "linh", "erik",
   "carmen", "kenji", "ruth", "pablo", "mira", "anton", "sara", "jamal", "vera", "luis", "emma", "tariq", "olga", "nico", "dana", "theo"};

static inline char *put_str(char *p, const char *s) {
   while (*s) *p++ = *s++;
   return p;
}
static inline char *put_u64(char *p, uint64_t v) {
   char tmp[24];
   int n = 0;
   do {
      tmp[n++] = (char)('0' + v % 10);
      v /= 10;
   } while (v);
   while (n) *p++ = tmp[--n];
   return p;
}

void zgen_json_file(uint8_t *out, uint32_t file_size, uint64_t index, uint64_t seed) {
   rng_t r;
   r.s = seed ^ (index * 0xA24BAED4963EE407ull);
   r.have = 0;
   (void)splitmix64(&r.s);
   uint64_t rid = draw(&r) & 0x3fffffffu;
   uint32_t pos = 0;
   char rec[512];
   while (pos < file_size) {
      char *p = rec;
      p = put_str(p, "{\"id\":");
      p = put_u64(p, rid);
      p = put_str(p, ",\"user\":\"");
      p = put_str(p, kNames[draw(&r) & 63u]);
      p = put_str(p, "\",\"ts\":");
      p = put_u64(p, 1700000000ull + (draw(&r) & 0xffffffu));
      p = put_str(p, ",\"tags\":[");
      const uint32_t ntags = draw(&r) % 5u;
      for (uint32_t t = 0; t < ntags; t++) {
         if (t) *p++ = ',';
         *p++ = '"';
         p = put_str(p, kNames[draw(&r) & 63u]);
         *p++ = '"';
      }
      p = put_str(p, "],\"v\":");
      const uint32_t v = draw(&r) % 10000000u;   /* 0.0000 .. 999.9999 */
      p = put_u64(p, v / 10000u);
      *p++ = '.';
      {
         uint32_t f = v % 10000u;
         *p++ = (char)('0' + f / 1000u);
         *p++ = (char)('0' + f / 100u % 10u);
         *p++ = (char)('0' + f / 10u % 10u);
         *p++ = (char)('0' + f % 10u);
      }
      *p++ = '}';
      *p++ = '\n';
      uint32_t len = (uint32_t)(p - rec);
      if (len > file_size - pos) len = file_size - pos;
      memcpy(out + pos, rec, len);
      pos += len;
      rid += 1 + draw(&r) % 3u;
   }
}

void zgen_json_files(uint8_t *out, uint64_t first_file, uint64_t nfiles, uint32_t file_size, uint64_t seed) {
#pragma omp parallel for schedule(static, 256)
   for (int64_t k = 0; k < (int64_t)nfiles; k++) zgen_json_file(out + (size_t)k * file_size, file_size, first_file + (uint64_t)k, seed);
}
