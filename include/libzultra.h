/*
 * libzultra.h — the drop-in public API of the MI355X build.
 *
 * Call-compatible with the reference's src/libzultra.h (names, argument meaning, status values, struct layout:
 * :49-61 status enum, :64-75 flags/defaults, :78-93 zultra_stream_t, :104-157 functions) and, for the helpers the
 * reference's CLI links against, with src/frame.h:65-119 and src/dictionary.h:60-67. A program written against the
 * reference relinks against libzultra_amd.so unchanged; the bytes it gets back are identical.
 *
 * Behavioural contract kept from the reference:
 *  - max-blocks are cut at exactly nMaxBlockSize bytes whatever the chunking of the calls (libzultra.c:259-269);
 *  - a full max-block is compressed only once more input is seen or ZULTRA_FINALIZE is given (:269);
 *  - ZULTRA_OK = "call again", ZULTRA_STREAM_END once after the footer is consumed, then ZULTRA_ERROR_COMPRESSION;
 *  - an empty input is never finalized (:275), so zultra_memory_compress(…, 0 bytes) returns (size_t)-1;
 *  - zultra_memory_compress returns (size_t)-1 if the output buffer cannot take the whole stream.
 * Difference in *how*: when a call hands over several max-blocks at once they are compressed as one device batch.
 * The per-block work runs on the GPU only; without a usable HIP device zultra_stream_init fails with
 * ZULTRA_ERROR_MEMORY... never with a silent CPU path.
 */
#ifndef LIBZULTRA_AMD_H
#define LIBZULTRA_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- format constants (reference src/format.h:37-50) ---- */
#define MIN_MATCH_SIZE 3
#define MAX_MATCH_SIZE 258
#define MIN_OFFSET 1
#define MAX_OFFSET 32768
#define HISTORY_SIZE 0x8000

/* ---- framing helpers (reference src/frame.h) ---- */
#define ZULTRA_HEADER_SIZE 4
#define ZULTRA_FRAME_SIZE 4
#define ZULTRA_FOOTER_SIZE 8
#define ZULTRA_ENCODE_ERR (-1)

typedef unsigned int zultra_frame_checksum_t;

int zultra_frame_get_header_size(const unsigned int nFlags, const void *pDictionaryData, const int nDictionarySize);
int zultra_frame_encode_header(unsigned char *pFrameData, const int nMaxFrameDataSize, const unsigned int nFlags,
                               const void *pDictionaryData, const int nDictionarySize);
zultra_frame_checksum_t zultra_frame_init_checksum(const unsigned int nFlags);
zultra_frame_checksum_t zultra_frame_update_checksum(zultra_frame_checksum_t nChecksum, const void *pData, size_t nDataSize,
                                                     const unsigned int nFlags);
int zultra_frame_get_footer_size(const unsigned int nFlags);
int zultra_frame_encode_footer(unsigned char *pFrameData, const int nMaxFrameDataSize, const zultra_frame_checksum_t nChecksum,
                               long long nOriginalSize, const unsigned int nFlags);

/* ---- status, flags ---- */
typedef enum _zultra_stream_e
#if defined(__cplusplus) && __cplusplus > 199711L
   : int
#endif
{
   ZULTRA_OK = 0,                 /* progress made; call again for more */
   ZULTRA_STREAM_END,             /* footer fully delivered */
   ZULTRA_ERROR_SRC = -1,
   ZULTRA_ERROR_DST = -2,
   ZULTRA_ERROR_DICTIONARY = -3,
   ZULTRA_ERROR_MEMORY = -4,
   ZULTRA_ERROR_COMPRESSION = -5,
} zultra_status_t;

#define ZULTRA_FLAG_DEFLATE_FRAMING 0   /* raw deflate */
#define ZULTRA_FLAG_ZLIB_FRAMING 1      /* RFC 1950 */
#define ZULTRA_FLAG_GZIP_FRAMING 2      /* RFC 1952 */

#define ZULTRA_CONTINUE 0
#define ZULTRA_FINALIZE 1

#define ZULTRA_DEFAULT_MAX_BLOCK_SIZE 1048576

/* ---- dictionary helpers (reference src/dictionary.h) ---- */
zultra_status_t zultra_dictionary_load(const char *pszDictionaryFilename, void **ppDictionaryData, int *pDictionaryDataSize);
void zultra_dictionary_free(void **ppDictionaryData);

/* ---- streaming API ---- */
typedef struct _zultra_compressor_s zultra_compressor_t;

typedef struct _zultra_stream_s {
   const unsigned char *next_in;
   size_t avail_in;
   unsigned long long total_in;

   unsigned char *next_out;
   size_t avail_out;
   unsigned long long total_out;

   void *(*zalloc)(void *opaque, unsigned int items, unsigned int size);
   void (*zfree)(void *opaque, void *address);
   void *opaque;

   zultra_compressor_t *state;
   zultra_frame_checksum_t adler;
} zultra_stream_t;

zultra_status_t zultra_stream_init(zultra_stream_t *pStream, const unsigned int nFlags, unsigned int nMaxBlockSize);
zultra_status_t zultra_stream_set_dictionary(zultra_stream_t *pStream, const void *pDictionaryData, const int nDictionaryDataSize);
zultra_status_t zultra_stream_compress(zultra_stream_t *pStream, const int nDoFinalize);
void zultra_stream_end(zultra_stream_t *pStream);

/* ---- in-memory API ---- */
size_t zultra_memory_bound(size_t nInputSize, const unsigned int nFlags, unsigned int nMaxBlockSize);
size_t zultra_memory_compress(const unsigned char *pInputData, size_t nInputSize, unsigned char *pOutBuffer,
                              size_t nMaxOutBufferSize, const unsigned int nFlags, unsigned int nMaxBlockSize);

/* ---- additions of this build (not in the reference) ---- */
/* As zultra_memory_compress with a preset dictionary (what zultra_stream_set_dictionary + one FINALIZE call do). */
size_t zultra_memory_compress_dict(const unsigned char *pInputData, size_t nInputSize, unsigned char *pOutBuffer,
                                   size_t nMaxOutBufferSize, const unsigned int nFlags, unsigned int nMaxBlockSize,
                                   const void *pDictionaryData, int nDictionaryDataSize);
/* Device the library runs on (default 0, or env ZULTRA_HIP_DEVICE); call before the first stream is created. */
void zultra_set_device(int nDevice);
/* Devices zultra_memory_compress spreads an input over (default: env ZULTRA_HIP_DEVICES="0,1,...", else the one device above). The
 * max-blocks of the input are cut into contiguous shards, one host thread and one device context per entry compress them side by
 * side, and the shards' bit strings are stitched in stream order on their own devices: the output bytes do not depend on the list.
 * An entry may repeat a device ("0,0": two contexts on device 0, the transfers of one shard next to the kernels of the other;
 * without a list ZULTRA_HIP_MEMORY_LANES=n does the same on the one device, default 1, and 0 sends the call through the stream API).
 * Returns nDevices, or -1 when an entry is not a visible device (the list is then left empty). nDevices = 0 clears the list. */
int zultra_set_devices(const int *pDevices, int nDevices);
/* zultra_stream_end frees everything the stream owns through zfree, like the reference (libzultra.c:521-565) — except that the
 * device context (device memory and pinned staging) of a finished stream is kept for the next stream of the same geometry:
 * creating one takes ~45 device allocations. zultra_release_cached_contexts() destroys the contexts kept that way (at most
 * two); with the environment variable ZULTRA_HIP_CACHE=0 none is ever kept and zultra_stream_end releases the device memory
 * itself. Returns the number of contexts destroyed. */
int zultra_release_cached_contexts(void);

#ifdef __cplusplus
}
#endif
#endif /* LIBZULTRA_AMD_H */
