/*
 * zultra_hip.h — C ABI of the MI355X (gfx950) device layer: batches of independent max-blocks in, per-sub-block
 * deflate bit strings out. Plain pointers and sizes only.
 *
 * This is the inner boundary of the reference: the five calls that zultra_stream_compress makes per max-block
 * (reference src/libzultra.c:287-343)
 *      zultra_build_suffix_array(win, prev+n)            :287   \
 *      zultra_skip_matches(0, prev)                      :291    >  stage 1  (zultra_hip: match rows)
 *      zultra_find_all_matches(prev, prev+n)             :293   /
 *      zultra_block_split(win, prev, n, 64, splits)      :303      stage 2  (zultra_hip: sub-block boundaries)
 *      zultra_block_prepare_cost_evaluation / _evaluate_static_cost / 2x _estimate_dynamic_codelens /
 *      _evaluate_dynamic_cost                            :317-321  stage 3a (static vs dynamic)
 *      zultra_block_deflate(win, start, size, isDynamic) :343      stage 3b (bits of the sub-block body)
 * are replaced by ONE batched call, zultra_hip_compress_blocks(), which runs them for many max-blocks at once
 * (max-blocks are independent given their 32 KiB of raw history). What stays on the caller's side
 * is exactly what depends on the running bit phase of the output stream: BFINAL/BTYPE bits, the stored-block
 * fallback and the bit carry between max-blocks (libzultra.c:327-398, 414-436) — done by zultra_hip_stitch().
 *
 * There is no CPU fallback: every entry point fails (NULL / negative) when no HIP device is usable.
 */
#ifndef ZULTRA_HIP_H
#define ZULTRA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct zultra_hip_ctx_s zultra_hip_ctx_t;

/* One max-block: window = data + win_off, `prev` bytes of history (0..32768) followed by `n` block bytes
 * (1..max_block_size). Mirrors the (pInWindow, nPreviousBlockSize, nInDataSize) triple of libzultra.c:287-303. */
typedef struct zultra_hip_block_s {
   uint64_t win_off;
   uint32_t prev;
   uint32_t n;
} zultra_hip_block_t;

/* One sub-block result, in stream order. Mirrors what libzultra.c:314-347 needs to frame it. */
typedef struct zultra_hip_subblock_s {
   uint32_t block;          /* index of the max-block in the batch */
   uint32_t start;          /* offset of the sub-block inside the max-block (nInStart, libzultra.c:297) */
   uint32_t size;           /* nBlockSize, libzultra.c:314 */
   uint32_t is_dynamic;     /* nIsDynamic, libzultra.c:323-324 */
   int32_t static_cost;     /* nStaticCost  (blockdeflate.c:538) */
   int32_t dynamic_cost;    /* nDynamicCost (blockdeflate.c:577) */
   uint32_t failed;         /* non-zero: zultra_block_deflate would have failed / outgrown its buffer -> store it */
   uint32_t reserved;
   uint64_t nbits;          /* exact bit length of the body (after the 3 header bits) */
   uint64_t bits_off;       /* byte offset of the body's bit string (bit phase 0) in the payload */
} zultra_hip_subblock_t;

/* Timings of the last batch, milliseconds, measured with HIP events on the context's stream. */
typedef struct zultra_hip_timing_s {
   float h2d_ms, matchfinder_ms, tokenize_split_ms, encode_ms, d2h_ms, total_ms;
   float group_ms, frontier_ms, stitch_ms;
   float init_ms, parse_ms, build_ms, post_ms, emit_ms;   /* parts of encode_ms: planning + zh_sb_init + zh_list_huge, 4 x the parse kernels, 4 x zh_sb_build, zh_post_tasks, zh_emit_tasks
                                                              (a batch runs as staggered runs on streams of their own: each figure is the SUM over the runs, which overlap in wall time) */
   float head_ms;                                          /* wall time from the batch's first launch to the end of the FIRST run's matchfinder: nothing of the batch can overlap it */
   float tail_ms;                                          /* wall time from the end of the LAST run's matchfinder to the batch's last completion: token chain, splitter, four parse / rebuild
                                                              rounds, literalisation, emission — it does not shrink with the batch (the strong-scaling floor) */
} zultra_hip_timing_t;

/* Number of usable HIP devices (0 if none). */
int zultra_hip_device_count(void);

/* Device self-check of the wave64 primitives the kernels rely on (DPP row reductions, readlane, prefix sums)
 * against plain LDS arithmetic. Returns 0 when they agree, a positive count of mismatches, negative on HIP errors. */
int zultra_hip_selftest(void);

/* Diagnostics: a streaming 4-byte-per-lane copy of `nbytes`, used to calibrate the rocprofv3 FETCH_SIZE / WRITE_SIZE
 * counters for this access width (profiles/, tools/pmc_traffic.py). Returns 0 on success. */
int zultra_hip_traffic_probe(size_t nbytes);

/* The second roofline denominator: measured bandwidth of a 16-byte-per-lane streaming copy of `nbytes`,
 * `iters` launches timed with HIP events on the null stream. Returns GB/s (bytes read + bytes written), negative on errors. */
double zultra_hip_copy_bandwidth(size_t nbytes, int iters);

/* Create a context on `device` able to take batches of up to max_blocks max-blocks of up to max_block_size bytes
 * (clamped like libzultra.c:87-92). All device memory is allocated here and released by zultra_hip_destroy
 * (the reference allocates in zultra_stream_init and frees in zultra_stream_end, libzultra.c:82-166,521-565).
 * Returns NULL on failure (no device, out of memory). */
zultra_hip_ctx_t *zultra_hip_create(int device, uint32_t max_block_size, uint32_t max_blocks);
void zultra_hip_destroy(zultra_hip_ctx_t *ctx);
const char *zultra_hip_last_error(const zultra_hip_ctx_t *ctx);

/* What a context was created with, and the device memory it holds (any pointer may be NULL). */
void zultra_hip_ctx_info(const zultra_hip_ctx_t *ctx, int *device, uint32_t *max_block_size, uint32_t *max_blocks, size_t *device_bytes);
/* Device bytes zultra_hip_create(device, max_block_size, max_blocks) will allocate, computed from the same layout without
 * allocating: callers size their batches against the memory they want to spend. */
size_t zultra_hip_context_bytes_on(int device, uint32_t max_block_size, uint32_t max_blocks);
size_t zultra_hip_context_bytes(uint32_t max_block_size, uint32_t max_blocks);   /* ... on the calling thread's current device */

/* Pinned host staging owned by the context (which: 0 = input side, 1 = output side), at least `size` bytes, valid until
 * the context is destroyed or a larger size is requested. The streaming API stages caller data through these: copies
 * from pageable memory run at a fraction of the PCIe rate. Returns NULL on failure. */
void *zultra_hip_staging(zultra_hip_ctx_t *ctx, int which, size_t size);

/* Bytes of input the context can hold per batch (max_blocks * max_block_size + 32768 of leading history). */
size_t zultra_hip_data_capacity(const zultra_hip_ctx_t *ctx);

/* Run stages 1-3 for a batch. `data` holds every window of the batch: host memory if data_on_device == 0 (uploaded in one piece:
 * best from pinned memory, e.g. zultra_hip_staging), a device pointer valid on the context's device if 1 (read in place, nothing
 * is copied), pageable host memory if 2 (staged through the context's pinned buffer and uploaded run by run, each run's bytes
 * under the kernels of the run before it). Returns the number of sub-blocks (>0) or a negative error. Results stay in the context
 * until the next batch. */
int zultra_hip_compress_blocks(zultra_hip_ctx_t *ctx, const void *data, size_t data_size, int data_on_device,
                               const zultra_hip_block_t *blocks, uint32_t nblocks);

/* Results of the last batch as host pointers owned by the context (valid until the next batch / destroy). */
const zultra_hip_subblock_t *zultra_hip_subblocks(const zultra_hip_ctx_t *ctx, uint32_t *count);
const uint8_t *zultra_hip_payload(const zultra_hip_ctx_t *ctx, size_t *size);
void zultra_hip_last_timing(const zultra_hip_ctx_t *ctx, zultra_hip_timing_t *t);

/* Shape of the last batch: how the parse was decomposed. huge_* = tasks (and their positions) that hold a
 * barrier-free run too long for the four-recurrences-per-wave kernel and went to the single-chain kernel instead. */
typedef struct zultra_hip_stats_s {
   uint64_t positions, huge_positions;
   uint32_t blocks, subblocks, tasks, huge_tasks;
   uint32_t cut_tasks, cut_segments;   /* chain tasks parsed as speculative segments, and their segments */
   uint32_t cut_redone;                /* segments whose speculated costs did not match and were parsed again (sum over the passes) */
   uint32_t runs;                      /* staggered runs the batch was cut into (ZULTRA_HIP_STREAMS, fewer for small batches) */
   uint32_t settled_passes;            /* (sub-block, parse pass) pairs not run: the sub-block's code lengths had reached a fixed point of the loop at
                                          blockdeflate.c:874-901, every further pass would have reproduced the parse it has */
   uint32_t settled_kib;               /* ... the input they cover, in KiB (of 4 x positions / 1024 for the four passes of everything) */
   uint32_t cut_demoted;               /* cut tasks that had several failed cuts in one pass and were parsed as one chain in the passes left */
   uint32_t runs_without_chain_kernels;   /* runs of the last batch enqueued without zh_parse_chain: their counterparts in the context's batch before it listed no chains (round 6) */
   uint32_t batches_rerun;                /* batches of this context run a second time because such a run listed chains after all (since the context was created) */
} zultra_hip_stats_t;
void zultra_hip_last_stats(const zultra_hip_ctx_t *ctx, zultra_hip_stats_t *out);

/* Diagnostics: with ZULTRA_HIP_CHAIN_TRACE=1 in the environment when the context is created, the chain kernels record per work
 * ticket {positions, start, end} on the device's 100 MHz clock; out = uint64[4 runs][4 passes][*slots][3]. Returns 0, or -1 when
 * tracing is off. */
int zultra_hip_chain_trace(zultra_hip_ctx_t *ctx, uint64_t *out, uint32_t *slots);

/* Diagnostics: the tasks of run `run` of the last batch that were cut into speculative segments, four words each: task index in
 * the run, number of segments K | (segment length / 32) << 12, first cost-vector slot, segment completions | (cuts that failed their check and
 * were parsed again) << 16, both counted over the four passes. Returns the number of entries written (at most cap), -1 on error. */
int zultra_hip_cut_tasks(zultra_hip_ctx_t *ctx, uint32_t run, uint32_t *out, uint32_t cap);

/* Stage outputs of the last batch, for parity tests (copied device -> host on request).
 *   matches: n*8 entries {u16 length, u16 offset} of max-block `block`          (match[], private.h:59-62,97)
 *   splits : absolute window offsets of the sub-block ends, last = prev+n; returns the count (nSplitOffset, libzultra.c:299)
 *   parse  : n entries {u16 length, u16 offset}, the final parse                (best_match[], private.h:98) */
int zultra_hip_get_matches(zultra_hip_ctx_t *ctx, uint32_t block, uint16_t *out);
int zultra_hip_get_splits(zultra_hip_ctx_t *ctx, uint32_t block, int *out /* [64] */);
int zultra_hip_get_parse(zultra_hip_ctx_t *ctx, uint32_t block, uint16_t *out);

/*
 * Stitcher (host): appends the framed sub-blocks of a batch to a deflate stream, reproducing libzultra.c:327-398
 * (3 header bits, compressed-or-stored decision from the running bit phase, stored pieces of <= 65535 bytes) and
 * the bit carry across max-blocks (:427-434).
 *   state        : running bit writer (zero-initialise before the first batch of a stream)
 *   raw          : the raw bytes of the batch's max-blocks, max-block b at raw + raw_off[b] (for stored sub-blocks)
 *   final_block  : index of the max-block that ends the stream (ZULTRA_FINALIZE and no more input), or -1
 * Writes whole bytes to out (capacity out_cap) and returns the number written, or (size_t)-1 on overflow / error
 * (the reference's ZULTRA_ERROR_DST). With out == NULL nothing is written: the call only advances `state` and returns
 * the byte count, which is how a rank learns at which bit offset its shard starts (multi-GPU).
 * `state` may start at any phase 0..7 with acc = 0: the first byte then carries only this shard's bits. Call zultra_hip_stitch_finish() after the last batch to pad to a byte
 * (libzultra.c:414-417).
 */
typedef struct zultra_hip_bitstate_s {
   uint32_t acc;     /* pending bits, LSB first (zultra_bitwriter_t.nEncBitsData, bitwriter.h:27-33) */
   uint32_t nacc;    /* 0..7             (nEncBitCount) */
} zultra_hip_bitstate_t;

size_t zultra_hip_stitch(zultra_hip_bitstate_t *state, const zultra_hip_subblock_t *subs, uint32_t nsubs,
                         const uint8_t *payload, const uint8_t *raw, const uint64_t *raw_off, uint32_t max_block_size,
                         int final_block, uint8_t *out, size_t out_cap);
size_t zultra_hip_stitch_finish(zultra_hip_bitstate_t *state, uint8_t *out, size_t out_cap);

/*
 * Stitcher (device): same result as zultra_hip_stitch for the last batch, with nothing on the host: a scan kernel takes the
 * phase-dependent decisions of libzultra.c:327-398 from the 48-byte descriptors (a transfer table bit phase -> bits added
 * per run of max-blocks, composed across the batch), a second kernel moves the bits; the stream stays in HBM.
 *   state->nacc : pending bits (0..7) before the batch; updated to the pending bits after it (state->acc is not used:
 *                 the caller ORs its pending bits into byte 0 of the result, and reads the new partial byte back).
 *   *end_bit    : bits from the start of byte 0 to the end of the batch; the buffer holds ceil(end_bit/8) bytes.
 * Returns 0, -2 where the reference fails with ZULTRA_ERROR_DST, -1 on HIP errors.
 */
int zultra_hip_stitch_device(zultra_hip_ctx_t *ctx, zultra_hip_bitstate_t *state, int final_block, uint64_t *end_bit);
/*
 * The next batch of max-blocks (zultra_hip_compress_blocks) is stitched at bit phase `phase` (0..7; final_block as for zultra_hip_stitch_device) right behind
 * its last kernel, before the host is woken: a caller that knows the phase its batch starts at — every batch of a stream compressed batch by batch
 * (libzultra.c:327-398 carries the phase from max-block to max-block), the first job of zultra_memory_compress — saves the second synchronisation. The
 * zultra_hip_stitch_device call that follows with the same phase and final_block returns that stitch's result without launching anything; with other
 * arguments it stitches again. The setting is consumed by one batch. enable = 0 disarms. Returns 0, -1 for a files context.
 */
int zultra_hip_stitch_with_batch(zultra_hip_ctx_t *ctx, int enable, uint32_t phase, int final_block);
/*
 * Where the last batch would end for each of the eight bit phases it could start at (the same scan, nothing written):
 * end_bits[p] counts from the start of the byte that holds the p pending bits; bit p of *failed_mask is set where the
 * reference would fail with ZULTRA_ERROR_DST from that phase. What a device hands its neighbours when one stream
 * (libzultra.c:601-619) is cut over several of them: a shard's bit length depends on the phase it starts at.
 */
int zultra_hip_stitch_phase_table(zultra_hip_ctx_t *ctx, uint64_t *end_bits /* 8 */, uint32_t *failed_mask);   /* (rewrites the last stitch's items and report: stitch again before reading the stream) */
const void *zultra_hip_stream_device(const zultra_hip_ctx_t *ctx);
int zultra_hip_stream_read(zultra_hip_ctx_t *ctx, void *out, size_t offset, size_t nbytes);

/*
 * Many small independent inputs ("files", BASELINE.json configuration 5: 4 KiB records, each its own stream). A files
 * context takes inputs below 8192 bytes — the splitter never cuts those (blockdeflate.c:646), so a batch needs no host
 * decision and its whole kernel sequence is replayed from one captured hipGraph. zultra_hip_compress_files runs
 * stages 1-3 with every input as a max-block without history, then lays the raw deflate streams end to end in the
 * device stream buffer (zultra_hip_stream_read): input i occupies bytes file_off[i] .. file_off[i+1]. Each equals what
 * zultra_memory_compress(input i, ZULTRA_FLAG_RAW_DEFLATE) produces; gzip framing = 10-byte header + these bytes +
 * CRC-32 (zultra_hip_block_crc32 / zultra_crc32_append) + ISIZE. zultra_hip_stitch_files does the assembly alone for a
 * batch already compressed with zultra_hip_compress_blocks. Returns the number of inputs, or a negative error.
 */
zultra_hip_ctx_t *zultra_hip_create_files(int device, uint32_t max_file_size, uint32_t max_files);
int zultra_hip_compress_files(zultra_hip_ctx_t *ctx, const void *data, size_t data_size, int data_on_device, const uint64_t *offsets,
                              const uint32_t *sizes, uint32_t nfiles, uint64_t *file_off /* nfiles + 1 */);
int zultra_hip_stitch_files(zultra_hip_ctx_t *ctx, uint64_t *file_off /* blocks + 1 */);

/* CRC-32 of every max-block of the last batch, computed on the device while the blocks are compressed: out[b] is the
 * linear part (zero initial state, no final inversion). zultra_crc32_append() folds one block into a running gzip CRC
 * exactly as zultra_frame_update_checksum(crc, block, len, GZIP) would (reference src/frame.c:324-354,473-480). */
int zultra_hip_block_crc32(const zultra_hip_ctx_t *ctx, uint32_t *out);
uint32_t zultra_crc32_append(uint32_t crc, uint32_t block_linear_crc, size_t block_len);
/* Same for zlib framing: out[2b] = sum of the bytes of max-block b, out[2b+1] = sum of (n - i) * byte[i], both mod 65521.
 * zultra_adler32_append() folds one block into a running Adler-32 exactly as zultra_frame_update_checksum(adler, block,
 * len, ZLIB) would (reference src/frame.c:74-138). */
int zultra_hip_block_adler32(const zultra_hip_ctx_t *ctx, uint32_t *out);
uint32_t zultra_adler32_append(uint32_t adler, uint32_t block_sum, uint32_t block_weighted_sum, size_t block_len);
uint32_t zultra_crc32_append_many(uint32_t crc, const uint32_t *block_linear_crc, const uint32_t *block_len, uint32_t nblocks);

#ifdef __cplusplus
}
#endif
#endif /* ZULTRA_HIP_H */
