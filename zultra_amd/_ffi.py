"""ctypes bindings for the C ABI of libzultra_amd.so (include/libzultra.h + include/zultra_hip.h).

Python here is plumbing for tests and benchmarks; the product is the shared library. The names mirror the
reference's API (src/libzultra.h:104-157): ``memory_bound``, ``memory_compress``, a ``Stream`` object for
``zultra_stream_init / set_dictionary / compress / end``, plus ``HipContext`` for the batched device layer.
"""
import ctypes as C

import numpy as np

ZULTRA_OK = 0
ZULTRA_STREAM_END = 1
ZULTRA_ERROR_SRC = -1
ZULTRA_ERROR_DST = -2
ZULTRA_ERROR_DICTIONARY = -3
ZULTRA_ERROR_MEMORY = -4
ZULTRA_ERROR_COMPRESSION = -5

FLAG_DEFLATE = 0
FLAG_ZLIB = 1
FLAG_GZIP = 2

CONTINUE = 0
FINALIZE = 1

_u8p = C.POINTER(C.c_uint8)
_SIZE_MAX = C.c_size_t(-1).value


class ZultraError(RuntimeError):
    pass


class _Stream(C.Structure):
    # layout of zultra_stream_t, reference src/libzultra.h:78-93
    _fields_ = [("next_in", C.c_void_p), ("avail_in", C.c_size_t), ("total_in", C.c_ulonglong),
                ("next_out", C.c_void_p), ("avail_out", C.c_size_t), ("total_out", C.c_ulonglong),
                ("zalloc", C.c_void_p), ("zfree", C.c_void_p), ("opaque", C.c_void_p),
                ("state", C.c_void_p), ("adler", C.c_uint)]


class Block(C.Structure):
    _fields_ = [("win_off", C.c_uint64), ("prev", C.c_uint32), ("n", C.c_uint32)]


class SubBlock(C.Structure):
    _fields_ = [("block", C.c_uint32), ("start", C.c_uint32), ("size", C.c_uint32), ("is_dynamic", C.c_uint32),
                ("static_cost", C.c_int32), ("dynamic_cost", C.c_int32), ("failed", C.c_uint32), ("reserved", C.c_uint32),
                ("nbits", C.c_uint64), ("bits_off", C.c_uint64)]


class Timing(C.Structure):
    _fields_ = [(k, C.c_float) for k in ("h2d_ms", "matchfinder_ms", "tokenize_split_ms", "encode_ms", "d2h_ms", "total_ms",
                                         "group_ms", "frontier_ms", "stitch_ms", "init_ms", "parse_ms", "build_ms", "post_ms", "emit_ms", "head_ms", "tail_ms")]


class Stats(C.Structure):
    _fields_ = [("positions", C.c_uint64), ("huge_positions", C.c_uint64), ("blocks", C.c_uint32), ("subblocks", C.c_uint32),
                ("tasks", C.c_uint32), ("huge_tasks", C.c_uint32), ("cut_tasks", C.c_uint32), ("cut_segments", C.c_uint32),
                ("cut_redone", C.c_uint32), ("runs", C.c_uint32), ("settled_passes", C.c_uint32), ("settled_kib", C.c_uint32), ("cut_demoted", C.c_uint32),
                ("runs_without_chain_kernels", C.c_uint32), ("batches_rerun", C.c_uint32)]


class BitState(C.Structure):
    _fields_ = [("acc", C.c_uint32), ("nacc", C.c_uint32)]


EXPORTS = [
    # include/libzultra.h
    "zultra_stream_init", "zultra_stream_set_dictionary", "zultra_stream_compress", "zultra_stream_end",
    "zultra_memory_bound", "zultra_memory_compress", "zultra_memory_compress_dict", "zultra_set_device", "zultra_set_devices",
    "zultra_frame_get_header_size", "zultra_frame_encode_header", "zultra_frame_init_checksum",
    "zultra_frame_update_checksum", "zultra_frame_get_footer_size", "zultra_frame_encode_footer",
    "zultra_dictionary_load", "zultra_dictionary_free",
    # include/zultra_hip.h
    "zultra_hip_device_count", "zultra_hip_selftest", "zultra_hip_traffic_probe", "zultra_hip_create", "zultra_hip_destroy", "zultra_hip_last_error",
    "zultra_hip_data_capacity", "zultra_hip_compress_blocks", "zultra_hip_subblocks", "zultra_hip_payload",
    "zultra_hip_last_timing", "zultra_hip_get_matches", "zultra_hip_get_splits", "zultra_hip_get_parse",
    "zultra_hip_stitch", "zultra_hip_stitch_finish",
    "zultra_hip_stitch_device", "zultra_hip_stitch_phase_table", "zultra_hip_stream_device", "zultra_hip_stream_read", "zultra_hip_block_crc32", "zultra_crc32_append", "zultra_crc32_append_many",
    "zultra_hip_create_files", "zultra_hip_compress_files", "zultra_hip_stitch_files", "zultra_hip_staging",
    "zultra_hip_block_adler32", "zultra_adler32_append", "zultra_hip_copy_bandwidth", "zultra_hip_last_stats",
    "zultra_hip_ctx_info", "zultra_hip_context_bytes", "zultra_hip_context_bytes_on", "zultra_release_cached_contexts", "zultra_hip_chain_trace", "zultra_hip_cut_tasks", "zultra_hip_stitch_with_batch",
]


def _as_u8(data):
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8)
    return np.frombuffer(bytes(data), dtype=np.uint8)


class Lib:
    def __init__(self, path, allow_missing=()):
        """allow_missing: exports an OLDER build of the library may lack (A/B tools under tools/ that load a previous round's build; the product loader passes none)."""
        self.path = path
        L = self.L = C.CDLL(path)
        missing = [n for n in EXPORTS if not hasattr(L, n) and n not in allow_missing]
        if missing:
            raise ZultraError("%s does not export %s" % (path, missing))
        L.zultra_stream_init.argtypes = [C.POINTER(_Stream), C.c_uint, C.c_uint]
        L.zultra_stream_init.restype = C.c_int
        L.zultra_stream_set_dictionary.argtypes = [C.POINTER(_Stream), C.c_void_p, C.c_int]
        L.zultra_stream_set_dictionary.restype = C.c_int
        L.zultra_stream_compress.argtypes = [C.POINTER(_Stream), C.c_int]
        L.zultra_stream_compress.restype = C.c_int
        L.zultra_stream_end.argtypes = [C.POINTER(_Stream)]
        L.zultra_stream_end.restype = None
        L.zultra_memory_bound.argtypes = [C.c_size_t, C.c_uint, C.c_uint]
        L.zultra_memory_bound.restype = C.c_size_t
        L.zultra_memory_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint, C.c_uint]
        L.zultra_memory_compress.restype = C.c_size_t
        L.zultra_memory_compress_dict.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint, C.c_uint, C.c_void_p, C.c_int]
        L.zultra_memory_compress_dict.restype = C.c_size_t
        L.zultra_set_device.argtypes = [C.c_int]
        L.zultra_frame_update_checksum.argtypes = [C.c_uint, C.c_void_p, C.c_size_t, C.c_uint]
        L.zultra_frame_update_checksum.restype = C.c_uint
        L.zultra_frame_init_checksum.argtypes = [C.c_uint]
        L.zultra_frame_init_checksum.restype = C.c_uint
        L.zultra_frame_encode_header.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_void_p, C.c_int]
        L.zultra_frame_encode_footer.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_longlong, C.c_uint]
        L.zultra_frame_get_header_size.argtypes = [C.c_uint, C.c_void_p, C.c_int]
        L.zultra_frame_get_footer_size.argtypes = [C.c_uint]
        L.zultra_hip_device_count.restype = C.c_int
        L.zultra_hip_create.argtypes = [C.c_int, C.c_uint32, C.c_uint32]
        L.zultra_hip_create.restype = C.c_void_p
        L.zultra_hip_destroy.argtypes = [C.c_void_p]
        L.zultra_hip_last_error.argtypes = [C.c_void_p]
        L.zultra_hip_last_error.restype = C.c_char_p
        L.zultra_hip_data_capacity.argtypes = [C.c_void_p]
        L.zultra_hip_data_capacity.restype = C.c_size_t
        L.zultra_hip_compress_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(Block), C.c_uint32]
        L.zultra_hip_compress_blocks.restype = C.c_int
        L.zultra_hip_subblocks.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        L.zultra_hip_subblocks.restype = C.POINTER(SubBlock)
        L.zultra_hip_payload.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        L.zultra_hip_payload.restype = _u8p
        L.zultra_hip_last_timing.argtypes = [C.c_void_p, C.POINTER(Timing)]
        L.zultra_hip_get_matches.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.zultra_hip_get_splits.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_int)]
        L.zultra_hip_get_parse.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.zultra_hip_stitch.argtypes = [C.POINTER(BitState), C.POINTER(SubBlock), C.c_uint32, C.c_void_p, C.c_void_p,
                                        C.POINTER(C.c_uint64), C.c_uint32, C.c_int, C.c_void_p, C.c_size_t]
        L.zultra_hip_stitch.restype = C.c_size_t
        L.zultra_hip_stitch_finish.argtypes = [C.POINTER(BitState), C.c_void_p, C.c_size_t]
        L.zultra_hip_stitch_finish.restype = C.c_size_t

    # ---- libzultra.h -------------------------------------------------------------------------------------
    def device_count(self):
        return self.L.zultra_hip_device_count()

    def traffic_probe(self, nbytes):
        self.L.zultra_hip_traffic_probe.argtypes = [C.c_size_t]
        return self.L.zultra_hip_traffic_probe(nbytes)

    def copy_bandwidth(self, nbytes=1 << 30, iters=5):
        """Measured GB/s (read + written) of a 16 B/lane streaming copy: the roofline's second denominator."""
        self.L.zultra_hip_copy_bandwidth.argtypes = [C.c_size_t, C.c_int]
        self.L.zultra_hip_copy_bandwidth.restype = C.c_double
        return self.L.zultra_hip_copy_bandwidth(nbytes, iters)

    def memory_bound(self, n, flags, max_block=0):
        return self.L.zultra_memory_bound(n, flags, max_block)

    def memory_compress(self, data, flags, max_block=0, dictionary=None, cap=None):
        """-> compressed bytes, or None where the reference returns (size_t)-1."""
        data = _as_u8(data)
        if cap is None:
            cap = self.memory_bound(len(data), flags, max_block) + 16
        out = np.empty(max(cap, 1), dtype=np.uint8)
        if dictionary is not None and len(dictionary):
            d = _as_u8(dictionary)
            r = self.L.zultra_memory_compress_dict(data.ctypes.data, len(data), out.ctypes.data, cap, flags, max_block, d.ctypes.data, len(d))
        else:
            r = self.L.zultra_memory_compress(data.ctypes.data, len(data), out.ctypes.data, cap, flags, max_block)
        if r == _SIZE_MAX:
            return None
        return out[:r].tobytes()

    def memory_compress_into(self, data, flags, max_block, out):
        """zultra_memory_compress into a caller-owned uint8 array (no allocation, no copy of the result): the number of bytes, or None."""
        data = _as_u8(data)
        r = self.L.zultra_memory_compress(data.ctypes.data, len(data), out.ctypes.data, len(out), flags, max_block)
        return None if r == _SIZE_MAX else int(r)

    def checksum(self, data, flags, start=None):
        data = _as_u8(data)
        if start is None:
            start = self.L.zultra_frame_init_checksum(flags)
        return self.L.zultra_frame_update_checksum(start, data.ctypes.data, len(data), flags)

    def crc32_append(self, crc, block_linear_crc, block_len):
        self.L.zultra_crc32_append.argtypes = [C.c_uint32, C.c_uint32, C.c_size_t]
        self.L.zultra_crc32_append.restype = C.c_uint32
        return self.L.zultra_crc32_append(crc, int(block_linear_crc), block_len)

    def adler32_append(self, adler, a, bw, block_len):
        self.L.zultra_adler32_append.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_size_t]
        self.L.zultra_adler32_append.restype = C.c_uint32
        return self.L.zultra_adler32_append(adler, int(a), int(bw), block_len)

    def crc32_append_many(self, crc, block_linear_crcs, block_lens):
        a = np.ascontiguousarray(block_linear_crcs, dtype=np.uint32)
        n = np.ascontiguousarray(block_lens, dtype=np.uint32)
        f = self.L.zultra_crc32_append_many
        f.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32]
        f.restype = C.c_uint32
        return f(crc, a.ctypes.data, n.ctypes.data, len(a))

    def stream(self, flags, max_block=0, zalloc=None, zfree=None):
        return Stream(self, flags, max_block, zalloc, zfree)

    def _stream_default(self, flags, max_block=0):
        return Stream(self, flags, max_block)

    def context(self, max_block, max_blocks, device=0):
        return HipContext(self, device, max_block, max_blocks)

    def files_context(self, max_file_size, max_files, device=0):
        return HipContext(self, device, max_file_size, max_files, files=True)


ZALLOC_T = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_uint, C.c_uint)
ZFREE_T = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)


class Stream:
    """zultra_stream_t driven the way tool/zultra.c:151-186 drives it."""

    def __init__(self, lib, flags, max_block=0, zalloc=None, zfree=None):
        """zalloc / zfree: ctypes callbacks (ZALLOC_T / ZFREE_T) for the stream's own memory, as libzultra.h:88-90; default malloc / free."""
        self.lib = lib
        self.s = _Stream()
        self._alloc = (zalloc, zfree)
        if zalloc is not None:
            self.s.zalloc = C.cast(zalloc, C.c_void_p)
            self.s.zfree = C.cast(zfree, C.c_void_p)
        rc = lib.L.zultra_stream_init(C.byref(self.s), flags, max_block)
        if rc != ZULTRA_OK:
            raise ZultraError("zultra_stream_init failed: %d (no HIP device? the library has no CPU path)" % rc)
        self._keep = []
        self.ended = False

    def set_dictionary(self, d):
        d = _as_u8(d)
        self._keep.append(d)
        return self.lib.L.zultra_stream_set_dictionary(C.byref(self.s), d.ctypes.data, len(d))

    def compress(self, chunk, finalize, out_chunk=1 << 16):
        """Feed one chunk; returns (status, bytes produced)."""
        chunk = _as_u8(chunk)
        self._keep = self._keep[-2:] + [chunk]
        self.s.next_in = chunk.ctypes.data if len(chunk) else None
        self.s.avail_in = len(chunk)
        out = bytearray()
        buf = np.empty(out_chunk, dtype=np.uint8)
        while True:
            self.s.next_out = buf.ctypes.data
            self.s.avail_out = out_chunk
            st = self.lib.L.zultra_stream_compress(C.byref(self.s), FINALIZE if finalize else CONTINUE)
            out += buf[: out_chunk - self.s.avail_out].tobytes()
            if st != ZULTRA_OK:
                return st, bytes(out)
            if self.s.avail_in == 0 and self.s.avail_out != 0:
                return st, bytes(out)

    @property
    def total_in(self):
        return self.s.total_in

    @property
    def total_out(self):
        return self.s.total_out

    def end(self):
        if not self.ended:
            self.lib.L.zultra_stream_end(C.byref(self.s))
            self.ended = True

    def __del__(self):
        try:
            self.end()
        except Exception:
            pass


class HipContext:
    """zultra_hip_ctx_t: batches of independent max-blocks (include/zultra_hip.h)."""

    def __init__(self, lib, device, max_block, max_blocks, files=False):
        self.lib = lib
        lib.L.zultra_hip_create_files.argtypes = [C.c_int, C.c_uint32, C.c_uint32]
        lib.L.zultra_hip_create_files.restype = C.c_void_p
        self.h = (lib.L.zultra_hip_create_files if files else lib.L.zultra_hip_create)(device, max_block, max_blocks)
        if not self.h:
            raise ZultraError("zultra_hip_create failed: no usable HIP device (there is no CPU fallback)")
        self.max_block = max_block
        self._data = None
        self._block_n = None   # sizes of the last batch's blocks (list or uint32 array)
        self._blocks_src = self._blocks_arr = None

    def close(self):
        if self.h:
            self.lib.L.zultra_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def compress_blocks(self, data, blocks, data_on_device=False, data_size=None):
        """data: uint8 array (host) or an integer device pointer; blocks: list of (win_off, prev, n)."""
        if blocks is self._blocks_src:   # same list object as last time: reuse the marshalled descriptors
            arr = self._blocks_arr
        else:
            arr = (Block * len(blocks))(*[Block(int(o), int(p), int(n)) for (o, p, n) in blocks])
            self._blocks_src, self._blocks_arr = blocks, arr
            self._block_n = [int(b[2]) for b in blocks]
        if data_on_device:
            ptr, size = int(data), int(data_size)
        else:
            self._data = _as_u8(data)
            ptr, size = self._data.ctypes.data, len(self._data)
        n = self.lib.L.zultra_hip_compress_blocks(self.h, ptr, size, 1 if data_on_device else 0, arr, len(blocks))
        if n <= 0:
            raise ZultraError("zultra_hip_compress_blocks: " + self.lib.L.zultra_hip_last_error(self.h).decode())
        return n

    def compress_files(self, data, offsets, sizes, data_on_device=False, data_size=None):
        """Each (offset, size) is an independent input; -> uint64 array file_off[n+1] into the device stream buffer."""
        offs = np.ascontiguousarray(offsets, dtype=np.uint64)
        szs = np.ascontiguousarray(sizes, dtype=np.uint32)
        self._block_n = szs
        out = np.zeros(len(offs) + 1, dtype=np.uint64)
        if data_on_device:
            ptr, size = int(data), int(data_size)
        else:
            self._data = _as_u8(data)
            ptr, size = self._data.ctypes.data, len(self._data)
        f = self.lib.L.zultra_hip_compress_files
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        n = f(self.h, ptr, size, 1 if data_on_device else 0, offs.ctypes.data, szs.ctypes.data, len(offs), out.ctypes.data)
        if n <= 0:
            raise ZultraError("zultra_hip_compress_files: " + self.lib.L.zultra_hip_last_error(self.h).decode())
        return out

    def subblocks(self):
        cnt = C.c_uint32()
        p = self.lib.L.zultra_hip_subblocks(self.h, C.byref(cnt))
        return [p[i] for i in range(cnt.value)], p, cnt.value

    def subblocks_raw(self):
        """-> (pointer to zultra_hip_subblock_t[count], count) without building Python objects."""
        cnt = C.c_uint32()
        p = self.lib.L.zultra_hip_subblocks(self.h, C.byref(cnt))
        return p, cnt.value

    def subblock_bits(self, sb):
        size = C.c_size_t()
        p = self.lib.L.zultra_hip_payload(self.h, C.byref(size))
        nbytes = (sb.nbits + 7) // 8
        return bytes(bytearray(p[sb.bits_off: sb.bits_off + nbytes]))

    def timing(self):
        t = Timing()
        self.lib.L.zultra_hip_last_timing(self.h, C.byref(t))
        return {k: getattr(t, k) for k, _ in Timing._fields_}

    def stats(self):
        st = Stats()
        self.lib.L.zultra_hip_last_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        self.lib.L.zultra_hip_last_stats.restype = None
        self.lib.L.zultra_hip_last_stats(self.h, C.byref(st))
        return {k: getattr(st, k) for k, _ in Stats._fields_}

    def matches(self, block):
        n = int(self._block_n[block])
        m = np.zeros((n, 8, 2), dtype=np.uint16)
        if self.lib.L.zultra_hip_get_matches(self.h, block, m.ctypes.data) != 0:
            raise ZultraError("get_matches")
        return m

    def splits(self, block):
        out = (C.c_int * 64)()
        k = self.lib.L.zultra_hip_get_splits(self.h, block, out)
        if k < 0:
            raise ZultraError("get_splits")
        return list(out[:k])

    def parse(self, block):
        n = int(self._block_n[block])
        m = np.zeros((n, 2), dtype=np.uint16)
        if self.lib.L.zultra_hip_get_parse(self.h, block, m.ctypes.data) != 0:
            raise ZultraError("get_parse")
        return m

    def stitch_with_batch(self, final_block, phase=0, enable=True):
        """Arms the next compress_blocks to stitch at `phase` behind its last kernel (zultra_hip_stitch_with_batch); stitch_device with the same
        arguments then returns that result without a launch."""
        L = self.lib.L
        L.zultra_hip_stitch_with_batch.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_int]
        if L.zultra_hip_stitch_with_batch(self.h, 1 if enable else 0, phase, final_block) != 0:
            raise ZultraError("zultra_hip_stitch_with_batch")

    def stitch_device(self, final_block, phase=0):
        """Device stitcher over the last batch -> (end_bit, new_phase); the stream stays in HBM (stream_ptr)."""
        L = self.lib.L
        L.zultra_hip_stitch_device.argtypes = [C.c_void_p, C.POINTER(BitState), C.c_int, C.POINTER(C.c_uint64)]
        st = BitState(0, phase)
        eb = C.c_uint64()
        rc = L.zultra_hip_stitch_device(self.h, C.byref(st), final_block, C.byref(eb))
        if rc != 0:
            raise ZultraError("zultra_hip_stitch_device: %d %s" % (rc, L.zultra_hip_last_error(self.h).decode()))
        return eb.value, st.nacc

    def stream_ptr(self):
        self.lib.L.zultra_hip_stream_device.argtypes = [C.c_void_p]
        self.lib.L.zultra_hip_stream_device.restype = C.c_void_p
        return self.lib.L.zultra_hip_stream_device(self.h)

    def stream_read(self, nbytes, offset=0, out=None):
        """-> uint8 array of the stream bytes [offset, offset + nbytes). out: a uint8 array to read into (its first nbytes are returned) — a
        caller that reads batch after batch hands in one pinned buffer instead of touching fresh pageable memory every time."""
        if out is None:
            out = np.empty(nbytes, dtype=np.uint8)
        else:
            assert out.dtype == np.uint8 and out.flags["C_CONTIGUOUS"] and out.size >= nbytes
            out = out[:nbytes]
        self.lib.L.zultra_hip_stream_read.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
        if self.lib.L.zultra_hip_stream_read(self.h, out.ctypes.data, offset, nbytes) != 0:
            raise ZultraError("stream_read")
        return out

    def block_adler32(self):
        """-> uint32 array [n, 2]: per max-block / input the two Adler-32 sums (A, Bw) the device took next to the compression (zultra_adler32_append folds them)."""
        n = len(self._block_n)
        out = np.zeros((n, 2), dtype=np.uint32)
        self.lib.L.zultra_hip_block_adler32.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.L.zultra_hip_block_adler32(self.h, out.ctypes.data)
        return out

    def block_crc32(self):
        n = len(self._block_n)
        out = np.zeros(n, dtype=np.uint32)
        self.lib.L.zultra_hip_block_crc32.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.L.zultra_hip_block_crc32(self.h, out.ctypes.data)
        return out

    def stitch(self, raw, raw_offs, max_block, final_block, state=None, finish=True):
        """Host stitcher over the last batch -> (bytes, BitState)."""
        subs, p, cnt = self.subblocks()
        size = C.c_size_t()
        payload = self.lib.L.zultra_hip_payload(self.h, C.byref(size))
        raw = _as_u8(raw)
        offs = (C.c_uint64 * len(raw_offs))(*raw_offs)
        st = state or BitState(0, 0)
        cap = len(raw) + 64 * 6 * len(raw_offs) + 1024
        out = np.empty(cap, dtype=np.uint8)
        w = self.lib.L.zultra_hip_stitch(C.byref(st), p, cnt, payload, raw.ctypes.data, offs, max_block, final_block, out.ctypes.data, cap)
        if w == _SIZE_MAX:
            return None, st
        if finish:
            f = self.lib.L.zultra_hip_stitch_finish(C.byref(st), out.ctypes.data + w, cap - w)
            w += f
        return out[:w].tobytes(), st
