"""ctypes access to the two CPU checkers used by the test-suite (TEST INFRASTRUCTURE ONLY).

* ``Oracle``  — oracle/_build/libzultra_oracle.so, our plain-C restatement (built by oracle/Makefile).
* ``Ref``     — oracle/_ref/libzultra_ref.so, the reference itself compiled from /root/reference plus
                oracle/ref_probe.c; present only where it was built (this container) or travelled to
                (the GPU box). Tests that need it skip when it is absent.

Nothing in the product package imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "_build", "libzultra_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libzultra_ref.so")

NM = 8
MAX_SPLITS = 64

_u8p = C.POINTER(C.c_uint8)


def _ptr(a, t=_u8p):
    return a.ctypes.data_as(t)


def build_oracle():
    """Compile the oracle (and the reference build when /root/reference is present)."""
    import fcntl
    os.makedirs(os.path.join(ORACLE_DIR, "_build"), exist_ok=True)
    with open(os.path.join(ORACLE_DIR, "_build", ".lock"), "w") as lock:   # several test processes may get here at once (pytest-xdist)
        fcntl.flock(lock, fcntl.LOCK_EX)
        subprocess.run(["make", "-s", "-C", ORACLE_DIR, "oracle"], check=True)
        if os.path.isdir("/root/reference/src"):
            subprocess.run(["make", "-s", "-C", ORACLE_DIR, "ref"], check=True)


def as_u8(data):
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8)
    return np.frombuffer(bytes(data), dtype=np.uint8).copy()


class Oracle:
    def __init__(self):
        if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(
            os.path.join(ORACLE_DIR, "zultra_oracle.c")
        ):
            build_oracle()
        L = self.lib = C.CDLL(ORACLE_SO)
        L.zo_find_matches.argtypes = [_u8p, C.c_int, C.c_int, C.c_void_p]
        L.zo_find_matches.restype = None
        L.zo_block_split.argtypes = [_u8p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.zo_block_split.restype = C.c_int
        L.zo_subblock_costs.argtypes = [_u8p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.zo_subblock_costs.restype = C.c_int
        L.zo_subblock_deflate.argtypes = [_u8p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, _u8p, C.c_size_t,
                                          C.POINTER(C.c_uint64), C.c_void_p, C.c_void_p, C.c_void_p]
        L.zo_subblock_deflate.restype = C.c_int
        L.zo_memory_bound.argtypes = [C.c_size_t, C.c_uint, C.c_uint]
        L.zo_memory_bound.restype = C.c_size_t
        L.zo_memory_compress_dict.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, C.c_uint, C.c_uint, _u8p, C.c_int]
        L.zo_memory_compress_dict.restype = C.c_size_t
        L.zo_crc32.argtypes = [C.c_uint32, _u8p, C.c_size_t]
        L.zo_crc32.restype = C.c_uint32
        L.zo_adler32.argtypes = [C.c_uint32, _u8p, C.c_size_t]
        L.zo_adler32.restype = C.c_uint32
        L.zo_last_match_candidates.restype = C.c_uint64

    def find_matches(self, win, prev, n):
        """-> uint16 array [n, 8, 2] (length, offset)."""
        win = as_u8(win)
        assert len(win) == prev + n
        m = np.zeros((n, NM, 2), dtype=np.uint16)
        self.lib.zo_find_matches(_ptr(win), prev, n, m.ctypes.data)
        return m

    def split(self, win, match, prev, n):
        win = as_u8(win)
        out = (C.c_int * MAX_SPLITS)()
        k = self.lib.zo_block_split(_ptr(win), match.ctypes.data, prev, n, out)
        return list(out[:k]) if k > 0 else k

    def costs(self, win, match, prev, start, size):
        win = as_u8(win)
        s, d = C.c_int(), C.c_int()
        dyn = self.lib.zo_subblock_costs(_ptr(win), match.ctypes.data, prev, start, size, C.byref(s), C.byref(d))
        return dyn, s.value, d.value

    def deflate(self, win, match, prev, start, size, is_dynamic):
        """-> (rc, nbits, bytes, best[size,2], lit_len[288], dist_len[32])"""
        win = as_u8(win)
        cap = 2 * size + 1024
        out = np.zeros(cap, dtype=np.uint8)
        nbits = C.c_uint64()
        best = np.zeros((size, 2), dtype=np.uint16)
        ll = np.zeros(288, dtype=np.int32)
        dl = np.zeros(32, dtype=np.int32)
        rc = self.lib.zo_subblock_deflate(_ptr(win), match.ctypes.data, prev, start, size, is_dynamic, _ptr(out), cap,
                                          C.byref(nbits), best.ctypes.data, ll.ctypes.data, dl.ctypes.data)
        nb = nbits.value
        return rc, nb, out[: min(cap, (nb + 7) // 8)].tobytes(), best, ll, dl

    def memory_bound(self, n, flags, max_block):
        return self.lib.zo_memory_bound(n, flags, max_block)

    def memory_compress(self, data, flags, max_block, dictionary=None, cap=None):
        data = as_u8(data)
        if cap is None:
            cap = self.memory_bound(len(data), flags, max_block) + 16
        out = np.zeros(max(cap, 1), dtype=np.uint8)
        if dictionary is not None and len(dictionary):
            d = as_u8(dictionary)
            r = self.lib.zo_memory_compress_dict(_ptr(data), len(data), _ptr(out), cap, flags, max_block, _ptr(d), len(d))
        else:
            r = self.lib.zo_memory_compress_dict(_ptr(data), len(data), _ptr(out), cap, flags, max_block, None, 0)
        if r == C.c_size_t(-1).value:
            return None
        return out[:r].tobytes()

    def crc32(self, data, crc=0):
        data = as_u8(data)
        return self.lib.zo_crc32(crc, _ptr(data), len(data))

    def adler32(self, data, adler=1):
        data = as_u8(data)
        return self.lib.zo_adler32(adler, _ptr(data), len(data))


def have_ref():
    return os.path.exists(REF_SO)


class Ref:
    """The compiled reference + probes (oracle/ref_probe.c)."""

    def __init__(self):
        L = self.lib = C.CDLL(REF_SO)
        L.zultra_memory_compress.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, C.c_uint, C.c_uint]
        L.zultra_memory_compress.restype = C.c_size_t
        L.zultra_memory_bound.argtypes = [C.c_size_t, C.c_uint, C.c_uint]
        L.zultra_memory_bound.restype = C.c_size_t
        L.ref_probe_memory_compress_dict.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, C.c_uint, C.c_uint, _u8p, C.c_int]
        L.ref_probe_memory_compress_dict.restype = C.c_size_t
        L.ref_probe_create.argtypes = [C.c_int]
        L.ref_probe_create.restype = C.c_void_p
        L.ref_probe_destroy.argtypes = [C.c_void_p]
        L.ref_probe_analyse.argtypes = [C.c_void_p, _u8p, C.c_int, C.c_int]
        L.ref_probe_analyse.restype = C.c_int
        L.ref_probe_get_matches.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_probe_split.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.ref_probe_split.restype = C.c_int
        L.ref_probe_costs.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ref_probe_costs.restype = C.c_int
        L.ref_probe_deflate.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int, C.POINTER(C.c_longlong), C.c_void_p]
        L.ref_probe_deflate.restype = C.c_int
        L.ref_probe_get_codelens.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.zultra_frame_update_checksum.argtypes = [C.c_uint, _u8p, C.c_size_t, C.c_uint]
        L.zultra_frame_update_checksum.restype = C.c_uint

    def memory_bound(self, n, flags, max_block):
        return self.lib.zultra_memory_bound(n, flags, max_block)

    def memory_compress(self, data, flags, max_block, dictionary=None, cap=None):
        data = as_u8(data)
        if cap is None:
            cap = self.memory_bound(len(data), flags, max_block) + 16
        out = np.zeros(max(cap, 1), dtype=np.uint8)
        if dictionary is not None and len(dictionary):
            d = as_u8(dictionary)
            r = self.lib.ref_probe_memory_compress_dict(_ptr(data), len(data), _ptr(out), cap, flags, max_block, _ptr(d), len(d))
        else:
            r = self.lib.zultra_memory_compress(_ptr(data), len(data), _ptr(out), cap, flags, max_block)
        if r == C.c_size_t(-1).value:
            return None
        return out[:r].tobytes()

    def checksum(self, data, flags, start):
        data = as_u8(data)
        return self.lib.zultra_frame_update_checksum(start, _ptr(data), len(data), flags)

    class Probe:
        def __init__(self, ref, max_block, win, prev, n):
            self.L = ref.lib
            self.win = as_u8(win)
            assert len(self.win) == prev + n
            self.prev, self.n = prev, n
            self.h = self.L.ref_probe_create(max_block)
            assert self.h
            rc = self.L.ref_probe_analyse(self.h, _ptr(self.win), prev, n)
            assert rc == 0, rc

        def close(self):
            if self.h:
                self.L.ref_probe_destroy(self.h)
                self.h = None

        def __enter__(self):
            return self

        def __exit__(self, *a):
            self.close()

        def matches(self):
            m = np.zeros((self.n, NM, 2), dtype=np.uint16)
            self.L.ref_probe_get_matches(self.h, m.ctypes.data)
            return m

        def split(self):
            out = (C.c_int * MAX_SPLITS)()
            k = self.L.ref_probe_split(self.h, out)
            return list(out[:k]) if k > 0 else k

        def costs(self, start, size):
            s, d = C.c_int(), C.c_int()
            dyn = self.L.ref_probe_costs(self.h, start, size, C.byref(s), C.byref(d))
            return dyn, s.value, d.value

        def deflate(self, start, size, is_dynamic):
            cap = 2 * size + 1024
            out = np.zeros(cap, dtype=np.uint8)
            nbits = C.c_longlong()
            best = np.zeros((size, 2), dtype=np.uint16)
            rc = self.L.ref_probe_deflate(self.h, start, size, is_dynamic, _ptr(out), cap, C.byref(nbits), best.ctypes.data)
            ll = np.zeros(288, dtype=np.int32)
            dl = np.zeros(32, dtype=np.int32)
            self.L.ref_probe_get_codelens(self.h, ll.ctypes.data, dl.ctypes.data)
            nb = nbits.value if rc == 0 else 0
            return rc, nb, out[: (nb + 7) // 8].tobytes(), best, ll, dl

    def probe(self, max_block, win, prev, n):
        return Ref.Probe(self, max_block, win, prev, n)


class RefStages:
    """The compiled reference behind the Oracle's interface (find_matches / split / costs / deflate / memory_compress /
    memory_bound): the GPU suite's checker wherever oracle/_ref travelled (tests/conftest.py `checker`). One probe of the
    reference's compressor state is kept per window and reused by the stage calls that follow find_matches, which is the
    order every caller uses (tests/parity_util.py)."""

    def __init__(self):
        self.ref = Ref()
        self._probe = None
        self._key = None

    def _get(self, win, prev, n):
        win = as_u8(win)
        key = (win.ctypes.data, len(win), prev, n, int(win[:: max(1, len(win) // 64)].sum()))
        if self._probe is None or self._key != key:
            if self._probe is not None:
                self._probe.close()
            mb = 32768
            while mb < n:
                mb <<= 1
            self._probe = self.ref.probe(mb, win, prev, n)   # (keeps its own reference to the window)
            self._key = key
        return self._probe

    def find_matches(self, win, prev, n):
        if self._probe is not None:   # a new analysis: never reuse a probe across calls on a buffer that may have been rewritten in place
            self._probe.close()
            self._probe = None
        return self._get(win, prev, n).matches()

    def split(self, win, match, prev, n):
        return self._get(win, prev, n).split()

    def costs(self, win, match, prev, start, size):
        p = self._probe
        assert p is not None and p.prev == prev, "costs() follows find_matches() on the same window"
        return p.costs(start, size)

    def deflate(self, win, match, prev, start, size, is_dynamic):
        p = self._probe
        assert p is not None and p.prev == prev, "deflate() follows find_matches() on the same window"
        return p.deflate(start, size, is_dynamic)

    def memory_bound(self, n, flags, max_block):
        return self.ref.memory_bound(n, flags, max_block)

    def memory_compress(self, data, flags, max_block, dictionary=None, cap=None):
        return self.ref.memory_compress(data, flags, max_block, dictionary=dictionary, cap=cap)
