"""zultra_amd — MI355X-native deflate block compressor, drop-in for zultra's libzultra API.

The product is ``libzultra_amd.so`` (HIP kernels for gfx950 + C ABI, see include/). This package only loads
it and exposes thin ctypes wrappers. There is no CPU implementation behind it: loading fails loudly when the
library has not been built, and every compression entry point fails when no HIP device is usable.
"""
import os

from ._ffi import (CONTINUE, FINALIZE, FLAG_DEFLATE, FLAG_GZIP, FLAG_ZLIB, ZULTRA_OK, ZULTRA_STREAM_END, HipContext, Lib,
                   Stream, ZultraError)

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libzultra_amd.so")

_lib = None


def csrc_digest():
    """sha-256 (first 16 hex digits) over the kernel and host sources the library is built from (zultra_amd/csrc, include/): what a committed profile
    (profiles/*.json, tools/pmc_traffic.py, tools/sq_profile.py) was measured on, and what bench.py compares it with before quoting it."""
    import hashlib
    root = os.path.dirname(os.path.abspath(__file__))
    h = hashlib.sha256()
    for d in (os.path.join(root, "csrc"), os.path.join(root, "..", "include")):
        for name in sorted(os.listdir(d)):
            if name.endswith((".h", ".hip", ".cpp", ".c")):
                with open(os.path.join(d, name), "rb") as f:
                    h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def lib():
    """The loaded product library (zultra_amd/libzultra_amd.so, built in-tree by zultra_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ZultraError("%s is missing: run `python -m zultra_amd.build` (hipcc, gfx950). "
                              "There is no fallback implementation." % LIB_PATH)
        _lib = Lib(LIB_PATH)
    return _lib


def memory_bound(n, flags, max_block=0):
    return lib().memory_bound(n, flags, max_block)


def memory_compress(data, flags, max_block=0, dictionary=None):
    return lib().memory_compress(data, flags, max_block, dictionary)


def stream(flags, max_block=0):
    return lib().stream(flags, max_block)


def context(max_block, max_blocks, device=0):
    return lib().context(max_block, max_blocks, device)
