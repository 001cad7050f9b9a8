#!/usr/bin/env python3
"""Diagnostics: per-kernel-group device times of one batch at a given max-block size (the reference's default is 1 MiB).
usage: python tools/profile_blocks.py [bytes] [max_block] [corpus: text|pysrc]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpus  # noqa: E402
import zultra_amd  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
kind = sys.argv[3] if len(sys.argv) > 3 else "text"
L = zultra_amd.lib()
if kind == "text":
    d = corpus.text_like_fast(size, 1000)
else:
    import glob
    import numpy as np
    buf = bytearray()
    for f in sorted(glob.glob("/usr/lib/python3*/**/*.py", recursive=True)) + sorted(glob.glob("/usr/local/lib/python3*/**/*.py", recursive=True)):
        try:
            buf += open(f, "rb").read()
        except OSError:
            pass
        if len(buf) >= size:
            break
    d = np.frombuffer(bytes(buf[:size]), dtype=np.uint8).copy()
    size = len(d)
nb = (size + bs - 1) // bs
blocks = [(b * bs - (32768 if b else 0), 32768 if b else 0, min(bs, size - b * bs)) for b in range(nb)]
ctx = L.context(bs, nb)
for it in range(3):
    ctx.compress_blocks(d, blocks)
t = ctx.timing()
print("%s %d bytes, %d max-blocks of %d" % (kind, size, nb, bs))
for k, v in t.items():
    print("  %-20s %9.3f ms" % (k, v))
print("  whole batch: %.1f MB/s" % (size / (t["total_ms"] * 1e-3) / 1e6))
