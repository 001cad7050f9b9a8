#!/usr/bin/env python3
"""Diagnostics: per-kernel-group device times of one batch (HIP events on the library's stream).
usage: python tools/profile_encode.py [bytes] [corpus: text|mixed|json]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpus  # noqa: E402
import zultra_amd  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "text"
bs = 65536
L = zultra_amd.lib()
if kind == "text":
    d = corpus.text_like_fast(size, 1000)
elif kind == "mixed":
    import numpy as np
    d = np.concatenate([corpus.mixed(1 << 22, 5 + k) for k in range((size + (1 << 22) - 1) >> 22)])[:size]
elif kind == "pysrc":   # real text: the Python sources of this image (same files on the GPU box), concatenated
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
else:
    import numpy as np
    d = np.concatenate([corpus.json_like(1 << 20, 5 + k) for k in range((size + (1 << 20) - 1) >> 20)])[:size]
nb = (size + bs - 1) // bs
blocks = [(b * bs - (32768 if b else 0), 32768 if b else 0, min(bs, size - b * bs)) for b in range(nb)]
ctx = L.context(bs, nb)
for it in range(3):
    ctx.compress_blocks(d, blocks)
t = ctx.timing()
subs, _, cnt = ctx.subblocks()
print("%s %d bytes, %d max-blocks, %d sub-blocks (%d dynamic)" % (kind, size, nb, cnt, sum(s.is_dynamic for s in subs)))
for k, v in t.items():
    print("  %-20s %9.3f ms" % (k, v))
print("  kernels only: %.1f MB/s" % (size / ((t["matchfinder_ms"] + t["tokenize_split_ms"] + t["encode_ms"]) * 1e-3) / 1e6))
print("  stats:", ctx.stats())
