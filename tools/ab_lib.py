#!/usr/bin/env python3
"""Diagnostics: per-kernel-group device times of one batch through a GIVEN build of the library (A/B of compile-time variants).
usage: python tools/ab_lib.py <path to .so> [bytes] [corpus: text|pysrc|json|mixed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import corpus  # noqa: E402
from zultra_amd._ffi import Lib  # noqa: E402

so = sys.argv[1]
size = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000_000
kind = sys.argv[3] if len(sys.argv) > 3 else "pysrc"
L = Lib(so)
if kind == "pysrc":
    d = corpus.real_text(size)
elif kind == "text":
    d = corpus.text_like_fast(size, 1000)
elif kind == "json":
    d = np.concatenate([corpus.json_like(1 << 20, 5 + k) for k in range((size + (1 << 20) - 1) >> 20)])[:size]
else:
    d = corpus.mixed_config4(0, size >> 20)
size, bs = len(d), 65536
nb = (size + bs - 1) // bs
blocks = [(b * bs - (32768 if b else 0), 32768 if b else 0, min(bs, size - b * bs)) for b in range(nb)]
ctx = L.context(bs, nb)
for it in range(3):
    ctx.compress_blocks(d, blocks)
t = ctx.timing()
print(os.path.basename(so), kind, size, " ".join("%s=%.2f" % (k[:-3], v) for k, v in t.items() if v))
