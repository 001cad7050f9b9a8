#!/usr/bin/env python3
"""Diagnostics: the device pipeline of one batch with the input already in HBM (what bench.py's stream legs time), through a given build of the library.
usage: python tools/step_dev.py <path to .so> [bytes] [corpus: text|pysrc|mixed] [repetitions]      (under rocprofv3 for a timeline: tools/timeline.py)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import corpus  # noqa: E402
from zultra_amd._ffi import Lib  # noqa: E402

so = sys.argv[1]
size = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
kind = sys.argv[3] if len(sys.argv) > 3 else "text"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 8
d = corpus.real_text(size) if kind == "pysrc" else corpus.text_like_fast(size, 1000) if kind == "text" else corpus.mixed_config4(0, size >> 20)
size, bs = len(d), 65536
nb = (size + bs - 1) // bs
blocks = [(b * bs - (32768 if b else 0), 32768 if b else 0, min(bs, size - b * bs)) for b in range(nb)]
import torch  # noqa: E402
L = Lib(so)
dd = torch.from_numpy(d).cuda()
torch.cuda.synchronize()
ctx = L.context(bs, nb)
ts = []
for it in range(reps):
    ctx.compress_blocks(dd.data_ptr(), blocks, data_on_device=True, data_size=dd.numel())
    ts.append(ctx.timing()["total_ms"])
t = ctx.timing()
print(os.path.basename(so), kind, size, "total min %.2f med %.2f |" % (min(ts[2:]), float(np.median(ts[2:]))), " ".join("%s=%.2f" % (k[:-3], v) for k, v in t.items() if v))
