#!/usr/bin/env python3
"""Diagnostics: the WHOLE step of bench.py's configuration 2 at one GPU — kernels, stitch, read-back of the stitched bytes into pinned memory — through several builds
of the library in one process, in alternating rounds (box-to-box differences of +-2.5 % drown a 2 % change measured on two boxes).
usage: python tools/ab_step.py [bytes] [corpus: pysrc|text|mixed] [rounds] -- <lib.so> [<lib.so> ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import corpus  # noqa: E402
from zultra_amd._ffi import Lib  # noqa: E402

args = sys.argv[1:]
sep = args.index("--")
size = int(args[0]) if sep > 0 else 100_000_000
kind = args[1] if sep > 1 else "pysrc"
rounds = int(args[2]) if sep > 2 else 5
libs = args[sep + 1:]
d = corpus.real_text(size) if kind == "pysrc" else corpus.text_like_fast(size, 1000) if kind == "text" else corpus.mixed_config4(0, size >> 20)
size, bs = len(d), 65536
nb = (size + bs - 1) // bs
blocks = [(b * bs - (32768 if b else 0), 32768 if b else 0, min(bs, size - b * bs)) for b in range(nb)]
import torch  # noqa: E402
dd = torch.from_numpy(d).cuda()
pin = torch.empty(size + (1 << 20), dtype=torch.uint8, pin_memory=True).numpy()
torch.cuda.synchronize()
ctxs = []
for so in libs:
    L = Lib(so, allow_missing=("zultra_hip_stitch_with_batch",))
    ctxs.append((so, L, L.context(bs, nb)))


def step(L, ctx):
    if os.environ.get("AB_NO_ARM") != "1" and hasattr(L.L, "zultra_hip_stitch_with_batch"):
        ctx.stitch_with_batch(nb - 1, phase=0)
    ctx.compress_blocks(dd.data_ptr(), blocks, data_on_device=True, data_size=dd.numel())
    end_bit, _ = ctx.stitch_device(nb - 1, phase=0)
    n = (end_bit + 7) // 8
    ctx.stream_read(n, out=pin)
    return n


res = {so: [] for so, _, _ in ctxs}
dev = {so: [] for so, _, _ in ctxs}
for so, L, ctx in ctxs:
    for _ in range(3):
        step(L, ctx)
for r in range(rounds):
    for so, L, ctx in ctxs:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            n = step(L, ctx)
        torch.cuda.synchronize()
        res[so].append((time.perf_counter() - t0) / 5 * 1e3)
        dev[so].append(ctx.timing()["total_ms"])
for so, L, ctx in ctxs:
    print("%-40s step ms: min %.2f med %.2f | device pipeline (last of each round) min %.2f med %.2f | %d bytes out" % (
        os.path.basename(so), min(res[so]), float(np.median(res[so])), min(dev[so]), float(np.median(dev[so])), n))
