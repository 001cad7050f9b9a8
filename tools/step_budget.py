#!/usr/bin/env python3
"""Diagnostics: where the wall time of bench.py's configuration-2 step goes on the host side — laps around the calls of one step (N = 1, input in HBM).
usage: python tools/step_budget.py [bytes]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import corpus  # noqa: E402
import zultra_amd  # noqa: E402
from zultra_amd import sharded  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
d = corpus.real_text(size)
bs = 65536
nb = (size + bs - 1) // bs
blocks = [(b * bs - (32768 if b else 0), 32768 if b else 0, min(bs, size - b * bs)) for b in range(nb)]
lens = np.array([b[2] for b in blocks], dtype=np.uint32)
import torch  # noqa: E402
L = zultra_amd.lib()
dev = torch.device("cuda:0")
dd = torch.from_numpy(d).to(dev)
torch.cuda.synchronize()
ctx = L.context(bs, nb)
laps = []
for it in range(12):
    torch.cuda.synchronize()
    t = [time.perf_counter()]
    ctx.stitch_with_batch(nb - 1, phase=0)
    ctx.compress_blocks(dd.data_ptr(), blocks, data_on_device=True, data_size=dd.numel()); t.append(time.perf_counter())
    tm = ctx.timing(); t.append(time.perf_counter())
    v = L.crc32_append_many(0xFFFFFFFF, ctx.block_crc32(), lens); t.append(time.perf_counter())
    end_bit, _ = ctx.stitch_device(nb - 1, phase=0); t.append(time.perf_counter())
    n = (end_bit + 7) // 8
    st = sharded._stream_tensor(ctx, torch, dev, n); t.append(time.perf_counter())
    body = sharded._to_host(torch, st); t.append(time.perf_counter())
    torch.cuda.synchronize(); t.append(time.perf_counter())
    if it >= 4:
        laps.append([(b - a) * 1e3 for a, b in zip(t[:-1], t[1:])] + [tm["total_ms"]])
a = np.median(np.array(laps), axis=0)
print("ms (median of 8): compress_blocks call %.3f (device pipeline by events %.3f) | timing() %.3f | crc fold %.3f | stitch_device (cached) %.3f | stream tensor %.3f | D2H of %d bytes %.3f | final sync %.3f | sum %.3f"
      % (a[0], a[7], a[1], a[2], a[3], a[4], n, a[5], a[6], a[:7].sum()))
