#!/usr/bin/env python3
"""Diagnostics: where do the chain workgroups of a batch spend their time? (ZULTRA_HIP_CHAIN_TRACE)
usage: python tools/chain_trace.py [bytes] [corpus: pysrc|json|mixed]"""
import ctypes as C
import os
import sys

os.environ["ZULTRA_HIP_CHAIN_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import corpus  # noqa: E402
import zultra_amd  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "pysrc"
bs = 65536
L = zultra_amd.lib()
if kind == "pysrc":
    d = corpus.real_text(size)
elif kind == "json":
    d = np.concatenate([corpus.json_like(1 << 20, 5 + k) for k in range((size + (1 << 20) - 1) >> 20)])[:size]
else:
    d = corpus.mixed_config4(0, size >> 20)
size = len(d)
nb = (size + bs - 1) // bs
blocks = [(b * bs - (32768 if b else 0), 32768 if b else 0, min(bs, size - b * bs)) for b in range(nb)]
ctx = L.context(bs, nb)
for it in range(2):
    ctx.compress_blocks(d, blocks)
slots = C.c_uint32()
L.L.zultra_hip_chain_trace.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
L.L.zultra_hip_chain_trace(ctx.h, None, C.byref(slots))
tr = np.zeros((4, 4, slots.value, 3), dtype=np.uint64)
assert L.L.zultra_hip_chain_trace(ctx.h, tr.ctypes.data, C.byref(slots)) == 0
print(ctx.timing())
print(ctx.stats())
for run in range(3):
    for p in range(4):
        t = tr[run, p]
        t = t[t[:, 0] > 0]
        if not len(t):
            continue
        first = int(t[:, 1].min())
        pos, st, en = t[:, 0].astype(np.int64), (t[:, 1].astype(np.int64) - first) / 100.0, (t[:, 2].astype(np.int64) - first) / 100.0
        order = np.argsort(-pos)[:5]
        print("run %d pass %d: %d chains (first %d recorded), %d positions, last end %.0f us" % (run, p, len(t), slots.value, pos.sum(), en.max()))
        for i in order:
            print("    ticket %4d: %6d positions  start %8.0f us  end %8.0f us  %.3f us per position" % (i, pos[i], st[i], en[i], (en[i] - st[i]) / pos[i]))
        late = np.argsort(-en)[:3]
        for i in late:
            print("    latest end: ticket %4d: %6d positions  start %8.0f us  end %8.0f us" % (i, pos[i], st[i], en[i]))

# the cut tasks: how many of their cuts failed the check (and were parsed again, one after the other, by the wave that checks the task)
L.L.zultra_hip_cut_tasks.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
for run in range(ctx.stats()["runs"]):
    buf = np.zeros((1 << 16, 4), dtype=np.uint32)
    n = L.L.zultra_hip_cut_tasks(ctx.h, run, buf.ctypes.data, len(buf))
    if n <= 0:
        continue
    t = buf[:n]
    K, S, fails = t[:, 1] & 0xfff, (t[:, 1] >> 12) << 5, t[:, 3] >> 16
    checks = 4 * (K - 1)
    print("run %d: %d cut tasks, %d segments, %d failed cuts of %d checks over four passes" % (run, n, K.sum(), fails.sum(), checks.sum()))
    frac = fails / np.maximum(1, checks)
    for lo, hi in ((0, 0.0001), (0.0001, 0.1), (0.1, 0.25), (0.25, 0.5), (0.5, 0.75), (0.75, 1.01)):
        m = (frac >= lo) & (frac < hi)
        print("    failed %3.0f%%..%3.0f%% of their checks: %5d tasks, %7d segments, %6d failures, serial redo positions per pass (worst task) %d"
              % (100 * lo, 100 * hi, m.sum(), K[m].sum(), fails[m].sum(), int((fails[m] * S[m]).max() / 4) if m.any() else 0))
