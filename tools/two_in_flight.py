#!/usr/bin/env python3
"""Diagnostics: throughput of N batches in flight on one GPU (one context and one host thread each, every batch the same 100 MB of
real text) against one batch at a time. usage: python tools/two_in_flight.py [bytes] [threads] [batches per thread] [path of another build of the library]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpus  # noqa: E402
import zultra_amd  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 2
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
bs = 65536
if len(sys.argv) > 4:
    from zultra_amd._ffi import Lib
    L = Lib(sys.argv[4])
else:
    L = zultra_amd.lib()
d = corpus.real_text(size)
size = len(d)
nb = (size + bs - 1) // bs
blocks = [(b * bs - (32768 if b else 0), 32768 if b else 0, min(bs, size - b * bs)) for b in range(nb)]
ctxs = [L.context(bs, nb) for _ in range(nthreads)]
for c in ctxs:
    c.compress_blocks(d, blocks)


def run(c, n):
    for _ in range(n):
        c.compress_blocks(d, blocks)


t0 = time.perf_counter()
run(ctxs[0], reps)
t1 = time.perf_counter()
one = size * reps / (t1 - t0) / 1e6
ths = [threading.Thread(target=run, args=(c, reps)) for c in ctxs]
t0 = time.perf_counter()
for t in ths:
    t.start()
for t in ths:
    t.join()
t1 = time.perf_counter()
many = size * reps * nthreads / (t1 - t0) / 1e6
print("%d bytes per batch: one at a time %.0f MB/s (%.2f ms per batch); %d in flight %.0f MB/s (%.2f ms per batch)" % (size, one, size / one / 1e3, nthreads, many, size / many / 1e3))
