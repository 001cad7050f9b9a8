#!/usr/bin/env python3
"""The drop-in entry as lzbench would call it: zultra_memory_compress on a host buffer (PCIe-inclusive, pageable memory).
Reported next to bench.py's HBM-resident number; never the headline value."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpus  # noqa: E402
import zultra_amd  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
L = zultra_amd.lib()
d = corpus.text_like_fast(size, 1000)
for flags, bs in ((2, 65536), (1, 65536), (0, 65536), (2, 0), (1, 32768)):
    best = None
    for it in range(3):
        t0 = time.perf_counter()
        out = L.memory_compress(d, flags, bs)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print("zultra_memory_compress flags=%d max_block=%d: %.1f MB/s (%d -> %d bytes)" % (flags, bs, size / best / 1e6, size, len(out)), flush=True)
