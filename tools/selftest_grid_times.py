#!/usr/bin/env python3
"""Diagnostics: device times by kernel group on the reference's self-test data (tool/zultra.c:425-463: a given alphabet size and match probability), 32 MiB each:
which corners of configuration 4's grid the matchfinder's class walk spends its time in. usage: python tools/selftest_grid_times.py [lib.so]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import corpus  # noqa: E402
import zultra_amd  # noqa: E402
from zultra_amd._ffi import Lib  # noqa: E402

import torch  # noqa: E402
torch.zeros(1).cuda()   # (torch first: it does not find the device behind the library's own initialisation)
L = Lib(sys.argv[1]) if len(sys.argv) > 1 else zultra_amd.lib()
size, bs = 32 << 20, 65536
nb = size // bs
blocks = [(b * bs - (32768 if b else 0), 32768 if b else 0, bs) for b in range(nb)]
ctx = L.context(bs, nb)
for nlit in (2, 3, 15, 96, 256):
    for prob in (0.0, 0.5, 0.9):
        d = np.concatenate([corpus.selftest_data(1 << 20, 1000 * nlit + k, nlit, prob) for k in range(size >> 20)])
        dd = torch.from_numpy(d).cuda()
        for _ in range(2):
            ctx.compress_blocks(dd.data_ptr(), blocks, data_on_device=True, data_size=dd.numel())
        t = ctx.timing()
        print("alphabet %3d match probability %.1f: total %7.2f ms | group %6.2f frontier %7.2f tok+split %6.2f parse %7.2f" % (
            nlit, prob, t["total_ms"], t["group_ms"], t["frontier_ms"], t["tokenize_split_ms"], t["parse_ms"]), flush=True)
