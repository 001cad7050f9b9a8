#!/usr/bin/env python3
"""Diagnostics: files mode (bench.py's configuration 5: 4 KiB JSON-like inputs, batches of 65 536, graphs replayed) through a given build of the library — batches per second
over a few hundred thousand inputs, kernels + stitch + read-back into pinned memory. One build per process (tools/r06_abf.sh alternates them).
usage: python tools/ab_files.py <lib.so> [inputs] [batch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import corpus  # noqa: E402
from zultra_amd._ffi import Lib  # noqa: E402

so = sys.argv[1]
nfiles = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
size = 4096
host = corpus.json_files(0, nfiles, size)
import torch  # noqa: E402
L = Lib(so, allow_missing=("zultra_hip_stitch_with_batch",))
d = torch.from_numpy(host).cuda()
torch.cuda.synchronize()
ctx = L.files_context(size, batch)
pinned = torch.empty(batch * (size + 64), dtype=torch.uint8, pin_memory=True).numpy()
sizes = np.full(batch, size, dtype=np.uint32)
nb = nfiles // batch


def sweep():
    tot = 0
    for b in range(nb):
        offs = (np.arange(batch, dtype=np.uint64) + np.uint64(b * batch)) * np.uint64(size)
        fo = ctx.compress_files(d.data_ptr(), offs, sizes, data_on_device=True, data_size=d.numel())
        ctx.stream_read(int(fo[-1]), out=pinned)
        tot += int(fo[-1])
    return tot


sweep()
ts = []
for r in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = sweep()
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
print("%-36s files/s: best %.0f median %.0f | ms per batch of %d: %.2f | %d bytes out" % (os.path.basename(so), nfiles / min(ts), nfiles / float(np.median(ts)), batch, min(ts) / nb * 1e3, out))
