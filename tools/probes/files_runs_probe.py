#!/usr/bin/env python3
"""Debugging aid: a files-mode batch of n inputs in ZULTRA_HIP_STREAMS runs, checked by inflating a sample.
usage: ZULTRA_HIP_STREAMS=3 python tools/probes/files_runs_probe.py n [device-resident 0|1]"""
import os
import sys
import zlib
import faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import corpus  # noqa: E402
import zultra_amd  # noqa: E402

n = int(sys.argv[1])
on_device = len(sys.argv) > 2 and sys.argv[2] == "1"
if on_device:   # (torch's HIP runtime has to be up before the library's: the other order leaves torch without devices)
    import torch
    torch.zeros(1, device="cuda:0")
L = zultra_amd.lib()
size = 4096
data = corpus.json_files(0, n, size)
offs = np.arange(n, dtype=np.uint64) * size
sizes = [size] * n
if on_device:
    d = torch.from_numpy(data).to("cuda:0")
    torch.cuda.synchronize()
ctx = L.files_context(size, n)
for rep in range(3):
    if on_device:
        fo = ctx.compress_files(d.data_ptr(), offs, np.array(sizes, dtype=np.uint32), data_on_device=True, data_size=d.numel())
    else:
        fo = ctx.compress_files(data, offs, sizes)
    st = ctx.stream_read(int(fo[-1]))
    for k in range(0, n, max(1, n // 50)):
        assert zlib.decompress(st[int(fo[k]):int(fo[k + 1])].tobytes(), -15) == data[k * size:(k + 1) * size].tobytes(), k
    print("rep", rep, "ok", ctx.stats(), flush=True)
