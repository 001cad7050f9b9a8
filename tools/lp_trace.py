#!/usr/bin/env python3
"""Diagnostics: when do the waves of zh_parse_lanes start, take their group of tasks and finish? (probe build -DZH_LP_TRACE, built into build/)
usage: python tools/lp_trace.py --build            (here, no GPU needed)
       python tools/lp_trace.py [bytes] [corpus]   (on the GPU box; ZULTRA_HIP_STREAMS=1 for one run alone on the chip: the trace holds the last launch of every pass)"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
SO = os.path.join(ROOT, "build", "libzultra_amd_lptrace.so")
CSRC = os.path.join(ROOT, "zultra_amd", "csrc")
if "--build" in sys.argv:
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip", "-DZH_LP_TRACE=1", "-DZH_TUNING_KNOBS=1", "-I", CSRC, "-o", SO,
                    os.path.join(CSRC, "zh_device.hip"), os.path.join(CSRC, "libzultra.cpp")], check=True)
    sys.exit(0)
import numpy as np  # noqa: E402

import corpus  # noqa: E402
from zultra_amd._ffi import Lib  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 33_554_432
kind = sys.argv[2] if len(sys.argv) > 2 else "pysrc"
L = Lib(SO)
d = corpus.real_text(size) if kind == "pysrc" else corpus.text_like_fast(size, 1000) if kind == "text" else corpus.mixed_config4(0, size >> 20)
size, bs = len(d), 65536
nb = (size + bs - 1) // bs
blocks = [(b * bs - (32768 if b else 0), 32768 if b else 0, min(bs, size - b * bs)) for b in range(nb)]
ctx = L.context(bs, nb)
f = L.L.zultra_hip_lp_trace
f.argtypes = [C.c_void_p, C.POINTER(C.c_uint)]
slots = C.c_uint()
f(None, C.byref(slots))
tr = np.zeros((4, slots.value, 4), dtype=np.uint64)
for it in range(3):
    ctx.compress_blocks(d, blocks)
    f(tr.ctypes.data, C.byref(slots))
print(ctx.timing())
print(ctx.stats())
for p in range(4):
    t = tr[p]
    t = t[t[:, 2] > 0]
    if not len(t):
        continue
    born, took, done = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64), t[:, 2].astype(np.int64)
    z = int(born.min())
    born, took, done = (born - z) / 100.0, (took - z) / 100.0, (done - z) / 100.0
    dur = done - took
    hw = t[:, 3] & np.uint64(0xffffffff)
    blk = (t[:, 3] >> np.uint64(32)).astype(np.int64)
    cu = ((hw >> np.uint64(8)) & np.uint64(15)).astype(np.int64) | (((hw >> np.uint64(13)) & np.uint64(7)).astype(np.int64) << 4)   # cu_id | se_id << 4 (within the XCD)
    print("pass %d: %d groups; waves born at %.0f..%.0f us (median %.0f); tickets taken %.0f..%.0f (median %.0f); done %.0f..%.0f (median %.0f) us" % (
        p, len(t), born.min(), born.max(), np.median(born), took.min(), took.max(), np.median(took), done.min(), done.max(), np.median(done)))
    print("   group duration us: min %.0f median %.0f p90 %.0f p99 %.0f max %.0f; distinct waves (blockIdx) %d; groups per wave max %d" % (
        dur.min(), np.median(dur), np.percentile(dur, 90), np.percentile(dur, 99), dur.max(), len(np.unique(blk)), np.bincount(blk - blk.min()).max()))
    h, _ = np.histogram(done, bins=10, range=(0, done.max()))
    print("   groups finishing per tenth of the pass:", h.tolist())
    h, _ = np.histogram(took, bins=10, range=(0, done.max()))
    print("   tickets taken per tenth of the pass:   ", h.tolist())
    late = np.argsort(-done)[:5]
    for i in late:
        print("      latest: ticket %5d born %.0f took %.0f done %.0f (%.0f us) hw 0x%x" % (i, born[i], took[i], done[i], dur[i], int(hw[i])))
