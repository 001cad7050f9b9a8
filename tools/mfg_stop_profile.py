#!/usr/bin/env python3
"""Diagnostics: zh_mf_group alone (ZH_MF_STOP) with its phase profile, for a given -DZH_MFG_PROFILE build.
usage: ZH_MF_STOP=5 python tools/mfg_stop_profile.py <lib.so> [bytes] [corpus]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import corpus  # noqa: E402
from zultra_amd._ffi import Lib, ZultraError  # noqa: E402

L = Lib(sys.argv[1])
size = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000_000
kind = sys.argv[3] if len(sys.argv) > 3 else "pysrc"
d = corpus.real_text(size) if kind == "pysrc" else corpus.text_like_fast(size, 1000) if kind == "text" else corpus.mixed_config4(0, size >> 20)
size, bs = len(d), 65536
nb = (size + bs - 1) // bs
blocks = [(b * bs - (32768 if b else 0), 32768 if b else 0, min(bs, size - b * bs)) for b in range(nb)]
ctx = L.context(bs, nb)
f = L.L.zultra_hip_mfg_profile
f.argtypes = [C.c_void_p, C.c_int]
for it in range(2):
    f(None, 1)
    try:
        ctx.compress_blocks(d, blocks)
    except ZultraError:
        pass
out = np.zeros(32, dtype=np.uint64)
f(out.ctypes.data, 0)
o = [float(x) for x in out]
segs = max(1.0, o[15])
npass = max(1.0, 4.0 * o[12])
print("%s %s stop=%s group_ms %.3f | per segment: hbm passes %.0f, load+boundary %.0f, passes %.0f, sweep %.0f, oversized %.0f | per pass: count %.0f waitA %.0f totals %.0f waitB %.0f scan %.0f waitC %.0f scatter %.0f waitD %.0f" % (
    os.path.basename(sys.argv[1]), kind, os.environ.get("ZH_MF_STOP", "0"), ctx.timing()["group_ms"], o[1] / segs, o[2] / segs, o[3] / segs, o[4] / segs, o[5] / segs,
    *[o[16 + i] / npass for i in range(8)]))
