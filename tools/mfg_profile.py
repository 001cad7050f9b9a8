#!/usr/bin/env python3
"""Diagnostics: cycles of zh_mf_group by phase (thread 0 of every workgroup), from a profiling build of the library (-DZH_MFG_PROFILE, built into build/).
usage: python tools/mfg_profile.py --build            (here, no GPU needed)
       python tools/mfg_profile.py [bytes] [corpus]   (on the GPU box)"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
SO = os.path.join(ROOT, "build", "libzultra_amd_mfgprof.so")
CSRC = os.path.join(ROOT, "zultra_amd", "csrc")
if "--build" in sys.argv:
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip", "-DZH_MFG_PROFILE=1", "-I", CSRC, "-o", SO,
                    os.path.join(CSRC, "zh_device.hip"), os.path.join(CSRC, "libzultra.cpp")], check=True)
    sys.exit(0)
import numpy as np  # noqa: E402

import corpus  # noqa: E402
from zultra_amd._ffi import Lib  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "pysrc"
L = Lib(SO)
d = corpus.real_text(size) if kind == "pysrc" else corpus.text_like_fast(size, 1000) if kind == "text" else corpus.mixed_config4(0, size >> 20)
size, bs = len(d), 65536
nb = (size + bs - 1) // bs
blocks = [(b * bs - (32768 if b else 0), 32768 if b else 0, min(bs, size - b * bs)) for b in range(nb)]
ctx = L.context(bs, nb)
ctx.compress_blocks(d, blocks)
f = L.L.zultra_hip_mfg_profile
f.argtypes = [C.c_void_p, C.c_int]
f(None, 1)
ctx.compress_blocks(d, blocks)
out = np.zeros(32, dtype=np.uint64)
f(out.ctypes.data, 0)
t = ctx.timing()
names = ["window staging + ticket", "two passes through HBM", "chunk load + boundary", "four passes of a chunk", "sweep", "oversized classes", "run starts", "first run order",
         "run lengths", "second run order", "second table"]
o = [float(x) for x in out]
tot = sum(o[:11])
segs = max(1.0, o[15])
print("%s %d bytes: group %.2f ms (sum over the runs), %d segments, %.1f chunks and %.2f oversized classes (%.0f entries) per segment" % (kind, size, t["group_ms"], segs, o[12] / segs, o[13] / segs, o[14] / segs))
for i, n in enumerate(names):
    print("   %-26s %5.1f %%   %8.0f cycles per segment" % (n, 100.0 * o[i] / tot, o[i] / segs))
print("   total %.0f cycles per segment" % (tot / segs))
inner = ["counting", "wait A", "totals", "wait B", "scan + places", "wait C", "scatter", "wait D"]
npass = max(1.0, 4.0 * o[12])
print("   inside a pass of a chunk (thread 0), cycles per pass: " + ", ".join("%s %.0f" % (n, o[16 + i] / npass) for i, n in enumerate(inner)))
