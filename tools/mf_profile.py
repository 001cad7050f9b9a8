#!/usr/bin/env python3
"""Diagnostics: wave-cycles of zh_mf_frontier by phase, from a profiling build of the library (-DZH_MF_PROFILE, built into build/).
usage: python tools/mf_profile.py --build            (here, no GPU needed)
       python tools/mf_profile.py [bytes] [corpus]   (on the GPU box)"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
SO = os.path.join(ROOT, "build", "libzultra_amd_mfprof.so")
CSRC = os.path.join(ROOT, "zultra_amd", "csrc")
if "--build" in sys.argv:
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip", "-DZH_MF_PROFILE=1", "-I", CSRC, "-o", SO,
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
f = L.L.zultra_hip_mf_profile
f.argtypes = [C.c_void_p, C.c_int]
f(None, 1)
ctx.compress_blocks(d, blocks)
out = np.zeros(16, dtype=np.uint64)
f(out.ctypes.data, 0)
t = ctx.timing()
names = ["window staging", "chunk head", "byte-run path", "class walk", "row store"]
tot = float(out[:5].sum())
print("%s %d bytes: frontier %.2f ms (both runs), group %.2f ms" % (kind, size, t["frontier_ms"], t["group_ms"]))
for i, n in enumerate(names):
    print("   %-16s %5.1f %%   %.0f wave-cycles per position" % (n, 100.0 * float(out[i]) / tot, float(out[i]) * 64 / size))
o = [float(x) for x in out]
steps = max(1.0, o[8])
print("   chunks %d, walk steps per chunk %.1f, lanes alive per step %.1f" % (o[6], o[8] / max(1.0, o[6]), o[9] / steps))
print("   per step: %.0f cycles to the probes' answer, %.0f cycles of verification (%.2f of the steps have one: %.0f cycles each), %.2f whole-wave extensions"
      % (o[10] / steps, o[11] / steps, o[12] / steps, o[11] / max(1.0, o[12]), o[13] / steps))
rp = max(1.0, o[5] + o[7] + o[14] + o[15])
print("   byte-run path, lane 0's share by part: own run %.0f %%, earlier runs (same byte) %.0f %%, second search %.0f %%, runs followed by the same byte %.0f %%"
      % (100 * o[5] / rp, 100 * o[7] / rp, 100 * o[14] / rp, 100 * o[15] / rp))
