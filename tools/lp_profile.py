#!/usr/bin/env python3
"""Diagnostics: step and lane utilisation of zh_parse_lanes, from a profiling build of the library (-DZH_LP_PROFILE, built into build/).
usage: python tools/lp_profile.py --build            (here, no GPU needed)
       python tools/lp_profile.py [bytes] [corpus]   (on the GPU box)
       python tools/lp_profile.py [inputs] files     (files mode: that many 4 KiB inputs in one batch)"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
SO = os.environ.get("LP_LIB") or os.path.join(ROOT, "build", "libzultra_amd_lpprof.so")   # (LP_LIB: another profiling build)
CSRC = os.path.join(ROOT, "zultra_amd", "csrc")
if "--build" in sys.argv:
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip", "-DZH_LP_PROFILE=1", "-I", CSRC, "-o", SO,
                    os.path.join(CSRC, "zh_device.hip"), os.path.join(CSRC, "libzultra.cpp")], check=True)
    sys.exit(0)
import numpy as np  # noqa: E402

import corpus  # noqa: E402
from zultra_amd._ffi import Lib  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "pysrc"
L = Lib(SO)
if kind == "files":   # files mode: `size` inputs of 4096 bytes (configuration 5's JSON-like files), one batch
    nfiles, fsz = size, 4096
    d = corpus.json_files(0, nfiles, fsz)
    ctx = L.files_context(fsz, nfiles)
    offs = np.arange(nfiles, dtype=np.uint64) * np.uint64(fsz)
    sizes = np.full(nfiles, fsz, dtype=np.uint32)
    ctx.compress_files(d, offs, sizes)
    f = L.L.zultra_hip_lp_profile
    f.argtypes = [C.c_void_p, C.c_int]
    f(None, 1)
    ctx.compress_files(d, offs, sizes)
    out = np.zeros(16, dtype=np.uint64)
    f(out.ctypes.data, 0)
    o = [float(x) for x in out]
    print("files %d x %d bytes" % (nfiles, fsz))
    print("   groups %d, pieces per group %.1f, steps per group %.1f" % (o[3], o[6] / max(1, o[3]), o[0] / max(1, o[3])))
    print("   quads with a position per step %.1f of 16" % (o[1] / max(1, o[0])))
    print("   cycles per step %.0f; per group: setup %.0f, steps %.0f, histogram %.0f cycles" % (o[2] / max(1, o[0]), o[4] / max(1, o[3]), o[2] / max(1, o[3]), o[5] / max(1, o[3])))
    print("   histogram phase per group: waiting for the group's stores + clearing %.0f, the walks %.0f, storing the counters %.0f cycles" % (o[12] / max(1, o[3]), o[13] / max(1, o[3]), (o[5] - o[12] - o[13]) / max(1, o[3])))
    sys.exit(0)
d = corpus.real_text(size) if kind == "pysrc" else corpus.text_like_fast(size, 1000) if kind == "text" else corpus.mixed_config4(0, size >> 20)
size, bs = len(d), 65536
nb = (size + bs - 1) // bs
blocks = [(b * bs - (32768 if b else 0), 32768 if b else 0, min(bs, size - b * bs)) for b in range(nb)]
ctx = L.context(bs, nb)
ctx.compress_blocks(d, blocks)
f = L.L.zultra_hip_lp_profile
f.argtypes = [C.c_void_p, C.c_int]
f(None, 1)
ctx.compress_blocks(d, blocks)
out = np.zeros(16, dtype=np.uint64)
f(out.ctypes.data, 0)
t = ctx.timing()
o = [float(x) for x in out]
print("%s %d bytes: parse %.2f ms (all runs, 4 passes)" % (kind, size, t["parse_ms"]))
print("   groups %d, pieces per group %.1f, steps per group %.1f" % (o[3], o[6] / max(1, o[3]), o[0] / max(1, o[3])))
print("   quads with a position per step %.1f of 16, batches with a second plane %.2f" % (o[1] / max(1, o[0]), 4 * o[7] / max(1, o[0])))
print("   cycles per step %.0f; per group: setup %.0f, steps %.0f, histogram %.0f cycles" % (o[2] / max(1, o[0]), o[4] / max(1, o[3]), o[2] / max(1, o[3]), o[5] / max(1, o[3])))
print("   histogram phase per group: waiting for the group's stores + clearing %.0f, the walks %.0f, storing the counters %.0f cycles" % (o[12] / max(1, o[3]), o[13] / max(1, o[3]), (o[5] - o[12] - o[13]) / max(1, o[3])))
lap = max(1.0, o[8] + o[9] + o[10] + o[11])
print("   step loop by part (s_memtime laps, which drain the LDS queue: shares, not cycles): staging a batch %.0f %%, stage C (decision) %.0f %%, stage B (window, prefix minima) %.0f %%, stage A (slots, prices) %.0f %%"
      % (100 * o[8] / lap, 100 * o[9] / lap, 100 * o[10] / lap, 100 * (o[11]) / lap))
