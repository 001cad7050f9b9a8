#!/usr/bin/env python3
"""Diagnostics: device pipeline time of one 100 MB batch (first launch to last completion, HIP events) under different settings of the library's
tuning knobs (environment variables read at context creation). The shipped library has its tuning knobs compiled in: build a probe library first —
tools/build_variant.sh knobs -DZH_TUNING_KNOBS — and name it in KNOB_LIB=build/libzultra_amd_knobs.so.
usage: KNOB_LIB=... [KNOB_BLOCK=32768] python tools/knob_sweep.py [bytes] [pysrc|text|binary|mixed] -- KEY=VAL,KEY=VAL ... (one group per setting)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import corpus  # noqa: E402
import zultra_amd  # noqa: E402

args = sys.argv[1:]
sep = args.index("--") if "--" in args else len(args)
size = int(args[0]) if sep > 0 else 100_000_000
kind = args[1] if sep > 1 else "pysrc"
settings = args[sep + 1:] or [""]
def _binaries(size):   # configuration 3's stand-in (bench.py: binary_corpus)
    import glob
    parts, total = [], 0
    for f in sorted(glob.glob("/usr/lib/x86_64-linux-gnu/*.so*")) + sorted(glob.glob("/usr/bin/*")):
        if os.path.isfile(f) and not os.path.islink(f):
            try:
                b = np.fromfile(f, dtype=np.uint8)
            except OSError:
                continue
            parts.append(b)
            total += b.size
            if total >= size:
                break
    return np.concatenate(parts)[:size].copy()


d = corpus.real_text(size) if kind == "pysrc" else corpus.text_like_fast(size, 1000) if kind == "text" else _binaries(size) if kind == "binary" else corpus.mixed_config4(0, size >> 20)
size, bs = len(d), int(os.environ.get("KNOB_BLOCK", "65536"))
nb = (size + bs - 1) // bs
blocks = [(b * bs - (32768 if b else 0), 32768 if b else 0, min(bs, size - b * bs)) for b in range(nb)]
import torch  # noqa: E402
if os.environ.get("KNOB_LIB"):   # another build of the library (tools/build_variant.sh)
    from zultra_amd._ffi import Lib
    L = Lib(os.environ["KNOB_LIB"])
else:
    L = zultra_amd.lib()
dd = torch.from_numpy(d).cuda()
torch.cuda.synchronize()
prev_keys = []
for st in settings:
    for k in prev_keys:
        os.environ.pop(k, None)
    prev_keys = []
    for kv in [x for x in st.split(",") if x]:
        k, v = kv.split("=")
        os.environ[k] = v
        prev_keys.append(k)
    ctx = L.context(bs, nb)
    ts = []
    for it in range(7):
        try:
            ctx.compress_blocks(dd.data_ptr(), blocks, data_on_device=True, data_size=dd.numel())
        except Exception:   # (debug builds that skip kernels produce streams that fail later checks)
            pass
        ts.append(ctx.timing()["total_ms"])
    t = ctx.timing()
    print("%-60s total min %.2f med %.2f | group %.1f frontier %.1f tok+split %.1f parse %.1f build %.1f post %.1f emit %.1f" % (
        st or "(defaults)", min(ts[2:]), float(np.median(ts[2:])), t["group_ms"], t["frontier_ms"], t["tokenize_split_ms"], t["parse_ms"], t["build_ms"], t["post_ms"], t["emit_ms"]), flush=True)
    ctx.close()
