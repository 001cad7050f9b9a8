#!/usr/bin/env python3
"""Timing experiment: zh_mf_group alone (ZH_MF_STOP = stage to stop after, zh_matchfinder.h), on the bench's real text.
usage: ZH_MF_STOP=5 python tools/probes/mf_stop_probe.py [bytes] [path of an alternative library build]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpus  # noqa: E402
from zultra_amd._ffi import Lib, ZultraError  # noqa: E402
import zultra_amd  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
L = Lib(sys.argv[2]) if len(sys.argv) > 2 else zultra_amd.lib()
d = corpus.real_text(size)
bs = 65536
nb = (size + bs - 1) // bs
blocks = [(b * bs - (32768 if b else 0), 32768 if b else 0, min(bs, size - b * bs)) for b in range(nb)]
ctx = L.context(bs, nb)
for it in range(3):
    try:
        ctx.compress_blocks(d, blocks)
    except ZultraError as e:
        pass
    print("stop=%s group_ms %.3f" % (os.environ.get("ZH_MF_STOP", "0"), ctx.timing()["group_ms"]))
