#!/usr/bin/env python3
"""Diagnostics: where does zh_encode spend its shader clocks? Runs one batch with the in-kernel phase profile on
(zultra_hip_set_profile) and prints per-phase cycle totals plus the critical (longest) sub-block."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpus  # noqa: E402
import zultra_amd  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
bs = 65536
L = zultra_amd.lib()
d = corpus.text_like_fast(size, 1000)
nb = (size + bs - 1) // bs
blocks = [(b * bs - (32768 if b else 0), 32768 if b else 0, min(bs, size - b * bs)) for b in range(nb)]
ctx = L.context(bs, nb)
ctx.set_profile(True)
for it in range(2):
    ctx.compress_blocks(d, blocks)
print("timing", ctx.timing())
P = ctx.profile().astype(np.int64)
subs, _, cnt = ctx.subblocks()
sizes = np.array([s.size for s in subs])
dyn = np.array([s.is_dynamic for s in subs])
names = ["cost_eval", "tentative_codes", "parse0", "hist+codes0", "parse1", "hist+codes1", "parse2", "hist+codes2", "parse3",
         "hist+codes3", "literalize", "alt_tables", "header+masks", "tokens"]
D = np.diff(P[:, :15], axis=1)
D = np.where(dyn[:, None] == 1, D, 0)
tot = D.sum(axis=0)
print("sub-blocks %d (dynamic %d), size min/mean/max %d/%d/%d" % (cnt, dyn.sum(), sizes.min(), sizes.mean(), sizes.max()))
print("%-16s %14s %7s %12s" % ("phase", "sum cycles", "share", "cyc/byte"))
for k, nme in enumerate(names):
    print("%-16s %14d %6.1f%% %12.2f" % (nme, tot[k], 100.0 * tot[k] / tot.sum(), tot[k] / sizes[dyn == 1].sum()))
print("pass-3 parse: step-loop clocks %d of %d (%.1f%%)" % (P[:, 15].sum(), D[:, 8].sum(), 100.0 * P[:, 15].sum() / D[:, 8].sum()))
life = P[:, 14] - P[:, 0]
worst = int(np.argmax(life))
print("longest sub-block: size %d, %d cycles; phases:" % (sizes[worst], life[worst]), dict(zip(names, D[worst].tolist())))
span = P[:, 14].max() - P[:, 0].min()
print("kernel span in shader clocks: %d ; sum of lifetimes / span = %.1f waves busy on average" % (span, life.sum() / span))
