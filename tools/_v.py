import glob, os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import corpus
from zultra_amd._ffi import Lib
d = corpus.text_like_fast(100_000_000, 1000)
bs = 65536
lib = sys.argv[1]
L = Lib(lib)
size = len(d); nb = (size + bs - 1) // bs
blocks = [(b * bs - (32768 if b else 0), 32768 if b else 0, min(bs, size - b * bs)) for b in range(nb)]
ctx = L.context(bs, nb)
for it in range(2):
    try: ctx.compress_blocks(d, blocks)
    except Exception as e: pass
t = ctx.timing()
print(os.path.basename(lib), "group %.2f" % t["group_ms"], flush=True)
