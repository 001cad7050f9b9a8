import glob, os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import corpus
from zultra_amd._ffi import Lib
def pysrc(size):
    buf = bytearray()
    for f in sorted(glob.glob("/usr/lib/python3*/**/*.py", recursive=True)) + sorted(glob.glob("/usr/local/lib/python3*/**/*.py", recursive=True)):
        try: buf += open(f, "rb").read()
        except OSError: pass
        if len(buf) >= size: break
    return np.frombuffer(bytes(buf[:size]), dtype=np.uint8).copy()
L = Lib(os.path.join(ROOT, "build/variants/lib_sp.so"))
f = L.L.zultra_hip_debug_split_profile
f.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
bs = 65536
names = ["total histogram", "total cost", "interval stats", "trigger scan", "eval histograms (+wait at end)", "eval 2 x dynamic cost", "barrier wait", "selection"]
for name, d in (("pysrc", pysrc(50_000_000)), ("mixed", np.concatenate([corpus.mixed(1 << 22, 5 + k) for k in range(12)])), ("text", corpus.text_like_fast(100_000_000, 1000))):
    size = len(d); nb = (size + bs - 1) // bs
    blocks = [(b * bs - (32768 if b else 0), 32768 if b else 0, min(bs, size - b * bs)) for b in range(nb)]
    ctx = L.context(bs, nb)
    ctx.compress_blocks(d, blocks)
    out = (C.c_ulonglong * 16)()
    f(out, 1)
    ctx.compress_blocks(d, blocks)
    f(out, 1)
    t = ctx.timing()
    tot = sum(out[:8])
    print(name, "split stream time %.1f ms; wave-cycles by phase:" % t["tokenize_split_ms"])
    for i in range(8):
        print("   %-34s %6.1f %%  (%.1f ms per wave-slot if spread over %d blocks x 8 waves)" % (names[i], 100.0 * out[i] / tot, out[i] / 2.4e6 / (nb * 8), nb))
    ctx.close()
