import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import corpus, zultra_amd
L = zultra_amd.lib()
size=100_000_000; bs=65536
d = corpus.text_like_fast(size, 1000)
nb = (size + bs - 1)//bs
blocks = [(b*bs - (32768 if b else 0), 32768 if b else 0, min(bs, size-b*bs)) for b in range(nb)]
ctx = L.context(bs, nb)
for stop in (1,2,3,4,5,0):
    os.environ["ZH_MF_STOP"] = str(stop)
    for it in range(2):
        try: ctx.compress_blocks(d, blocks)
        except Exception as e: pass
    print("stop", stop, "group_ms %.3f" % ctx.timing()["group_ms"], flush=True)
