import os, sys, time, threading
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import corpus, zultra_amd
L = zultra_amd.lib()
size=100_000_000; bs=65536
d = corpus.text_like_fast(size, 1000)
dd = torch.from_numpy(d).cuda(); torch.cuda.synchronize()
nb = (size + bs - 1)//bs
blocks = [(b*bs - (32768 if b else 0), 32768 if b else 0, min(bs, size-b*bs)) for b in range(nb)]
def run(K):
    parts = [blocks[(nb*k)//K:(nb*(k+1))//K] for k in range(K)]
    ctxs = [L.context(bs, len(p)) for p in parts]
    def work(k):
        ctxs[k].compress_blocks(dd.data_ptr(), parts[k], data_on_device=True, data_size=dd.numel())
    best = 1e9
    for it in range(4):
        torch.cuda.synchronize(); t0=time.perf_counter()
        th=[threading.Thread(target=work,args=(k,)) for k in range(K)]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize(); dt=time.perf_counter()-t0
        best=min(best,dt)
    print("streams %d: %.2f ms  (sum of per-ctx total_ms %.2f)" % (K, best*1e3, sum(c.timing()["total_ms"] for c in ctxs)), flush=True)
    for c in ctxs: c.close()
for K in (1,2,3,4): run(K)
