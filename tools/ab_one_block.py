import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, corpus
from zultra_amd._ffi import Lib
so=sys.argv[1]
L=Lib(so)
d=corpus.real_text(48944)
ctx=L.context(65536,1)
best=None
for it in range(20):
    ctx.compress_blocks(d,[(0,0,len(d))])
    t=ctx.timing()
    if best is None or t["total_ms"]<best["total_ms"]: best=t
print(os.path.basename(so), " ".join("%s=%.3f"%(k[:-3],v) for k,v in best.items() if v))
