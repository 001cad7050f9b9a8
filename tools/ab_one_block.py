import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, corpus
from zultra_amd._ffi import Lib
so=sys.argv[1]
L=Lib(so)
size=int(sys.argv[2]) if len(sys.argv)>2 else 48944
d=corpus.real_text(size)
nb=(size+65535)//65536
blocks=[(b*65536-(32768 if b else 0), 32768 if b else 0, min(65536,size-b*65536)) for b in range(nb)]
ctx=L.context(65536,nb)
best=None
for it in range(20):
    ctx.compress_blocks(d,blocks)
    t=ctx.timing()
    if best is None or t["total_ms"]<best["total_ms"]: best=t
print(os.path.basename(so), size, " ".join("%s=%.3f"%(k[:-3],v) for k,v in best.items() if v))
