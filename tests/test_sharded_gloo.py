"""CPU, world_size 2, gloo: the N>1 path of bench.py / zultra_amd.sharded — max-blocks sharded over ranks, descriptor
all-gather, per-rank stitch at the true bit phase, byte gather to rank 0 — must reproduce the single-stream bytes.
Compute runs on the CPU emulator build of the kernels (tests/emu); the collectives are real torch.distributed calls."""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r); sys.path.insert(0, %(here)r); sys.path.insert(0, os.path.join(%(here)r, "emu"))
import build_emu, corpus, zlibs
from zultra_amd._ffi import Lib
from zultra_amd import sharded

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
E = Lib(build_emu.build())
bs = 32768
t = corpus.text_like(40000, 5)
data = np.concatenate([t[:30000], corpus.noise(6000, 3), t[30000:31000]])   # 37000 B: 2 max-blocks (one per rank), a stored sub-block across the cut
n = len(data)
nb = (n + bs - 1) // bs
lo, hi = sharded.shard_range(nb, rank, world)
blocks = []
first = lo * bs - (32768 if lo else 0)
for b in range(lo, hi):
    prev = 32768 if b else 0
    blocks.append((b * bs - prev - first, prev, min(bs, n - b * bs)))
ctx = E.context(bs, hi - lo)
ctx.compress_blocks(data[first:min(n, hi * bs)], blocks)
stream, info = sharded.assemble(E, ctx, bs, dist, torch, torch.device("cpu"), is_stream_end_rank=(rank == world - 1), nblocks_local=hi - lo)
if rank == 0:
    want = zlibs.Oracle().memory_compress(data, 0, bs)
    got = stream.tobytes()
    assert got == want, (len(got), len(want))
    crc = 0
    print("SHARDED_OK", len(got))
dist.destroy_process_group()
'''


def test_two_rank_assembly_matches_single_stream(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "here": HERE})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=900)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "SHARDED_OK" in outs[0], outs[0]


def test_shard_range_partitions_blocks():
    from zultra_amd import sharded
    for nb in (1, 2, 7, 1526, 131072):
        for world in (1, 2, 4, 8):
            cuts = [sharded.shard_range(nb, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == nb
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
