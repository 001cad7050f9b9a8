"""CPU, gloo, world_size 2 / 3 / 4 / 8: the N>1 path of bench.py / zultra_amd.sharded — max-blocks sharded over ranks, phase-table
all-gather, per-rank stitch at the true bit phase, exact-length transfers to rank 0 — must reproduce the single-stream
bytes. Compute runs on the CPU emulator build of the kernels (tests/emu); the collectives are real torch.distributed calls.

Inputs put incompressible bytes right behind every shard cut, so that the first sub-block of a shard is a stored one that
starts at whatever bit phase the shards before it leave (its padding depends on that phase)."""
import os
import subprocess
import sys

import pytest

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
nblocks_full, tail = %(nfull)d, %(tail)d
data = corpus.text_like(nblocks_full * bs + tail, 5)
for b in range(1, nblocks_full + 1):
    cut = b * bs
    k = min(12000, len(data) - cut)
    data[cut:cut + k] = corpus.noise(k, cut)          # stored sub-block right behind every max-block boundary
n = len(data)
nb = (n + bs - 1) // bs
lo, hi = sharded.shard_range(nb, rank, world)
ctx = None
if hi > lo:
    first = lo * bs - min(32768, lo * bs)
    blocks = []
    for b in range(lo, hi):
        prev = min(32768, b * bs)
        blocks.append((b * bs - prev - first, prev, min(bs, n - b * bs)))
    ctx = E.context(bs, hi - lo)
    ctx.compress_blocks(data[first:min(n, hi * bs)], blocks)
final_local = (nb - 1 - lo) if (hi == nb and hi > lo) else -1
stream, info = sharded.assemble(E, ctx, bs, dist, torch, torch.device("cpu"), final_local)
phases = [None] * world
dist.all_gather_object(phases, (info["start_phase"], info["shard_bytes"]))
if rank == 0:
    want = zlibs.Oracle().memory_compress(data, 0, bs)
    got = stream.tobytes()
    assert got == want, (len(got), len(want), phases)
    print("SHARDED_OK", len(got), phases)
dist.destroy_process_group()
'''


def _run(tmp_path, world, nfull, tail, port):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "here": HERE, "nfull": nfull, "tail": tail})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world))
    procs = []
    for r in range(world):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=1200)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "SHARDED_OK" in outs[0], outs[0]
    return outs[0]


def test_two_rank_assembly_matches_single_stream(tmp_path):
    _run(tmp_path, 2, 1, 5000, 29533)          # 2 max-blocks, one per rank


def test_four_rank_assembly_matches_single_stream(tmp_path):
    out = _run(tmp_path, 4, 3, 5000, 29534)    # 4 max-blocks, one per rank, a stored sub-block behind every cut
    assert "SHARDED_OK" in out


def test_more_ranks_than_blocks(tmp_path):
    _run(tmp_path, 3, 1, 5000, 29535)          # 2 max-blocks over 3 ranks: rank 0 (the gather root) has an empty shard


def test_eight_rank_assembly_matches_single_stream(tmp_path):
    """The size of the node BASELINE.json's metric is quoted on: nine max-blocks over eight ranks (one rank takes two), a stored sub-block behind every
    cut, every rank's shard starting at the bit phase the seven tables before it imply."""
    out = _run(tmp_path, 8, 8, 5000, 29536)
    assert "SHARDED_OK" in out


def test_shard_range_partitions_blocks():
    from zultra_amd import sharded
    for nb in (1, 2, 7, 1526, 131072):
        for world in (1, 2, 3, 4, 8):
            cuts = [sharded.shard_range(nb, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == nb
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
