"""Multi-GPU assembly: max-blocks shard across ranks with no exchange during compute; the only
collectives are (1) an all-gather of the small per-sub-block descriptors, from which every rank derives the bit
offset at which its shard starts in the stream (the stored-vs-compressed decision of libzultra.c:345-347 depends on
the running bit phase, so offsets come from a dry run of the stitcher over the preceding shards), and (2) one
variable-length gather of the stitched shard bytes to rank 0, which ORs the shared boundary bytes together.

`dist` is torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
"""
import ctypes as C

import numpy as np

from ._ffi import BitState, SubBlock

_SIZE_MAX = C.c_size_t(-1).value


def shard_range(nblocks, rank, world):
    """Contiguous max-block range of a rank."""
    lo = (nblocks * rank) // world
    hi = (nblocks * (rank + 1)) // world
    return lo, hi


def _subs_to_array(p, cnt):
    a = np.zeros((cnt, C.sizeof(SubBlock)), dtype=np.uint8)
    if cnt:
        C.memmove(a.ctypes.data, p, cnt * C.sizeof(SubBlock))
    return a


def assemble(lib, ctx, raw_shard, raw_offs, max_block, dist, torch, device, is_stream_end_rank):
    """Stitch this rank's last batch at its true bit offset and gather the stream on rank 0.

    raw_shard : uint8 array with the raw bytes of this rank's max-blocks (for stored sub-blocks)
    raw_offs  : offset of each max-block inside raw_shard
    Returns (stream_bytes or None on ranks > 0, dict of sizes).
    """
    rank, world = dist.get_rank(), dist.get_world_size()
    subs, p, cnt = ctx.subblocks()
    mine = _subs_to_array(p, cnt)
    rec = C.sizeof(SubBlock)

    # (1) descriptors of every rank (tiny: 48 B per sub-block)
    counts = torch.zeros(world, dtype=torch.int64, device=device)
    counts[rank] = cnt
    dist.all_reduce(counts)
    counts = [int(x) for x in counts.cpu()]
    maxc = max(counts)
    pad = np.zeros((maxc, rec), dtype=np.uint8)
    pad[:cnt] = mine
    send = torch.from_numpy(pad).to(device)
    allsubs = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(allsubs, send)

    # bit offset of this shard = dry run of the stitcher over the shards before it
    st = BitState(0, 0)
    start_bytes = 0
    nblocks_dummy = (C.c_uint64 * 1)(0)
    for r in range(rank):
        a = allsubs[r].cpu().numpy()[:counts[r]].copy()
        sp = a.ctypes.data_as(C.POINTER(SubBlock))
        w = lib.L.zultra_hip_stitch(C.byref(st), sp, counts[r], None, None, nblocks_dummy, max_block, -1, None, 0)
        if w == _SIZE_MAX:
            raise RuntimeError("stitch planning failed")
        start_bytes += w
    start_phase = st.nacc

    # (2) local stitch at the true phase: first byte carries only our bits
    size = C.c_size_t()
    payload = lib.L.zultra_hip_payload(ctx.h, C.byref(size))
    raw = np.ascontiguousarray(raw_shard, dtype=np.uint8)
    offs = (C.c_uint64 * len(raw_offs))(*raw_offs)
    st2 = BitState(0, start_phase)
    cap = len(raw) + 64 * 6 * max(1, len(raw_offs)) + 1024
    out = np.zeros(cap, dtype=np.uint8)
    final_block = len(raw_offs) - 1 if is_stream_end_rank else -1
    w = lib.L.zultra_hip_stitch(C.byref(st2), p, cnt, payload, raw.ctypes.data, offs, max_block, final_block, out.ctypes.data, cap)
    if w == _SIZE_MAX:
        raise RuntimeError("stitch failed")
    if st2.nacc:   # trailing partial byte: high bits belong to the next shard (or are padding at the stream end)
        out[w] = st2.acc & ((1 << st2.nacc) - 1)
        w += 1

    # (3) variable-length gather of the stitched bytes to rank 0 (RCCL on the GPU box)
    lens = torch.zeros(world, dtype=torch.int64, device=device)
    lens[rank] = w
    dist.all_reduce(lens)
    lens = [int(x) for x in lens.cpu()]
    maxl = max(lens)
    sendb = torch.zeros(maxl, dtype=torch.uint8, device=device)
    sendb[:w] = torch.from_numpy(out[:w]).to(device)
    if rank == 0:
        parts = [torch.empty(maxl, dtype=torch.uint8, device=device) for _ in range(world)]
        dist.gather(sendb, parts, dst=0)
    else:
        dist.gather(sendb, None, dst=0)
        return None, {"shard_bytes": w, "start_bit": start_bytes * 8 + start_phase}

    # rank 0: concatenate; a shard that starts mid-byte shares that byte with its predecessor
    stream = bytearray()
    bitpos = 0
    for r in range(world):
        b = parts[r][:lens[r]].cpu().numpy()
        if len(b) == 0:
            continue
        if bitpos & 7:
            stream[-1] |= int(b[0])
            stream += b[1:].tobytes()
        else:
            stream += b.tobytes()
        # advance by the exact bit length of shard r: dry run again (cheap) to know its end phase
        st3 = BitState(0, bitpos & 7)
        a = allsubs[r].cpu().numpy()[:counts[r]].copy()
        wbytes = lib.L.zultra_hip_stitch(C.byref(st3), a.ctypes.data_as(C.POINTER(SubBlock)), counts[r], None, None,
                                         nblocks_dummy, max_block, -1, None, 0)
        bitpos = (bitpos & ~7) + wbytes * 8 + st3.nacc
    return bytes(stream), {"shard_bytes": w, "start_bit": 0}
