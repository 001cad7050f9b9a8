"""Multi-GPU assembly: max-blocks shard across ranks with no exchange during compute; the only
collectives are (1) an all-gather of the small per-sub-block descriptors, from which every rank derives the bit
offset at which its shard starts in the stream (the stored-vs-compressed decision of libzultra.c:345-347 depends on
the running bit phase, so offsets come from a dry run of the stitch planner over the preceding shards), and (2) one
variable-length gather of the stitched shard bytes — stitched on each GPU by the zh_stitch kernel at the shard's true
bit phase — to rank 0, which ORs the shared boundary bytes together.

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


class _DevBuf:
    """Expose a raw HIP device pointer to torch through __cuda_array_interface__ (no copy)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def _stream_tensor(ctx, torch, device, nbytes):
    ptr = ctx.stream_ptr()
    if device.type == "cuda":
        return torch.as_tensor(_DevBuf(ptr, nbytes), device=device)
    buf = (C.c_uint8 * nbytes).from_address(ptr)   # CPU emulator build: the "device" buffer is host memory
    return torch.from_numpy(np.ctypeslib.as_array(buf))


def _plan_bits(lib, subs_arr, count, max_block, phase):
    """Dry run of the stitcher over one shard's descriptors: (whole bytes, trailing bits)."""
    st = BitState(0, phase)
    dummy = (C.c_uint64 * 1)(0)
    sp = subs_arr.ctypes.data_as(C.POINTER(SubBlock))
    w = lib.L.zultra_hip_stitch(C.byref(st), sp, count, None, None, dummy, max_block, -1, None, 0)
    if w == _SIZE_MAX:
        raise RuntimeError("stitch planning failed (ZULTRA_ERROR_DST)")
    return w, st.nacc


_pinned = {}


def _to_host(torch, t):
    """Device tensor -> numpy through a reused pinned staging buffer (pageable D2H is several times slower)."""
    if t.device.type != "cuda":
        return t.numpy()
    n = t.numel()
    buf = _pinned.get("buf")
    if buf is None or buf.numel() < n:
        buf = _pinned["buf"] = torch.empty(max(n, 1 << 20) * 5 // 4, dtype=torch.uint8, pin_memory=True)
    buf[:n].copy_(t)
    return buf[:n].numpy()


def assemble(lib, ctx, max_block, dist, torch, device, is_stream_end_rank, nblocks_local):
    """Stitch this rank's last batch on its GPU at its true bit offset and gather the stream on rank 0.
    Returns (stream bytes as a uint8 numpy array on rank 0 / None elsewhere, info dict)."""
    rank, world = dist.get_rank(), dist.get_world_size()
    p, cnt = ctx.subblocks_raw()
    rec = C.sizeof(SubBlock)
    mine = np.zeros((cnt, rec), dtype=np.uint8)
    C.memmove(mine.ctypes.data, p, cnt * rec)

    start_phase = 0
    allsubs, counts = None, [cnt]
    if world > 1:
        # (1) descriptors of every rank (48 B per sub-block)
        ct = torch.zeros(world, dtype=torch.int64, device=device)
        ct[rank] = cnt
        dist.all_reduce(ct)
        counts = [int(x) for x in ct.cpu()]
        pad = np.zeros((max(counts), rec), dtype=np.uint8)
        pad[:cnt] = mine
        send = torch.from_numpy(pad).to(device)
        allsubs = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(allsubs, send)
        allsubs = [t.cpu().numpy() for t in allsubs]
        phase = 0
        for r in range(rank):
            _, phase = _plan_bits(lib, np.ascontiguousarray(allsubs[r][:counts[r]]), counts[r], max_block, phase)
        start_phase = phase

    # (2) stitch on the device at the shard's true phase: byte 0 carries only this shard's bits
    end_bit, _ = ctx.stitch_device(nblocks_local - 1 if is_stream_end_rank else -1, phase=start_phase)
    nbytes = (end_bit + 7) // 8
    local = _stream_tensor(ctx, torch, device, nbytes)

    if world == 1:
        return _to_host(torch, local), {"shard_bytes": nbytes, "start_phase": 0}

    # (3) variable-length gather of the stitched bytes to rank 0 (RCCL over xGMI on the GPU box)
    lt = torch.zeros(world, dtype=torch.int64, device=device)
    lt[rank] = nbytes
    dist.all_reduce(lt)
    lens = [int(x) for x in lt.cpu()]
    maxl = max(lens)
    sendb = torch.zeros(maxl, dtype=torch.uint8, device=device)
    sendb[:nbytes] = local
    if rank != 0:
        dist.gather(sendb, None, dst=0)
        return None, {"shard_bytes": nbytes, "start_phase": start_phase}
    parts = [torch.empty(maxl, dtype=torch.uint8, device=device) for _ in range(world)]
    dist.gather(sendb, parts, dst=0)

    # rank 0: join on the device (a shard that starts mid-byte shares that byte with its predecessor), then one pinned D2H
    total = sum(lens)
    stream = torch.zeros(total + 1, dtype=torch.uint8, device=device)
    pos = 0
    phase = 0
    for r in range(world):
        b = parts[r][:lens[r]]
        if phase:   # first byte overlaps the previous shard's last (partial) byte
            stream[pos - 1] |= b[0]
            stream[pos:pos + lens[r] - 1] = b[1:]
            pos += lens[r] - 1
        else:
            stream[pos:pos + lens[r]] = b
            pos += lens[r]
        _, phase = _plan_bits(lib, np.ascontiguousarray(allsubs[r][:counts[r]]), counts[r], max_block, phase)
    return _to_host(torch, stream[:pos]), {"shard_bytes": nbytes, "start_phase": 0}
