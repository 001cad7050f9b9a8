"""Multi-GPU assembly: max-blocks shard across ranks with no exchange during compute. What crosses
ranks after the compute:

  (1) one all-gather of a 128-byte *phase table* per rank. The stored-vs-compressed decision of libzultra.c:345-347
      compares whole flushed bytes, so the bit length of a shard depends on the bit phase (0..7) it starts at. Every rank
      dry-runs the stitch planner over its own sub-block descriptors for the eight possible start phases and publishes
      (bits written, touched) per phase; chaining the tables of the ranks before it gives a rank its true start phase and
      byte offset without any descriptor leaving its rank.
  (2) each rank stitches its shard on its own GPU (zh_stitch) at that phase: byte 0 of its local stream carries only this
      shard's bits.
  (3) one gather of each shard's first byte, and exact-length point-to-point transfers (batch_isend_irecv: RCCL
      send/recv over the direct xGMI link to rank 0 on the GPU box) straight into their final place in rank 0's stream
      buffer. A shard that starts mid-byte shares that byte with its predecessor: it sends its bytes from the second one
      on, and rank 0 ORs the two halves of the shared byte together from the boundary records.

`dist` is torch.distributed (backend "nccl" = RCCL on the GPU box, "gloo" in the CPU tests).
"""
import ctypes as C
import os

import numpy as np

from ._ffi import BitState, SubBlock

_SIZE_MAX = C.c_size_t(-1).value


def shard_range(nblocks, rank, world):
    """Contiguous max-block range of a rank (empty when world > nblocks for some ranks)."""
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
    if "emu" in os.path.basename(ctx.lib.path):
        buf = (C.c_uint8 * nbytes).from_address(ptr)   # CPU emulator build: the "device" buffer is host memory
        return torch.from_numpy(np.ctypeslib.as_array(buf))
    return torch.from_numpy(ctx.stream_read(nbytes).copy())   # real device, collectives on CPU tensors (gloo tests): through the host


def phase_table(lib, ctx, max_block):
    """int64[8, 2]: for start phase p, (end_bit, ok) of this rank's last batch, end_bit counted from the start of the byte
    that holds the p pending bits (zh_stitch_plan's origin). An empty shard leaves the phase alone: end_bit = p."""
    tab = np.zeros((8, 2), dtype=np.int64)
    if ctx is None:
        tab[:, 0] = np.arange(8)
        tab[:, 1] = 1
        return tab
    # the device scan of the stitcher walks the shard's descriptors once for all eight start phases (zh_stitch_scan): no host planning
    ends = (C.c_uint64 * 8)()
    failed = C.c_uint32(0)
    lib.L.zultra_hip_stitch_phase_table.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    rc = lib.L.zultra_hip_stitch_phase_table(ctx.h, ends, C.byref(failed))
    if rc != 0:
        raise RuntimeError("zultra_hip_stitch_phase_table: %d %s" % (rc, lib.L.zultra_hip_last_error(ctx.h).decode()))
    for ph in range(8):
        if (failed.value >> ph) & 1:
            tab[ph] = (ph, 0)   # the reference fails with ZULTRA_ERROR_DST at this phase; only an error if it is the true one
        else:
            tab[ph] = (int(ends[ph]), 1)
    return tab


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


def _to_host_async(torch, t):
    """Starts the copy of a device tensor into the reused pinned buffer and returns (numpy view, wait): the caller does other host work, then calls wait()."""
    if t.device.type != "cuda":
        return t.numpy(), (lambda: None)
    n = t.numel()
    buf = _pinned.get("buf")
    if buf is None or buf.numel() < n:
        buf = _pinned["buf"] = torch.empty(max(n, 1 << 20) * 5 // 4, dtype=torch.uint8, pin_memory=True)
    buf[:n].copy_(t, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return buf[:n].numpy(), ev.synchronize


def assemble(lib, ctx, max_block, dist, torch, device, final_block_local, extra=None, force_collectives=False):
    """Stitch this rank's last batch on its GPU at its true bit offset and gather the stream on rank 0.
    ctx = None for a rank whose shard is empty (it still takes part in the collectives).
    final_block_local = index (in this rank's batch) of the last max-block of the whole stream, or -1.
    extra = a small int64 vector per rank (same length everywhere) that rides along with the phase tables, e.g. the shard's
    checksum contribution; info["extras"] holds every rank's, in rank order.
    force_collectives: a world of ONE rank normally returns before the first collective (there is nothing to exchange); with this flag
    it walks the same path as N > 1 — all_gather of the phase table, gather of the first byte, the slice views of the stream buffer —
    so that a 1-GPU box executes those calls on device tensors over RCCL (tests, `bench.py --gpus 1 --scaling strong`).
    info["collective_ms"] = host wall time of the exchange steps (1) and (3) on this rank.
    Returns (stream bytes as a uint8 numpy array on rank 0 / None elsewhere, info dict)."""
    import time
    rank, world = dist.get_rank(), dist.get_world_size()

    if world == 1 and not force_collectives:
        # (`extra` may be a callable here: what the caller folds on the host — the shard's checksums — goes on while the stitched bytes come back)
        end_bit, _ = ctx.stitch_device(final_block_local, phase=0)
        nbytes = (end_bit + 7) // 8
        body, wait = _to_host_async(torch, _stream_tensor(ctx, torch, device, nbytes))
        if callable(extra):
            extra = extra()
        extra = np.zeros(0, dtype=np.int64) if extra is None else np.ascontiguousarray(extra, dtype=np.int64)
        wait()
        return body, {"shard_bytes": nbytes, "start_phase": 0, "sent_bytes": 0, "extras": [extra], "collective_ms": 0.0}
    if callable(extra):
        extra = extra()
    extra = np.zeros(0, dtype=np.int64) if extra is None else np.ascontiguousarray(extra, dtype=np.int64)

    # (1) phase tables of every rank -> start phase and byte offset of every shard
    t_coll = time.perf_counter()
    mine = torch.from_numpy(np.concatenate([phase_table(lib, ctx, max_block).reshape(-1), extra])).to(device)
    tabs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(tabs, mine)
    tabs = [t.cpu().numpy() for t in tabs]
    extras = [t[16:] for t in tabs]
    tabs = [t[:16].reshape(8, 2) for t in tabs]
    phase, off = 0, 0            # off = index of the byte that holds the pending bits (or the next free byte at phase 0)
    starts = []
    for r in range(world):
        end_bit, ok = int(tabs[r][phase][0]), int(tabs[r][phase][1])
        if not ok:
            raise RuntimeError("stream assembly overflows the reference's per-block buffer bound on rank %d (ZULTRA_ERROR_DST)" % r)
        starts.append((phase, off, end_bit))
        off += end_bit >> 3
        phase = end_bit & 7
    total_bytes = off + (1 if phase else 0)
    my_phase, my_off, my_end = starts[rank]
    coll_s = time.perf_counter() - t_coll

    # (2) stitch on the device at the shard's true phase: byte 0 carries only this shard's bits
    nbytes = 0
    local = None
    if ctx is not None:
        end_bit, _ = ctx.stitch_device(final_block_local, phase=my_phase)
        assert end_bit == my_end, (end_bit, my_end)
        nbytes = (end_bit + 7) // 8
        local = _stream_tensor(ctx, torch, device, nbytes)

    # (3) first byte of every shard, then exact-length transfers into place on rank 0
    t_coll = time.perf_counter()
    edge = torch.zeros(1, dtype=torch.uint8, device=device)
    if nbytes:
        edge[0] = local[0]
    edges = [torch.empty_like(edge) for _ in range(world)] if rank == 0 else None
    dist.gather(edge, edges, dst=0)

    def touched(r):   # bytes shard r owns in the stream, and how many of them it transfers (all but a shared first byte)
        ph, o, eb = starts[r]
        n = (eb + 7) // 8 if eb != ph else 0   # an empty shard writes nothing
        skip = 1 if (n and ph) else 0
        return o, n, skip

    info = {"shard_bytes": nbytes, "start_phase": my_phase, "sent_bytes": 0, "extras": extras}
    if rank != 0:
        o, n, skip = touched(rank)
        if n - skip > 0:
            reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, local[skip:n], 0)])
            for q in reqs:
                q.wait()
            info["sent_bytes"] = n - skip
        info["collective_ms"] = (coll_s + time.perf_counter() - t_coll) * 1e3
        return None, info

    stream = torch.zeros(total_bytes + 1, dtype=torch.uint8, device=device)
    ops = []
    for r in range(1, world):
        o, n, skip = touched(r)
        if n - skip > 0:
            ops.append(dist.P2POp(dist.irecv, stream[o + skip:o + n], r))
    reqs = dist.batch_isend_irecv(ops) if ops else []
    if nbytes:
        stream[:nbytes] = local   # rank 0's own shard (phase 0, offset 0), while the transfers are in flight
    for q in reqs:
        q.wait()
    # shared bytes: a shard that starts mid-byte ORs its first byte into the byte its predecessors left partial
    edges = [e.cpu().numpy() for e in edges]
    for r in range(1, world):
        o, n, skip = touched(r)
        if skip:
            stream[o] |= int(edges[r][0])
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    info["collective_ms"] = (coll_s + time.perf_counter() - t_coll) * 1e3
    info["received_bytes"] = int(sum(max(0, touched(r)[1] - touched(r)[2]) for r in range(1, world)))
    return _to_host(torch, stream[:total_bytes]), info
