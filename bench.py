#!/usr/bin/env python3
"""bench.py — BASELINE.json metric: input MB/s of the per-block deflate hot path on MI355X, gzip, 64 KiB max-blocks,
enwik8-sized text (config[1]); one JSON line on rank 0.

A "step" = one pass of the whole job over this rank's 100 MB shard, input already resident in HBM:
   stage 1-3 kernels (match rows, token chain + splitter, sub-block coder: 4 x (task-parallel optimal parse, code
   rebuild), literalisation, emission) -> per-sub-block bit strings
   -> device stitch (zh_stitch) at the shard's true bit offset, per-block CRC-32 on the device (zh_crc32_blocks)
   -> (N>1: descriptor all-gather and byte gather over RCCL) -> D2H of the finished deflate bytes on rank 0
   -> gzip stream on rank 0 (header, deflate bits, CRC-32/ISIZE footer).
N>1 is weak scaling: every rank compresses its own 100 MB shard of one N x 100 MB stream.

Extra objects on the line: `roofline` for the dominant kernel (live HIP-event duration on the library's stream),
`cpu_baseline` = the compiled reference (oracle/_ref, kind "reference") or the oracle port, timed on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

# the library's streams need their own hardware queues (zh_device.hip: zh_runtime_hints); torch initialises HIP before the
# library is loaded here, so the hint has to be in the environment already
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def committed_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/*_traffic.json, written by
    tools/pmc_traffic.py from separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same command), or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        t = json.load(f)
    k = t.get("kernels", {}).get(kernel) or t.get("kernels", {}).get(kernel.split("+")[0])   # a timed group is priced by its main kernel
    if not k:
        return None, os.path.basename(files[-1])
    return int(k["hbm_bytes_per_launch"]), os.path.basename(files[-1])


def find_enwik8():
    for p in (os.environ.get("ZULTRA_ENWIK8"), os.path.join(ROOT, "data", "enwik8"), "/data/enwik8", os.path.expanduser("~/enwik8")):
        if p and os.path.exists(p) and os.path.getsize(p) == 100_000_000:
            return p
    return None


def make_shard(size, rank):
    """This rank's shard of the stream plus the 32 KiB that precede it (history of its first max-block)."""
    import corpus
    p = find_enwik8()
    if p and rank == 0:
        return np.fromfile(p, dtype=np.uint8)[:size], "enwik8"
    return corpus.text_like_fast(size, seed=1000 + rank), "synthetic"


def cpu_baseline(sample, flags, bs):
    """Reference CPU path (or the oracle port when oracle/_ref did not travel) on a bounded sample, 1 thread."""
    import zlibs
    if zlibs.have_ref():
        impl, kind = zlibs.Ref(), "reference"
    else:
        impl, kind = zlibs.Oracle(), "port"
    best = None
    out = None
    for _ in range(2 if kind == "reference" else 1):
        t0 = time.perf_counter()
        out = impl.memory_compress(sample, flags, bs)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return {"value": round(len(sample) / best / 1e6, 3), "unit": "MB/s", "cores": 1, "kind": kind,
            "sample": "first %d bytes of rank 0's shard, zultra_memory_compress gzip %d-byte blocks, best of %d, %s" % (
                len(sample), bs, 2 if kind == "reference" else 1,
                "compiled reference oracle/_ref" if kind == "reference" else "oracle/zultra_oracle.c")}, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=100_000_000, help="bytes per GPU (enwik8 = 100 000 000)")
    ap.add_argument("--block", type=int, default=65536)
    ap.add_argument("--cpu-sample", type=int, default=32 << 20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-run", action="store_true",
                    help="for rocprofv3 passes: only the steps (no round-trip / ratio / CPU extras that would add dispatches), then the PMC calibration probe")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import zultra_amd
    from zultra_amd import sharded

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    L = zultra_amd.lib()   # raises if libzultra_amd.so is missing: there is no fallback path
    if L.device_count() < 1:
        raise RuntimeError("no HIP device")

    bs, flags = args.block, 2
    shard, data_kind = make_shard(args.size, rank)
    n = len(shard)
    # history of this shard's first max-block = tail of the previous rank's shard (one continuous stream)
    if rank > 0:
        import corpus
        prev_tail = corpus.text_like_fast(args.size, seed=1000 + rank - 1)[-32768:]
        host = np.concatenate([prev_tail, shard])
        lead = 32768
    else:
        host, lead = shard, 0
    nblocks = (n + bs - 1) // bs
    blocks = []
    for b in range(nblocks):
        prev = 32768 if (b > 0 or lead) else 0
        blocks.append((lead + b * bs - prev, prev, min(bs, n - b * bs)))
    raw_offs = [b * bs for b in range(nblocks)]
    block_lens = np.array([b[2] for b in blocks], dtype=np.uint32)

    d_data = torch.from_numpy(host).to(device)   # input resident in HBM before the timed region
    torch.cuda.synchronize()
    ctx = L.context(bs, nblocks, device=local_rank)

    class OneRank:   # N == 1: same code path without a process group
        @staticmethod
        def get_rank():
            return 0

        @staticmethod
        def get_world_size():
            return 1

    D = dist if world > 1 else OneRank
    timings = []

    def step():
        ctx.compress_blocks(d_data.data_ptr(), blocks, data_on_device=True, data_size=d_data.numel())
        timings.append(ctx.timing())
        # gzip footer CRC-32: per-max-block values computed on the device next to the compression, folded on the host
        crc = L.crc32_append_many(0, ctx.block_crc32(), block_lens)
        body, info = sharded.assemble(L, ctx, bs, D, torch, device, is_stream_end_rank=(rank == world - 1), nblocks_local=nblocks)
        timings[-1]["stitch_ms"] = ctx.timing()["stitch_ms"]
        return body, crc

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    timings.clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        body, crc = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        ms = dt / args.steps * 1e3
        total_in = n * world
        # per-kernel device times (HIP events on the library stream), averaged over the timed steps
        avg = {k: float(np.mean([t[k] for t in timings])) for k in timings[0]}
        kernels = {"zh_mf_group": avg["group_ms"], "zh_mf_frontier": avg["frontier_ms"],
                   "zh_barriers+zh_tokenize_spans+zh_split": avg["tokenize_split_ms"], "zh_plan_subblocks+zh_sb_init": avg["init_ms"],
                   "zh_parse_tasks+zh_parse_huge": avg["parse_ms"], "zh_sb_build": avg["build_ms"], "zh_post_tasks": avg["post_ms"],
                   "zh_emit_tasks": avg["emit_ms"], "zh_stitch": avg["stitch_ms"]}
        # the library runs a batch as `runs` staggered runs of max-blocks on separate streams (ZULTRA_HIP_STREAMS, default 2):
        # every kernel is launched once per run (the parse / code-rebuild pair once per pass and run) over 1/runs of the batch
        runs = max(1, min(4, int(os.environ.get("ZULTRA_HIP_STREAMS", "2"))))
        if nblocks < 4 * runs or n < (runs << 22):
            runs = 1
        launches = {k: runs for k in kernels}
        launches["zh_parse_tasks+zh_parse_huge"] = launches["zh_sb_build"] = 4 * runs
        launches["zh_stitch"] = 1
        dom = max(kernels, key=lambda k: kernels[k])
        out_bytes = len(body) / world
        # 1 B read + r B written per input byte; one launch covers 1/runs of the batch
        alg_bytes = (n + out_bytes) / (runs if dom != "zh_stitch" else 1)
        launch_ms = kernels[dom] / launches[dom]
        achieved = alg_bytes / (launch_ms * 1e-3) / 1e9
        traffic, traffic_src = committed_traffic(dom)
        line = {
            "metric": "input MB/s, gzip 64 KiB max-blocks, enwik8-sized text, bit-exact vs CPU zultra",
            "value": round(total_in / (dt / args.steps) / 1e6, 3), "unit": "MB/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/int32", "data": data_kind,
            "config": {"workload": "%s %d B per GPU, gzip (nFlags=2), nMaxBlockSize=%d, %d max-blocks per GPU" % (
                "enwik8" if data_kind == "enwik8" else "enwik8-sized seeded text-like synthetic (enwik8 absent)", n, bs, nblocks),
                "parallelism": "blocks sharded over %d GPU(s), descriptor all-gather + byte gather to rank 0" % world},
            "kernel_ms": {k: round(v, 3) for k, v in kernels.items()},
            "device_pipeline_ms": round(avg["total_ms"], 3), "d2h_ms": round(avg["d2h_ms"], 3),
            "kernel_only_MBps": round(n / (sum(kernels.values()) * 1e-3) / 1e6, 3),
            "compressed_bytes_per_gpu": int(out_bytes),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 4), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": int(alg_bytes), "launch_ms": round(launch_ms, 3),
                         "launches_per_step": launches[dom]},
        }
        if args.profile_run:
            L.traffic_probe(256 << 20)
            print(json.dumps(line), flush=True)
            ctx.close()
            return
        # outside the timed region: the stream must inflate to the input, and match zlib-9 ratio expectations
        import zlib
        hdr = bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 2, 255])
        footer = int(crc).to_bytes(4, "little") + int(n & 0xffffffff).to_bytes(4, "little")
        body = body.tobytes()
        if world == 1:
            gz = hdr + body + footer
            ok = zlib.decompress(gz, 31) == shard.tobytes()
            line["inflate_roundtrip_ok"] = bool(ok)
            z9 = len(zlib.compress(shard[: 8 << 20].tobytes(), 9))
            mine = len(L.memory_compress(shard[: 8 << 20], 2, bs))
            line["size_vs_zlib9_first_8MiB"] = round(mine / z9, 5)
        else:
            d = zlib.decompressobj(-15)
            first = d.decompress(body, n)   # rank 0's shard must come back exactly
            line["inflate_roundtrip_ok"] = bool(first == shard.tobytes())
        if not args.no_cpu_baseline and world == 1:   # reported at N=1 only
            sample = shard[: min(args.cpu_sample, n)]
            cb, ref_out = cpu_baseline(sample, flags, bs)
            line["cpu_baseline"] = cb
            gpu_out = L.memory_compress(sample, flags, bs)
            line["bit_exact_vs_cpu_on_sample"] = bool(gpu_out == ref_out)
        print(json.dumps(line), flush=True)

    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
