#!/usr/bin/env python3
"""bench.py — the BASELINE.json metric on MI355X: input MB/s of zultra's per-block deflate hot path, bit-exact with the
CPU reference, one JSON line on rank 0.

    python bench.py [--config 1|2|3|4|5] [--gpus N] [--steps K] [--warmup W]

--config selects one of BASELINE.json's configurations (default 2, the one the metric is quoted on):
  1  bootstrap.min.js (the image's 39 680-byte v3.4.1 stands in), raw deflate, one max-block: known answer 10 523 B
  2  enwik8-sized text, gzip, 64 KiB max-blocks. enwik8 is not in the image: `value` is measured on REAL text (the image's
     Python sources, 100 MB), the kind seeded synthetic text of round 1 is reported beside it as `synthetic_text`
  3  silesia/mozilla-sized binary (shared libraries of the image stand in), zlib framing, 32 KiB max-blocks, ratio vs zlib-9
  4  the 8 GiB synthetic mixed-entropy corpus, 64 KiB max-blocks, 1 GiB (16 384 max-blocks) per GPU
  5  1 000 000 x 4 KiB JSON-like inputs per GPU, each its own gzip stream, hipGraph-replayed batches

A "step" = one pass of the whole job over this rank's shard, input already resident in HBM: stage 1-3 kernels -> device
stitch at the shard's true bit offset + per-block checksums on the device -> (N>1: phase-table all-gather and exact-length
transfers to rank 0 over RCCL) -> D2H of the finished deflate bytes on rank 0 -> frame on rank 0.
--gpus N > 1: this process spawns N rank processes itself (one per GPU, RCCL) before anything touches a GPU — or runs as
one rank when a launcher (torch.distributed.run) already set RANK/WORLD_SIZE; a WORLD_SIZE that differs from --gpus is an
error, and so are fewer visible GPUs than N. Weak scaling: every rank takes its own shard of ONE N-times-larger stream.

Every line carries `roofline` (dominant kernel, live HIP-event duration on the library's streams) and `cpu_baseline` (the
compiled reference oracle/_ref — or the oracle port where it did not travel — on a bounded sample, rank 0, N=1).
The process exits non-zero when the inflate round trip or the bit-exactness check fails.
"""
import argparse
import glob
import hashlib
import json
import os
import socket
import subprocess
import sys
import time
import zlib

# the library's streams need their own hardware queues (zh_device.hip: zh_runtime_hints); torch initialises HIP before the
# library is loaded here, so the hint has to be in the environment already
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured with a float4 copy)
HIST = 32768


# ---------------------------------------------------------------------------------------------------------------------
# launcher
# ---------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def count_gpus():
    """GPUs of this node from sysfs — no HIP / torch call, so the process that spawns the ranks never touches a GPU: KFD topology nodes
    with SIMDs (CPUs have simd_count 0), else the DRM render nodes."""
    n = 0
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            for line in open(f):
                if line.startswith("simd_count") and int(line.split()[1]) > 0:
                    n += 1
        except (OSError, ValueError):
            pass
    if n == 0:
        n = len(glob.glob("/dev/dri/renderD*"))
    vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
    if vis:
        n = min(n, len([v for v in vis.split(",") if v.strip() != ""]))
    return n


def spawn_ranks(args, argv):
    """--gpus N without a launcher: start N rank processes (from a process that has not touched a GPU) and watch them: when one
    fails the others would wait in the rendezvous or in a collective forever, so they are stopped and the failure is reported."""
    have = count_gpus()
    if have < args.gpus:
        sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) visible — refusing to report a smaller job as N=%d\n" % (args.gpus, have, args.gpus))
        return 2
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    out0 = os.path.join(os.environ.get("TMPDIR", "/tmp"), "zultra_bench_rank0.%d.out" % os.getpid())
    with open(out0, "wb") as f0:
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=f0 if r == 0 else subprocess.DEVNULL))
        bad = []
        while True:
            rcs = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(rcs) if c not in (None, 0)]
            if bad or all(c is not None for c in rcs):
                break
            time.sleep(0.2)
        if bad:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t_end = time.time() + 10
            for p in procs:
                try:
                    p.wait(timeout=max(0.1, t_end - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
    sys.stdout.write(open(out0).read())
    sys.stdout.flush()
    os.unlink(out0)
    if bad:
        sys.stderr.write("bench.py: rank(s) failed: %s (the other ranks were stopped)\n" % bad)
        return 1
    return 0


# ---------------------------------------------------------------------------------------------------------------------
# corpora: every rank can produce any range of ONE global stream
# ---------------------------------------------------------------------------------------------------------------------
def find_enwik8():
    for p in (os.environ.get("ZULTRA_ENWIK8"), os.path.join(ROOT, "data", "enwik8"), "/data/enwik8", os.path.expanduser("~/enwik8")):
        if p and os.path.exists(p) and os.path.getsize(p) == 100_000_000:
            return p
    return None


def _cyclic(base, start, size):
    n = len(base)
    start %= n
    parts = []
    while size > 0:
        k = min(size, n - start)
        parts.append(base[start:start + k])
        size -= k
        start = 0
    return np.concatenate(parts) if len(parts) > 1 else parts[0].copy()


def _files_corpus(patterns, need, cap=256 << 20):
    """Files of the image matching `patterns`, sorted by path and concatenated until `need` (at most `cap`) bytes are there."""
    parts, total = [], 0
    for pat in patterns:
        for f in sorted(glob.glob(pat, recursive=True)):
            if total >= min(need, cap):
                break
            try:
                if os.path.islink(f) or not os.path.isfile(f):
                    continue
                b = np.fromfile(f, dtype=np.uint8)
            except OSError:
                continue
            if b.size:
                parts.append(b)
                total += b.size
    if not parts:
        raise RuntimeError("no files found for %s" % (patterns,))
    return np.concatenate(parts)


class CyclicCorpus:
    """A byte corpus, cycled: windows are 32 KiB, so the tiling is invisible to the compressor."""

    def __init__(self, name, base):
        self.name, self.base = name, base

    def shard(self, rank, size):
        return self.range(rank * size, size)

    def range(self, start, size):
        """bytes [start, start + size) of the stream, and the 32 KiB in front of them (None at the stream's start)"""
        lead = _cyclic(self.base, start - HIST, HIST) if start else None
        return lead, _cyclic(self.base, start, size)


class SyntheticText:
    """Round 1's seeded Zipf word stream (tests/corpus.py: text_like_fast). Shard r ends with a 32 KiB piece of its own seed, so
    rank r+1 regenerates only those 32 KiB for its history."""
    name = "synthetic"

    def shard(self, rank, size):
        import corpus
        body = corpus.text_like_fast(size - HIST, seed=1000 + rank)
        tail = corpus.text_like_fast(HIST, seed=7000 + rank)
        lead = corpus.text_like_fast(HIST, seed=7000 + rank - 1) if rank else None
        return lead, np.concatenate([body, tail])


class MixedConfig4:
    name = "synthetic"

    def shard(self, rank, size):
        import corpus
        seg = corpus.CONFIG4_SEGMENT
        assert size % seg == 0
        first = rank * (size // seg)
        lead = corpus.mixed_config4(first - 1, 1)[-HIST:].copy() if rank else None
        return lead, corpus.mixed_config4(first, size // seg)

    def range(self, start, size):
        import corpus
        seg = corpus.CONFIG4_SEGMENT
        s0, s1 = start // seg, (start + size + seg - 1) // seg
        body = corpus.mixed_config4(s0, s1 - s0)[start - s0 * seg: start - s0 * seg + size].copy()
        lead = None
        if start:
            l0 = (start - HIST) // seg
            lead = corpus.mixed_config4(l0, (start + seg - 1) // seg - l0)[start - HIST - l0 * seg: start - l0 * seg].copy()
        return lead, body


def text_corpus(world, size):
    p = find_enwik8()
    if p:
        return CyclicCorpus("enwik8", np.fromfile(p, dtype=np.uint8)), "enwik8 (real file)"
    need = world * size
    base = _files_corpus(["/usr/lib/python3*/**/*.py", "/usr/local/lib/python3*/**/*.py"], need)
    return CyclicCorpus("real_text", base), "real text: the image's Python sources, sorted by path, cycled to size (enwik8 absent)"


def find_file(env, names, size):
    """A named corpus file when it is on the box: $env, ./data/<name>, /data/<name>, ~/<name> — of exactly `size` bytes."""
    for p in [os.environ.get(env)] + [os.path.join(d, n) for n in names for d in (os.path.join(ROOT, "data"), "/data", os.path.expanduser("~"))]:
        if p and os.path.isfile(p) and os.path.getsize(p) == size:
            return p
    return None


def binary_corpus(world, size):
    p = find_file("ZULTRA_MOZILLA", ["mozilla", "silesia/mozilla"], 51_220_480)
    if p:
        return CyclicCorpus("real_mozilla", np.fromfile(p, dtype=np.uint8)), "silesia/mozilla (real file)"
    base = _files_corpus(["/usr/lib/x86_64-linux-gnu/*.so*", "/usr/bin/*"], world * size)
    return CyclicCorpus("real_binary", base), "real binaries: shared libraries and executables of the image, sorted by path (silesia/mozilla absent)"


# ---------------------------------------------------------------------------------------------------------------------
# CPU side: reference baseline (before this process touches the GPU) and framing helpers
# ---------------------------------------------------------------------------------------------------------------------
def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_impl():
    import zlibs
    if zlibs.have_ref():
        return zlibs.Ref(), "reference", "compiled reference oracle/_ref"
    return zlibs.Oracle(), "port", "oracle/zultra_oracle.c"


def cpu_baseline_stream(sample, flags, bs, what):
    """Reference CPU path on a bounded sample, 1 thread, best of 3 -> (cpu_baseline object, compressed bytes)."""
    impl, kind, label = cpu_impl()
    runs = 3 if kind == "reference" else 1
    best, out = None, None
    for _ in range(runs):
        t0 = time.perf_counter()
        out = impl.memory_compress(sample, flags, bs)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return {"value": round(len(sample) / best / 1e6, 3), "unit": "MB/s", "cores": 1, "kind": kind, "nproc": os.cpu_count(), "cpu_model": cpu_model(),
            "sample": "%s: first %d bytes of rank 0's shard, zultra_memory_compress flags=%d max block %d, 1 thread, best of %d, %s" % (
                what, len(sample), flags, bs, runs, label)}, out


def _reference_check_worker(conn, data, flags, bs):
    """Child process (forked before the parent touches a GPU): the CPU reference over `data` in ONE call — the stream's bit phases carry from
    max-block to max-block, so a byte-for-byte comparison needs the whole stream — reported as (length, sha-256, seconds)."""
    try:
        impl, kind, label = cpu_impl()
        t0 = time.perf_counter()
        out = impl.memory_compress(data, flags, bs)
        conn.send((len(out), hashlib.sha256(out).hexdigest(), time.perf_counter() - t0, kind))
    except Exception as e:   # noqa: BLE001 (reported by the parent)
        conn.send((-1, repr(e), 0.0, "error"))
    conn.close()


def start_reference_check(data, flags, bs):
    """Starts the CPU reference on ALL of `data` in the background (one host core; the GPU legs run meanwhile). -> handle for finish_reference_check."""
    import multiprocessing as mp
    ctx = mp.get_context("fork")
    recv, send = ctx.Pipe(False)
    p = ctx.Process(target=_reference_check_worker, args=(send, data, flags, bs), daemon=True)
    p.start()
    send.close()
    return {"proc": p, "recv": recv, "bytes": len(data)}


def finish_reference_check(handle, gpu_stream_bytes):
    """-> fields for the JSON line: is the GPU's stream over the same bytes identical to the reference's, in full."""
    n, digest, secs, kind = handle["recv"].recv()
    handle["proc"].join()
    same = n == len(gpu_stream_bytes) and digest == hashlib.sha256(gpu_stream_bytes).hexdigest()
    return {"bit_exact_vs_reference_full": bool(same), "bit_exact_checked_input_bytes": handle["bytes"], "bit_exact_stream_bytes": int(n), "bit_exact_checker": kind,
            "reference_seconds_one_core": round(secs, 1)}, bool(same)


def _files_worker(job):
    data, size, flags = job
    impl, _, _ = cpu_impl()
    h = hashlib.sha256()
    t0 = time.perf_counter()
    for k in range(len(data) // size):
        h.update(impl.memory_compress(data[k * size:(k + 1) * size], flags, 0))
    return time.perf_counter() - t0, h.hexdigest()


def cpu_baseline_files(sample, size, flags, ntimed):
    """Config 5: the reference over a bounded sample of files — one thread on the first `ntimed` of them (the baseline's value), then nproc processes over
    shards of ALL the sample's files (independent inputs: still the reference's exact output; the all-cores rate, and the digests the GPU's streams are
    compared with). Must run before this process initialises the GPU (it forks). -> (cpu_baseline object, [digest per shard], files per shard)"""
    import multiprocessing as mp
    _, kind, label = cpu_impl()
    nfiles = len(sample) // size
    ntimed = min(ntimed, nfiles)
    t1, _ = _files_worker((sample[: ntimed * size], size, flags))
    nproc = os.cpu_count() or 1
    per = (nfiles + nproc - 1) // nproc
    jobs = [(sample[i * per * size:min(nfiles, (i + 1) * per) * size], size, flags) for i in range(nproc) if i * per < nfiles]
    with mp.get_context("fork").Pool(len(jobs)) as pool:
        res = pool.map(_files_worker, jobs)
    tn = max(t for t, _ in res)   # the slowest worker's own clock: process start-up is not compression time
    return {"value": round(ntimed / t1, 1), "unit": "files/s", "cores": 1, "kind": kind, "nproc": nproc, "cpu_model": cpu_model(),
            "all_cores_value": round(nfiles / tn, 1), "all_cores_processes": len(jobs),
            "sample": "first %d files of rank 0's shard, one zultra_memory_compress (gzip) per file, 1 thread; then %d processes over shards of its first %d files; %s" % (
                ntimed, len(jobs), nfiles, label)}, [d for _, d in res], per


def frame(L, flags, body, checksum, total_in):
    if flags == 2:
        return bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 2, 255]) + body + int(checksum).to_bytes(4, "little") + int(total_in & 0xffffffff).to_bytes(4, "little")
    if flags == 1:
        return bytes([0x78, 0xda]) + body + int(checksum).to_bytes(4, "big")
    return body


def inflate_check(flags, framed, first_shard, total_in):
    """Inflate the whole stream in chunks (zlib verifies the gzip CRC-32 / zlib Adler-32 footer over ALL ranks' bytes);
    rank 0's own shard must come back byte for byte and the total length must be the input's."""
    d = zlib.decompressobj({2: 31, 1: 15, 0: -15}[flags])
    got, ok = 0, True
    mv = memoryview(framed)
    try:
        for pos in range(0, len(mv), 8 << 20):
            out = d.decompress(mv[pos:pos + (8 << 20)])
            if got < len(first_shard):
                k = min(len(out), len(first_shard) - got)
                ok = ok and out[:k] == first_shard[got:got + k].tobytes()
            got += len(out)
        got += len(d.flush())
    except zlib.error:
        return False
    return bool(ok and d.eof and got == total_in)


def _csrc_digest():
    import zultra_amd
    return zultra_amd.csrc_digest()


def committed_traffic(kernel, config):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this configuration (profiles/*_traffic_c<N>.json,
    or an older *_traffic.json that names the configuration; written by tools/pmc_traffic.py from separate --pmc FETCH_SIZE /
    WRITE_SIZE runs of this same command) -> (bytes or None, file name or None, stale): stale = the file was measured on other sources than the
    ones in the tree now (its csrc_digest differs or is missing): the figure is quoted, but flagged."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic_c%s.json" % config)))
    if not files:
        files = [f for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json"))) if json.load(open(f)).get("config") == config]
    if not files:
        return None, None, None
    with open(files[-1]) as f:
        t = json.load(f)
    stale = t.get("csrc_digest") != _csrc_digest()
    ks = t.get("kernels", {})
    k = ks.get(kernel) or ks.get(kernel.split("+")[0]) or ks.get(kernel.split("+")[-1])   # a timed group is priced by its main kernel
    if kernel.startswith("zh_parse") and "zh_parse_lanes" in ks:   # the parse group: its kernels run side by side
        tot = sum(ks[n]["hbm_bytes_per_launch"] * ks[n]["launches"] for n in ("zh_parse_lanes", "zh_parse_chain", "zh_parse_segments", "zh_parse_tasks") if n in ks)
        return int(tot / max(1, ks["zh_parse_lanes"]["launches"])), os.path.basename(files[-1]), stale
    if not k:
        return None, os.path.basename(files[-1]), stale
    return int(k["hbm_bytes_per_launch"]), os.path.basename(files[-1]), stale


def committed_sq(config, ms_per_step):
    """Issue and residency figures from the committed SQ-counter pass of this configuration (profiles/*_sq_c<N>.json, tools/sq_profile.py):
    valu_issue_frac = the PROFILED step's vector instructions x 2 cycles / (1024 SIMDs x the cycles of the PROFILED step at 2.4 GHz) — numerator and
    denominator from the same pass — and the average number of resident waves of every kernel while it ran. `stale`: measured on other sources
    than the tree's. None when no pass is committed."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sq_c%s.json" % config)))
    if not files:
        return None
    with open(files[-1]) as f:
        t = json.load(f)
    issue_ms = float(t.get("valu_issue_ms_per_step", 0.0))
    step_ms = float(t.get("step_ms_profiled", 0.0)) or ms_per_step   # (files of round 4 did not record their own step: this run's, flagged stale anyway)
    return {"valu_issue_frac": round(issue_ms / step_ms, 4) if step_ms > 0 else None, "valu_issue_ms_per_step": issue_ms, "step_ms_profiled": t.get("step_ms_profiled"),
            "valu_insts_per_step": t.get("valu_insts_per_step"),
            "resident_waves_avg": {k: v.get("resident_waves_avg") for k, v in t.get("kernels", {}).items() if v.get("resident_waves_avg") is not None},
            "source": os.path.basename(files[-1]), "stale": bool(t.get("csrc_digest") != _csrc_digest())}


class OneRank:   # N == 1: same code path without a process group
    @staticmethod
    def get_rank():
        return 0

    @staticmethod
    def get_world_size():
        return 1


# ---------------------------------------------------------------------------------------------------------------------
# one leg of a stream configuration (2, 3, 4): compress this rank's shard `steps` times, assemble on rank 0
# ---------------------------------------------------------------------------------------------------------------------
def run_stream_leg(env, lead, shard, flags, bs, steps, warmup, last_rank=None):
    """last_rank: the rank that holds the stream's last max-block (strong scaling: ranks behind it have empty shards)."""
    L, torch, dist, device, rank, world = env["L"], env["torch"], env["dist"], env["device"], env["rank"], env["world"]
    group = bool(env.get("group"))   # a process group exists (always for N > 1; for N = 1 with --scaling strong: the same calls over RCCL)
    last_rank = world - 1 if last_rank is None else last_rank
    from zultra_amd import sharded
    import ctypes as C
    n = len(shard)
    host = np.concatenate([lead, shard]) if lead is not None else shard
    nlead = len(lead) if lead is not None else 0
    nblocks = (n + bs - 1) // bs
    blocks = []
    for b in range(nblocks):
        prev = HIST if (b > 0 or nlead) else 0
        blocks.append((nlead + b * bs - prev, prev, min(bs, n - b * bs)))
    block_lens = np.array([b[2] for b in blocks], dtype=np.uint32)

    d_data = torch.from_numpy(host if n else np.zeros(1, dtype=np.uint8)).to(device)   # input resident in HBM before the timed region
    torch.cuda.synchronize()
    ctx = L.context(bs, nblocks, device=env["local_rank"]) if nblocks else None   # (strong scaling: more ranks than max-blocks leaves empty shards)
    D = dist if group else OneRank
    L.L.zultra_adler32_append.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_size_t]
    L.L.zultra_adler32_append.restype = C.c_uint32
    L.L.zultra_hip_block_adler32.argtypes = [C.c_void_p, C.c_void_p]
    timings = []

    def shard_checksum():
        """This shard's checksum contribution from the per-max-block values the device computed next to the compression, as
        what a zero initial state becomes: gzip -> linear CRC-32 state, zlib -> (A, B) sums."""
        if flags == 2:
            # running value v of zultra_crc32_append is the finalized CRC; the raw register is ~v: start it at 0 -> v = ~0
            v = L.crc32_append_many(0xFFFFFFFF, ctx.block_crc32(), block_lens)
            return (~v) & 0xFFFFFFFF
        if flags == 1:
            ab = np.zeros(2 * nblocks, dtype=np.uint32)
            L.L.zultra_hip_block_adler32(ctx.h, ab.ctypes.data)
            a = 0
            for b in range(nblocks):
                a = L.L.zultra_adler32_append(a, int(ab[2 * b]), int(ab[2 * b + 1]), int(block_lens[b]))
            return a
        return 0

    colls = []

    def step():
        if ctx is not None:
            if world == 1 and not group:   # one rank: the stream starts at phase 0 here — the stitch goes out with the batch (zultra_hip_stitch_with_batch)
                ctx.stitch_with_batch(nblocks - 1, phase=0)
            ctx.compress_blocks(d_data.data_ptr(), blocks, data_on_device=True, data_size=d_data.numel())
            t = ctx.timing()
            # (one rank: the fold of the per-block checksums runs while the stitched bytes come back — sharded.assemble calls it behind the start of the copy)
            extra = (lambda: np.array([shard_checksum(), n], dtype=np.int64)) if (world == 1 and not group) else np.array([shard_checksum(), n], dtype=np.int64)
        else:
            t = None
            extra = np.array([0, 0], dtype=np.int64)   # (no bytes: a zero state stays zero, for both checksums)
        body, info = sharded.assemble(L, ctx, bs, D, torch, device, nblocks - 1 if (rank == last_rank and nblocks) else -1, extra=extra, force_collectives=group)
        if t is not None:
            t["stitch_ms"] = ctx.timing()["stitch_ms"]
            timings.append(t)
        colls.append((info.get("collective_ms", 0.0), info.get("sent_bytes", 0), info.get("received_bytes", 0)))
        return body, info

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    timings.clear()
    colls.clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        body, info = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0 and ctx is None:
        raise RuntimeError("bench.py: rank 0 has no max-block (fewer max-blocks than ranks)")
    res = {"n": n, "nblocks": nblocks, "dt": dt, "stats": ctx.stats() if ctx is not None else None, "ctx": ctx, "d_data": d_data, "blocks": blocks}
    # what every rank spent where: device pipeline (first launch to last completion of a batch), the exchange steps of the assembly
    # (head = first launch to the end of the first run's matchfinder, tail = end of the last run's matchfinder to the last completion + the stitch: neither
    # shrinks with the shard; what lies between does — DESIGN.md 5's model of N ranks on one stream, checkable from this line)
    tm = (lambda k: float(np.mean([t[k] for t in timings])) if timings else 0.0)
    mine = [tm("total_ms"), float(np.mean([c[0] for c in colls])), float(colls[-1][1]), float(colls[-1][2]), float(n), tm("head_ms"), tm("tail_ms") + tm("stitch_ms")]
    per_rank = [mine]
    if group and world > 1:
        tt = torch.tensor(mine, dtype=torch.float64, device=device)
        allr = [torch.empty_like(tt) for _ in range(world)]
        dist.all_gather(allr, tt)
        per_rank = [[float(x) for x in a.cpu().tolist()] for a in allr]
    res["per_rank"] = {"device_pipeline_ms": [round(p[0], 3) for p in per_rank], "collective_ms": [round(p[1], 3) for p in per_rank],
                       "sent_bytes": [int(p[2]) for p in per_rank], "received_bytes_rank0": int(per_rank[0][3]), "input_bytes": [int(p[4]) for p in per_rank],
                       "head_ms": [round(p[5], 3) for p in per_rank], "tail_ms": [round(p[6], 3) for p in per_rank],
                       "throughput_part_ms": [round(max(0.0, p[0] - p[5] - p[6]), 3) for p in per_rank],   # (tail_ms includes the stitch, which follows device_pipeline_ms: a lower bound)
                       "model": "a rank's step = head + throughput part + tail; only the throughput part shrinks with the rank's share of ONE stream (DESIGN.md 5); unmeasured beyond the ranks of this run"}
    if rank == 0:
        # fold the ranks' checksum contributions in stream order
        total_in, chk = 0, (0 if flags == 2 else 1)
        for ex in info["extras"]:
            c, ln = int(ex[0]), int(ex[1])
            if flags == 2:
                chk = L.crc32_append(chk, c, ln)
            elif flags == 1:
                chk = L.L.zultra_adler32_append(chk, c & 0xffff, c >> 16, ln)
            total_in += ln
        avg = {k: float(np.mean([t[k] for t in timings])) for k in timings[0]}
        kernels = {"zh_mf_group": avg["group_ms"], "zh_mf_frontier": avg["frontier_ms"],
                   "zh_barriers+zh_tokenize_spans+zh_split": avg["tokenize_split_ms"], "zh_plan_subblocks+zh_sb_init": avg["init_ms"],
                   "zh_parse_lanes+zh_parse_chain": avg["parse_ms"], "zh_sb_build": avg["build_ms"], "zh_post_tasks": avg["post_ms"],
                   "zh_emit_tasks": avg["emit_ms"], "zh_stitch": avg["stitch_ms"]}
        # the library runs a batch as staggered runs of max-blocks on separate streams (ZULTRA_HIP_STREAMS): every kernel is launched
        # once per run (the parse / code-rebuild pair once per pass and run) over 1/runs of the batch
        runs = max(1, int(res["stats"]["runs"]))
        launches = {k: runs for k in kernels}
        launches["zh_parse_lanes+zh_parse_chain"] = launches["zh_sb_build"] = 4 * runs
        launches["zh_stitch"] = 1
        dom = max(kernels, key=lambda k: kernels[k])
        out_bytes = len(body) / world
        # 1 B read + r B written per input byte; one launch covers 1/runs of the batch
        alg_bytes = (n + out_bytes) / (runs if dom != "zh_stitch" else 1)
        launch_ms = kernels[dom] / launches[dom]
        achieved = alg_bytes / (launch_ms * 1e-3) / 1e9
        traffic, traffic_src, traffic_stale = committed_traffic(dom, env.get("config"))
        st = res["stats"]
        res.update({
            "body": body, "checksum": chk, "total_in": total_in, "ms_per_step": dt / steps * 1e3,
            "MBps": total_in / (dt / steps) / 1e6,
            # HIP-event intervals on the library's streams, SUMMED over the batch's staggered runs: the runs overlap in wall time, so these add up to more
            # than the step (each is the sum of a kernel group's launch durations, what the roofline's launch_ms divides)
            "kernel_stream_interval_sums_ms": {k: round(v, 3) for k, v in kernels.items()},
            "device_pipeline_ms": round(avg["total_ms"], 3), "d2h_ms": round(avg["d2h_ms"], 3),
            "sub_blocks_per_block": round(st["subblocks"] / max(1, st["blocks"]), 3),
            "parse_huge_share_of_positions": round(st["huge_positions"] / max(1, st["positions"]), 4),
            "parse_tasks": st["tasks"], "parse_huge_tasks": st["huge_tasks"],
            "chain_cut": {"tasks": st["cut_tasks"], "segments": st["cut_segments"], "redone_over_4_passes": st["cut_redone"],
                          "tasks_handed_to_the_chain_kernel": st["cut_demoted"]},
            # sub-blocks whose code lengths reached a fixed point of the four-pass loop are not parsed again: the share of the
            # 4 x positions of parse work that was not run
            "parse_settled_share": round(st["settled_kib"] * 1024 / max(1, 4 * st["positions"]), 4),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 4), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src, "traffic_stale": traffic_stale,
                         "algorithmic_bytes_per_launch": int(alg_bytes), "launch_ms": round(launch_ms, 3),
                         "launches_per_step": launches[dom]},
        })
    return res


def summarize_leg(r):
    return {k: r[k] for k in ("ms_per_step", "kernel_stream_interval_sums_ms", "device_pipeline_ms", "d2h_ms", "sub_blocks_per_block",
                              "parse_huge_share_of_positions", "parse_settled_share", "parse_tasks", "parse_huge_tasks", "chain_cut", "per_rank")}


# ---------------------------------------------------------------------------------------------------------------------
def prepare_stream_config(args, rank, world):
    """CPU side of configurations 2-4, before this process touches the GPU: this rank's shard and the CPU reference timing."""
    cfg = args.config
    if cfg == 2:
        flags, bs, size = 2, args.block or 65536, args.size or 100_000_000
        corp, data_note = text_corpus(world, size)
        metric = "input MB/s, gzip 64 KiB max-blocks, enwik8-sized text, bit-exact vs CPU zultra"
        if getattr(args, "synthetic_leg", False):   # (a leg of the default run: round 1's headline corpus in a process of its own)
            corp, data_note = SyntheticText(), "round 1's seeded Zipf word stream (tests/corpus.py: text_like_fast): never splits, never meets the chain parse"
            metric = "input MB/s, gzip 64 KiB max-blocks, synthetic text, bit-exact vs CPU zultra"
    elif cfg == 3:
        flags, bs, size = 1, args.block or 32768, args.size or 51_220_480
        corp, data_note = binary_corpus(world, size)
        metric = "input MB/s + size vs zlib-9, zlib framing 32 KiB max-blocks, silesia/mozilla-sized binary, bit-exact vs CPU zultra"
    else:
        flags, bs, size = 2, args.block or 65536, args.size or (1 << 30)
        corp, data_note = MixedConfig4(), "synthetic 8 GiB mixed-entropy corpus (tests/gen/zgen.c: splitmix64 seed 0x5EED, 1 MiB segments over the self-test grid, 1/16 noise, 1/16 constant): 1 GiB per GPU"
        metric = "input MB/s, gzip 64 KiB max-blocks, synthetic mixed-entropy corpus 1 GiB per GPU, bit-exact vs CPU zultra"
    last_rank = world - 1
    if args.scaling == "strong" or getattr(args, "strong_leg", False):
        # strong scaling: the configuration's ONE stream of `size` bytes, its max-blocks cut contiguously over the ranks (zultra_amd.sharded.shard_range)
        from zultra_amd.sharded import shard_range
        nb_total = (size + bs - 1) // bs
        lo, hi = shard_range(nb_total, rank, world)
        last_rank = max(r for r in range(world) if shard_range(nb_total, r, world)[1] > shard_range(nb_total, r, world)[0])
        if hi > lo:
            lead, shard = corp.range(lo * bs, min(size, hi * bs) - lo * bs)
        else:
            lead, shard = None, np.zeros(0, dtype=np.uint8)
    else:
        lead, shard = corp.shard(rank, size)
    cb = ref_out = check = None
    sample = shard[: min(args.cpu_sample, len(shard))]
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.profile_run:
        # what is checked against the reference, byte for byte: the WHOLE shard for configurations 2 and 3 (the stream the timed steps produce), the first
        # 64 MiB of configuration 4's GiB (its reference run would take three minutes of one core) — in a child process, next to everything else; the
        # reference's SPEED is taken on the first --cpu-sample bytes, best of three, with nothing else running
        cb, ref_out = cpu_baseline_stream(sample, flags, bs, corp.name)
        nchk = len(shard) if (args.cpu_check == 0 and cfg != 4) else min(len(shard), args.cpu_check or (64 << 20))
        check = start_reference_check(shard[:nchk], flags, bs)
    return dict(cfg=cfg, flags=flags, bs=bs, size=size, corp=corp, data_note=data_note, metric=metric, lead=lead, shard=shard, sample=sample, cb=cb,
                ref_out=ref_out, last_rank=last_rank, check=check)


def run_projection(args, env, prep):
    """Strong-scaling PROJECTION from one GPU — NOT a scaling measurement: the configuration's one stream is cut over N = 1, 2, 4, 8 ranks by
    sharded.shard_range, and every one of the N shards is put through this GPU's whole step — kernels, phase table, stitch, the collectives of N > 1 over a 1-rank RCCL
    group, read-back — one after the other. A rank of an N-GPU job would do exactly that with its shard, next to the others; what the projection leaves out is the
    transfer of the other ranks' bytes to rank 0 (~2.5 MB per rank over xGMI at N = 8) and any interference between the ranks' collectives. Per N: every shard's ms per
    step, and the stream's size over the SLOWEST shard's step."""
    from zultra_amd.sharded import shard_range
    flags, bs, size, corp = (prep[k] for k in ("flags", "bs", "size", "corp"))
    nb_total = (size + bs - 1) // bs
    proj = {"note": "projection from ONE GPU, not a scaling measurement: each of the N shards of the one %d-byte stream timed on this GPU in turn (1-rank RCCL group for the collectives); "
                    "value_if_ranks_ran_side_by_side = stream bytes / slowest shard's step" % size, "by_ranks": {}}
    for n in (1, 2, 4, 8):
        ms, heads, tails = [], [], []
        for r in range(n):
            lo, hi = shard_range(nb_total, r, n)
            lead, shard = corp.range(lo * bs, min(size, hi * bs) - lo * bs)
            leg = run_stream_leg(env, lead, shard, flags, bs, args.steps, args.warmup, last_rank=0 if r == n - 1 else 1)   # (only the stream's last shard carries BFINAL)
            c = leg.pop("ctx")
            if c is not None:
                c.close()
            leg.pop("d_data")
            ms.append(round(leg["ms_per_step"], 3))
            heads.append(leg["per_rank"]["head_ms"][0])
            tails.append(leg["per_rank"]["tail_ms"][0])
        proj["by_ranks"][str(n)] = {"shard_bytes": int(min(size, shard_range(nb_total, 0, n)[1] * bs)), "ms_per_step_by_shard": ms, "slowest_shard_ms": max(ms), "head_ms": heads, "tail_ms": tails,
                                    "value_if_ranks_ran_side_by_side_MBps": round(size / (max(ms) * 1e-3) / 1e6, 1)}
    return {"strong_scaling_projection": proj}, False


def run_stream_config(args, env, prep):
    if getattr(args, "projection", False):
        return run_projection(args, env, prep)
    L, rank, world = env["L"], env["rank"], env["world"]
    cfg, flags, bs, size, corp, data_note, metric = (prep[k] for k in ("cfg", "flags", "bs", "size", "corp", "data_note", "metric"))
    lead, shard, sample, cb, ref_out = (prep[k] for k in ("lead", "shard", "sample", "cb", "ref_out"))
    strong = args.scaling == "strong"
    head = run_stream_leg(env, lead, shard, flags, bs, args.steps, args.warmup, last_rank=prep["last_rank"])
    line = None
    failed = False
    if rank == 0:
        line = {
            "metric": metric, "value": round(head["MBps"], 3), "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(head["ms_per_step"], 3), "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "u8/int32",
            "data": "real" if corp.name.startswith(("enwik8", "real")) else "synthetic",
            "config": {"workload": ("config %d: %s, %d B in all (one stream cut over the GPUs), flags=%d, nMaxBlockSize=%d, %d max-blocks on rank 0" if strong else
                                    "config %d: %s, %d B per GPU, flags=%d, nMaxBlockSize=%d, %d max-blocks per GPU") % (cfg, data_note, size, flags, bs, head["nblocks"]),
                       "parallelism": "blocks sharded over %d GPU(s), phase-table all-gather + exact-length transfers to rank 0%s" % (
                           world, " (1 rank: the same collectives over RCCL, nothing to transfer)" if (world == 1 and env.get("group")) else "")},
            "rccl_ranks_seen": env["ranks_seen"],
        }
        line.update(summarize_leg(head))
        line["compressed_bytes_total"] = len(head["body"])
        line["roofline"] = head["roofline"]
        sq = committed_sq(cfg, head["ms_per_step"])
        if sq:
            line["issue"] = sq
        if args.profile_run:
            L.traffic_probe(256 << 20)
            return line, False
        bw = L.copy_bandwidth(1 << 30, 5)
        if bw > 0:
            line["roofline"]["peak_measured"] = round(bw, 1)
            line["roofline"]["frac_of_measured"] = round(head["roofline"]["achieved"] / bw, 6)
        framed = frame(L, flags, head["body"].tobytes(), head["checksum"], head["total_in"])
        ok = inflate_check(flags, framed, shard, head["total_in"])
        line["inflate_roundtrip_ok"] = ok
        failed |= not ok
    ctx, d_data = head.pop("ctx"), head.pop("d_data")
    if ctx is not None:
        ctx.close()
    del d_data
    if world > 1 and not strong and not args.profile_run:
        # beside the weak-scaling value: the configuration's ONE stream (its own size) cut over the same ranks — what BASELINE.json's metric names
        args.strong_leg = True
        sp = prepare_stream_config(args, rank, world)
        args.strong_leg = False
        sl = run_stream_leg(env, sp["lead"], sp["shard"], flags, bs, args.steps, args.warmup, last_rank=sp["last_rank"])
        c2 = sl.pop("ctx")
        if c2 is not None:
            c2.close()
        sl.pop("d_data")
        if rank == 0:
            sframed = frame(L, flags, sl["body"].tobytes(), sl["checksum"], sl["total_in"])
            sok = inflate_check(flags, sframed, sp["shard"], sl["total_in"])
            failed |= not sok
            line["strong_scaling"] = {"value": round(sl["MBps"], 3), "unit": "MB/s", "ms_per_step": round(sl["ms_per_step"], 3), "total_input_bytes": sl["total_in"],
                                      "inflate_roundtrip_ok": sok, "per_rank": sl["per_rank"], "compressed_bytes_total": len(sl["body"])}

    if rank == 0 and world == 1:
        # whole-input ratio against zlib-9 (README.md:16-46 quotes sizes against zlib/zopfli)
        zs = shard if len(shard) <= (128 << 20) else shard[: 64 << 20]   # (zlib-9 runs at ~20 MB/s: a GiB shard is priced on its first 64 MiB)
        zbytes = zs.tobytes()
        t0 = time.perf_counter()
        z9 = len(zlib.compress(zbytes, 9))
        z9_s = time.perf_counter() - t0   # (zlib alone: not the copy to bytes, not our own compress of the sample below)
        ours = len(framed) if zs is shard else len(L.memory_compress(zs, flags, bs))
        line["size_vs_zlib9"] = round(ours / (z9 + (12 if flags == 2 else 0)), 5)
        if zs is not shard:
            line["size_vs_zlib9_sample_bytes"] = len(zs)
        line["zlib9_MBps_1core"] = round(len(zs) / z9_s / 1e6, 1)
        # the drop-in entry on a host buffer: H2D, kernels, stitch, D2H, frame (PCIe-inclusive; never `value`)
        # (into a buffer the caller owns and has touched, as lzbench and tool/zultra.c -cbench do: the call itself, not Python's allocation
        # and copy of the result)
        obuf = np.zeros(L.memory_bound(len(shard), flags, bs), dtype=np.uint8)
        best, nout = None, None
        for _ in range(2 if args.leg else 3):
            t0 = time.perf_counter()
            nout = L.memory_compress_into(shard, flags, bs, obuf)
            dtm = time.perf_counter() - t0
            best = dtm if best is None else min(best, dtm)
        out = obuf[:nout].tobytes() if nout is not None else None
        del obuf
        line["end_to_end_MBps"] = round(len(shard) / best / 1e6, 1)
        same = out == framed
        line["memory_compress_equals_sharded_pipeline"] = bool(same)
        failed |= not same
        # a named corpus is on the box: the size the reference's README quotes for it (README.md:16,25; lzbench, framing and block size
        # not stated: compared as raw deflate at the default block size, reported, not asserted)
        readme = {"enwik8": 35029585, "real_mozilla": 18280189}.get(corp.name)
        if readme and len(shard) == len(corp.base):
            raw = L.memory_compress(shard, 0, 0)
            line["readme_size_check"] = {"readme_bytes": readme, "raw_deflate_default_block_bytes": len(raw), "equal": bool(len(raw) == readme)}
        if cb is not None:
            line["cpu_baseline"] = cb
            chk = prep["check"]
            # the stream the timed steps produced when the check covers the whole shard; else the drop-in call on the checked prefix
            gpu_stream = framed if chk["bytes"] == len(shard) else L.memory_compress(shard[: chk["bytes"]], flags, bs)
            fields, same = finish_reference_check(chk, gpu_stream)
            line.update(fields)
            failed |= not same
        if cfg == 2 and not args.leg and not strong:
            # Not `value`: what a caller gets who keeps three jobs of this size in flight — a context and a host thread each, the same shard, the
            # same job (kernels, stitch, read-back into a pinned buffer of its own). One job at a time leaves the chip nearly idle for its last
            # ~3.5 ms (final code build, literalisation, emission, stitch plan on the host, read-back: DESIGN.md 4); other jobs fill that.
            import threading
            torch, device = env["torch"], env["device"]
            K, reps = 3, max(3, min(8, args.steps))
            blocks_ = head["blocks"]
            d_data_ = torch.from_numpy(shard if lead is None else np.concatenate([lead, shard])).to(device)   # (world 1: no lead; as run_stream_leg lays it out)
            ctxs = [L.context(bs, head["nblocks"], device=env["local_rank"]) for _ in range(K)]
            pins = [torch.empty(len(head["body"]) + (1 << 20), dtype=torch.uint8, pin_memory=True).numpy() for _ in range(K)]
            outs, errs = [None] * K, []

            def job(i, nrep):
                try:
                    for _ in range(nrep):
                        ctxs[i].compress_blocks(d_data_.data_ptr(), blocks_, data_on_device=True, data_size=d_data_.numel())
                        end_bit, _ = ctxs[i].stitch_device(head["nblocks"] - 1, phase=0)
                        outs[i] = ctxs[i].stream_read((end_bit + 7) // 8, out=pins[i])
                except Exception as e:   # noqa: BLE001 (reported below)
                    errs.append(repr(e))

            for i in range(K):
                job(i, 1)   # warm every context
            torch.cuda.synchronize()
            ths = [threading.Thread(target=job, args=(i, reps)) for i in range(K)]
            t0 = time.perf_counter()
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            torch.cuda.synchronize()
            dt3 = time.perf_counter() - t0
            same3 = not errs and all(o is not None and o.tobytes() == head["body"].tobytes() for o in outs)
            line["three_jobs_in_flight"] = {"MBps": round(K * reps * head["n"] / dt3 / 1e6, 1), "ms_per_job": round(dt3 / (K * reps) * 1e3, 3), "jobs": K * reps,
                                            "same_bytes_as_value_run": bool(same3), "note": "not `value`: three contexts, one host thread each, on this one GPU"}
            failed |= not same3
            for c_ in ctxs:
                c_.close()
            del d_data_
    return line, failed


def prepare_config1(args, rank, world):
    import corpus
    data = corpus.bootstrap_js()
    p = find_file("ZULTRA_BOOTSTRAP", ["bootstrap.min.js"], 48944)   # the README's file (README.md:43: 12 599 B): reported beside the stand-in's known answer
    if p:
        args.readme_bootstrap = np.fromfile(p, dtype=np.uint8)
    cb, ref_out = cpu_baseline_stream(data, 0, 0, "bootstrap.min.js v3.4.1")
    return dict(data=data, cb=cb, ref_out=ref_out)


def run_config1(args, env, prep):
    """Plumbing: one small file, raw deflate, default block size; known answer from the compiled reference (tests/golden)."""
    import golden_util
    L = env["L"]
    data, cb, ref_out = prep["data"], prep["cb"], prep["ref_out"]
    case_raw, case_gz = golden_util.stream_case("bootstrap_raw_default"), golden_util.stream_case("bootstrap_gzip_default")
    out = L.memory_compress(data, 0, 0)
    for _ in range(args.warmup):
        L.memory_compress(data, 0, 0)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = L.memory_compress(data, 0, 0)
    dt = (time.perf_counter() - t0) / args.steps
    gz = L.memory_compress(data, 2, 0)
    ok = out == case_raw["out"] and gz == case_gz["out"] and out == ref_out and zlib.decompress(out, -15) == data.tobytes()
    ctx = L.context(0, 1)
    ctx.compress_blocks(data, [(0, 0, len(data))])
    ctx.compress_blocks(data, [(0, 0, len(data))])
    t = ctx.timing()
    kern = {k: t[k] for k in ("group_ms", "frontier_ms", "tokenize_split_ms", "init_ms", "parse_ms", "build_ms", "post_ms", "emit_ms")}
    dom = max(kern, key=lambda k: kern[k])
    nl = 4 if dom in ("parse_ms", "build_ms") else 1
    achieved = (len(data) + len(out)) / (kern[dom] / nl * 1e-3) / 1e9
    ctx.close()
    line = {"metric": "input MB/s, bootstrap.min.js raw deflate, one max-block, known answer", "value": round(len(data) / dt / 1e6, 3), "unit": "MB/s",
            "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/int32", "data": "real",
            "config": {"workload": "config 1: the image's bootstrap.min.js v3.4.1 (39 680 B; the README's 48 944-byte file is absent), raw deflate, default block size, through zultra_memory_compress (host buffer, PCIe-inclusive: a single latency-bound call)"},
            "compressed_bytes": len(out), "expected_bytes_raw": case_raw["out_len"], "gzip_bytes": len(gz), "expected_bytes_gzip": case_gz["out_len"],
            "known_answer_ok": bool(ok), "device_ms": {k: round(v, 3) for k, v in kern.items()},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 5), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 8),
                         "traffic": None, "launch_ms": round(kern[dom] / nl, 3), "launches_per_step": nl},
            "cpu_baseline": cb}
    if getattr(args, "readme_bootstrap", None) is not None:
        rb = L.memory_compress(args.readme_bootstrap, 0, 0)
        line["readme_size_check"] = {"readme_bytes": 12599, "raw_deflate_default_block_bytes": len(rb), "equal": bool(len(rb) == 12599)}
    return line, not ok


def prepare_config5(args, rank, world):
    import corpus
    size, nfiles = 4096, args.files
    host = corpus.json_files(rank * nfiles, nfiles, size)
    cb = digest = None
    ncpu, per = min(nfiles, args.batch, args.cpu_check_files), 0   # files compared with the reference: inside the first device batch
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.profile_run:
        cb, digest, per = cpu_baseline_files(host[: ncpu * size], size, 2, args.cpu_files)   # forks: must precede any GPU call of this process
    return dict(host=host, cb=cb, digest=digest, ncpu=ncpu, per=per)


def run_config5(args, env, prep):
    """1 000 000 x 4 KiB JSON-like inputs per GPU, each its own gzip stream: files mode, captured hipGraphs replayed per batch (one, or two per run of inputs)."""
    L, torch, dist, device, rank, world = env["L"], env["torch"], env["dist"], env["device"], env["rank"], env["world"]
    size, nfiles, batch = 4096, args.files, args.batch
    host, cb, digest, ncpu = prep["host"], prep["cb"], prep["digest"], prep["ncpu"]
    d = torch.from_numpy(host).to(device)
    torch.cuda.synchronize()
    # Batches are independent jobs: `--contexts` device contexts (default one) take alternate batches, one host thread each (ctypes releases the GIL), so that
    # one batch's stitch, read-back and descriptor handling — and the poorly filled head and tail of its kernel sequence — run next to the other's kernels. (Measured, 1 000 000 inputs: 1.13-1.14 M files/s with one context, 1.135-1.139 with two, 1.159 with three — the chip
    # is full either way; the default stays one.)
    nctx = max(1, min(args.contexts, (nfiles + batch - 1) // batch)) if not args.profile_run else 1
    ctxs = [L.files_context(size, batch, device=env["local_rank"]) for _ in range(nctx)]
    ctx = ctxs[0]
    nb = (nfiles + batch - 1) // batch
    sizes_full = np.full(batch, size, dtype=np.uint32)
    # the batches' output comes back into one pinned buffer per context (an input deflates to less than its size + 64: reference bound libzultra.c:601-619)
    pinneds = [torch.empty(batch * (size + 64), dtype=torch.uint8, pin_memory=True).numpy() for _ in range(nctx)]
    timings = []
    keep = {}

    def lane(t, collect, totals):
        c, pinned = ctxs[t], pinneds[t]
        out_bytes = 0
        for b in range(t, nb, nctx):
            k = min(batch, nfiles - b * batch)
            offs = (np.arange(k, dtype=np.uint64) + np.uint64(b * batch)) * np.uint64(size)   # absolute: one base pointer -> one captured graph
            fo = c.compress_files(d.data_ptr(), offs, sizes_full[:k], data_on_device=True, data_size=d.numel())
            stream = c.stream_read(int(fo[-1]), out=pinned if int(fo[-1]) <= pinned.size else None)
            crcs = c.block_crc32()
            out_bytes += int(fo[-1]) + 18 * k
            timings.append(c.timing())
            if collect and b == 0:
                keep.update(fo=fo.copy(), stream=stream.copy(), crcs=crcs.copy())
        totals[t] = out_bytes

    def step(collect=False):
        totals = [0] * nctx
        if nctx == 1:
            lane(0, collect, totals)
        else:
            import threading
            ths = [threading.Thread(target=lane, args=(t, collect, totals)) for t in range(nctx)]
            for th in ths:
                th.start()
            for th in ths:
                th.join()
        return sum(totals)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(1, args.warmup)):   # the first pass captures the graphs
        step()
    timings.clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out_bytes = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    step(collect=True)
    line, failed = None, False
    if rank == 0:
        # checks on the first batch: every file framed as gzip inflates to its input; the first files equal the CPU reference
        fo, stream, crcs = keep["fo"], keep["stream"], keep["crcs"]
        k0 = min(batch, nfiles)
        hdr = bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 2, 255])
        per = prep["per"] or ncpu or 1
        hs = [hashlib.sha256() for _ in range((ncpu + per - 1) // per)]   # one digest per shard of the reference's worker pool (cpu_baseline_files)
        ok = True
        for k in range(min(k0, max(ncpu, 2048))):
            raw = stream[int(fo[k]):int(fo[k + 1])].tobytes()
            gz = hdr + raw + int(L.crc32_append(0, crcs[k], size)).to_bytes(4, "little") + int(size).to_bytes(4, "little")
            if k < ncpu:
                hs[k // per].update(gz)
            if k < 2048 and zlib.decompress(gz, 31) != host[k * size:(k + 1) * size].tobytes():
                ok = False
        failed |= not ok
        avg_graph = float(np.mean([t["encode_ms"] for t in timings]))
        avg_stitch = float(np.mean([t["stitch_ms"] for t in timings]))
        in_b, out_b = min(batch, nfiles) * size, out_bytes / nb
        achieved = (in_b + out_b) / (avg_graph * 1e-3) / 1e9
        line = {"metric": "files/s, 4 KiB JSON-like inputs, one gzip stream each, bit-exact vs CPU zultra", "value": round(world * nfiles / (dt / args.steps), 1),
                "unit": "files/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/int32", "data": "synthetic",
                "config": {"workload": "config 5: %d x %d B JSON-like inputs per GPU (tests/gen/zgen.c), each its own gzip stream, batches of %d inputs, %s + one stitch launch per batch; %s" % (
                    nfiles, size, batch, "one hipGraph replay" if ctx.stats()["runs"] <= 1 else "%d staggered runs of inputs, each two captured hipGraphs replayed" % ctx.stats()["runs"],
                    "one device context" if nctx == 1 else "%d device contexts take alternate batches, one host thread each" % nctx), "contexts": nctx},
                "rccl_ranks_seen": env["ranks_seen"], "input_MBps": round(world * nfiles * size / (dt / args.steps) / 1e6, 2),
                "ratio": round(out_bytes / (nfiles * size), 4), "graph_ms_per_batch": round(avg_graph, 3), "stitch_ms_per_batch": round(avg_stitch, 3),
                "gzip_roundtrip_ok_first_files": bool(ok),
                "roofline": {"bound": "hbm", "kernel": "hipGraph of stages 1-3 (one replay per batch)", "achieved": round(achieved, 4), "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": committed_traffic("graph", 5)[0], "traffic_source": committed_traffic("graph", 5)[1], "traffic_stale": committed_traffic("graph", 5)[2],
                             "algorithmic_bytes_per_launch": int(in_b + out_b), "launch_ms": round(avg_graph, 3), "launches_per_step": nb}}
        if args.profile_run:
            L.traffic_probe(256 << 20)
        elif cb is not None:
            line["cpu_baseline"] = cb
            same = [h.hexdigest() for h in hs] == digest
            line["bit_exact_vs_reference_full"] = bool(same)   # every compared file's gzip stream, byte for byte
            line["bit_exact_checked_files"] = ncpu
            line["bit_exact_checker"] = cb["kind"]
            failed |= not same
    for c in ctxs:
        c.close()
    return line, failed


# ---------------------------------------------------------------------------------------------------------------------
def run_other_configs(args):
    """The default run (configuration 2 on one GPU) also measures configurations 3, 4 and 5, bounded so that the whole command stays
    within a few minutes: each as a child process of its own (started before this process touches the GPU), its JSON line condensed."""
    # (each leg compares its stream with the CPU reference's in full — configuration 4: the first 64 MiB of its GiB, configuration 5: its first 65 536 files —
    # and times the reference on a smaller sample: ~25 s each)
    legs = {3: ["--cpu-sample", str(8 << 20)], 4: ["--cpu-sample", str(4 << 20)], 5: ["--files", "1000000", "--cpu-files", "2048"]}
    if not args.no_synthetic:
        # round 1's headline corpus, for continuity — in a process of its own since round 5: behind other contexts in ONE process the runtime puts two runs of a
        # batch on one hardware queue (28.5 ms per step there, 23.5 in a fresh process)
        legs["synthetic_text"] = ["--synthetic-leg", "--cpu-sample", str(8 << 20), "--cpu-check", str(32 << 20)]
    legs["strong_scaling_projection"] = ["--scaling", "strong", "--projection", "--no-cpu-baseline"]
    out = {}
    for cfg, extra in legs.items():
        cmd = [sys.executable, os.path.abspath(__file__), "--config", "2" if cfg in ("synthetic_text", "strong_scaling_projection") else str(cfg), "--gpus", "1",
               "--steps", "10" if cfg == "synthetic_text" else "5" if cfg == "strong_scaling_projection" else "2",
               "--warmup", "3" if cfg in ("synthetic_text", "strong_scaling_projection") else "1", "--leg"] + extra
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            lines = [x for x in r.stdout.decode().strip().splitlines() if x.startswith("{")]
            d = json.loads(lines[-1]) if lines else None
            rc = r.returncode
        except (subprocess.TimeoutExpired, ValueError) as e:
            d, rc = None, "%s" % type(e).__name__
        if d is None:
            out[str(cfg)] = {"error": "no line (rc %s)" % rc, "wall_s": round(time.perf_counter() - t0, 1)}
            continue
        if cfg == "strong_scaling_projection":
            out[cfg] = dict(d.get("strong_scaling_projection", {"error": "no projection in the leg's line"}), rc=rc, wall_s=round(time.perf_counter() - t0, 1))
            continue
        keep = ("metric", "value", "unit", "ms_per_step", "kernel_stream_interval_sums_ms", "issue", "graph_ms_per_batch", "input_MBps", "ratio", "size_vs_zlib9", "size_vs_zlib9_sample_bytes",
                "inflate_roundtrip_ok", "gzip_roundtrip_ok_first_files", "bit_exact_vs_reference_full", "bit_exact_checked_input_bytes", "bit_exact_checked_files", "bit_exact_checker",
                "memory_compress_equals_sharded_pipeline", "end_to_end_MBps",
                "compressed_bytes_total", "sub_blocks_per_block", "parse_huge_share_of_positions", "parse_settled_share", "chain_cut", "readme_size_check")
        o = {k: d[k] for k in keep if k in d}
        o["workload"] = d["config"]["workload"]
        o["roofline"] = d["roofline"]
        if "cpu_baseline" in d:
            o["cpu_baseline"] = {k: d["cpu_baseline"][k] for k in ("value", "unit", "cores", "kind", "all_cores_value") if k in d["cpu_baseline"]}
        o["rc"] = rc
        o["wall_s"] = round(time.perf_counter() - t0, 1)
        out[str(cfg)] = o
    return out


# ---------------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2, choices=[1, 2, 3, 4, 5], help="BASELINE.json configuration (default 2: the one the metric is quoted on)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): every rank takes a shard of the configuration's size of ONE N-times-larger stream; strong: the configuration's one stream is cut "
                         "over the N ranks (N = 1: the same job, driven through a 1-rank RCCL group so that the collectives of N > 1 execute). At N > 1 the weak "
                         "line carries the strong measurement beside it (strong_scaling)")
    ap.add_argument("--size", type=int, default=0, help="bytes per GPU (default: the configuration's own size)")
    ap.add_argument("--block", type=int, default=0, help="nMaxBlockSize (default: the configuration's own)")
    ap.add_argument("--files", type=int, default=1_000_000, help="config 5: inputs per GPU")
    ap.add_argument("--batch", type=int, default=1 << 16, help="config 5: inputs per device batch")
    ap.add_argument("--contexts", type=int, default=1, help="config 5: device contexts that take alternate batches, one host thread each (1: one batch at a time)")
    ap.add_argument("--cpu-sample", type=int, default=32 << 20, help="bytes of the shard the CPU reference is TIMED on (one core, best of three)")
    ap.add_argument("--cpu-check", type=int, default=0, help="bytes of the shard whose stream is COMPARED with the CPU reference's (0: all of it; configuration 4: 64 MiB)")
    ap.add_argument("--cpu-files", type=int, default=4096, help="config 5: files the CPU reference is timed on (one core)")
    ap.add_argument("--cpu-check-files", type=int, default=65536, help="config 5: files compared with the CPU reference (all host cores)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-synthetic", action="store_true", help="default run: skip the synthetic-text leg")
    ap.add_argument("--synthetic-leg", action="store_true", help="(internal) configuration 2's settings on round 1's synthetic text: a leg of the default run")
    ap.add_argument("--no-other-configs", action="store_true", help="default run (config 2, one GPU): skip the legs of configurations 3, 4 and 5")
    ap.add_argument("--leg", action="store_true", help="(internal) this process is one of those legs: bounded extras")
    ap.add_argument("--projection", action="store_true", help="(internal; --config 2 --gpus 1 --scaling strong) strong-scaling PROJECTION from one GPU: every shard of the configuration's "
                                                               "one stream cut over N = 2, 4, 8 ranks, timed on this GPU one after the other (a leg of the default run)")
    ap.add_argument("--profile-run", action="store_true",
                    help="for rocprofv3 passes: only the steps (no round-trip / ratio / CPU extras that would add dispatches), then the PMC calibration probe")
    args = ap.parse_args()

    have_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not have_launcher:
        sys.exit(spawn_ranks(args, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0")) if have_launcher else 0
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank))) if have_launcher else 0
    world = int(os.environ.get("WORLD_SIZE", "1")) if have_launcher else 1
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher set WORLD_SIZE=%d\n" % (args.gpus, world))
        sys.exit(2)
    if args.config == 1 and world > 1:
        sys.stderr.write("bench.py: configuration 1 is a single 39 680-byte input: there is nothing to shard\n")
        sys.exit(2)

    # Only the JSON line goes to this process's stdout: libraries that print there on their own (RCCL's version banner comes out of C stdio
    # at exit, i.e. AFTER the line) are sent to stderr for the whole run.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    # ---- the other configurations' legs, as child processes, before this process touches a GPU ---------------------------------------
    other = None
    if args.config == 2 and world == 1 and not have_launcher and not args.profile_run and not args.leg and not args.no_other_configs:
        other = run_other_configs(args)

    # ---- CPU side first: corpus generation and the CPU reference timing (config 5 forks worker processes) -------------------
    prep = {1: prepare_config1, 5: prepare_config5}.get(args.config, prepare_stream_config)(args, rank, world)

    import torch
    import torch.distributed as dist

    import zultra_amd

    if torch.cuda.device_count() <= local_rank:
        sys.stderr.write("bench.py: rank %d needs GPU %d but %d GPU(s) are visible\n" % (rank, local_rank, torch.cuda.device_count()))
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    ranks_seen = 1
    group = world > 1 or (args.scaling == "strong" and args.config in (2, 3, 4))
    if group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        dist.init_process_group("nccl", rank=rank, world_size=world)
        one = torch.ones(1, dtype=torch.int64, device=device)
        dist.all_reduce(one)   # every rank present and reachable over RCCL
        ranks_seen = int(one.item())
        if ranks_seen != world:
            sys.stderr.write("bench.py: %d ranks answered, %d expected\n" % (ranks_seen, world))
            sys.exit(2)

    L = zultra_amd.lib()   # raises if libzultra_amd.so is missing: there is no fallback path
    if L.device_count() < 1:
        raise RuntimeError("no HIP device")
    env = {"L": L, "torch": torch, "dist": dist, "device": device, "rank": rank, "local_rank": local_rank, "world": world, "ranks_seen": ranks_seen, "config": args.config,
           "group": group}

    line, failed = {1: run_config1, 5: run_config5}.get(args.config, run_stream_config)(args, env, prep)
    if rank == 0:
        if other is not None:
            failed |= any(o.get("rc", 1) != 0 for o in other.values())
            if "synthetic_text" in other:   # (under the name it has had since round 2)
                line["synthetic_text"] = dict(other.pop("synthetic_text"), note="a child process of its own since round 5")
                line["synthetic_text"]["MBps"] = line["synthetic_text"].get("value")
            if "strong_scaling_projection" in other:
                line["strong_scaling_projection"] = other.pop("strong_scaling_projection")
            line["other_configs"] = other
        real_stdout.write(json.dumps(line) + "\n")
        real_stdout.flush()
    if group:
        flag = torch.tensor([1 if failed else 0], dtype=torch.int64, device=device)
        dist.all_reduce(flag)
        failed = bool(flag.item())
        dist.destroy_process_group()
    if failed:
        sys.stderr.write("bench.py: a correctness check failed (see the JSON line)\n")
        sys.exit(1)


if __name__ == "__main__":
    main()
