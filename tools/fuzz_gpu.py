#!/usr/bin/env python3
"""Diagnostics / soak test on the GPU box: seeded inputs stitched together from the test corpora (text, byte runs, near copies, tables,
noise, constant and periodic stretches, JSON, the mixed stream), random sizes from 1 byte to a few MB, the three framings, 32 KiB / 64 KiB /
default max-blocks — the product library's zultra_memory_compress against the compiled reference (oracle/_ref, built here and carried over by
gpurun) byte for byte. The reference outputs come from a pool of processes started BEFORE this process touches the GPU.
usage: python tools/fuzz_gpu.py [cases] [seed] [max bytes per case] | --files [inputs] [seed] [batch] | --stream [cases] [seed] [max bytes]     exit code 1 on the first difference (the case is written to
gpurun_out/fuzz_fail_<seed>_<case>.bin)"""
import multiprocessing as mp
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import corpus  # noqa: E402


def make_case(seed, k, max_bytes):
    rs = np.random.RandomState((seed * 1000003 + k) & 0x7fffffff)
    kind = rs.randint(0, 10)
    if kind == 0:
        size = int(rs.randint(1, 300))
    elif kind < 4:
        size = int(rs.randint(300, 70000))
    elif kind < 8:
        size = int(rs.randint(70000, min(max_bytes, 600000)))
    else:
        size = int(rs.randint(600000, max_bytes)) if max_bytes > 600000 else int(rs.randint(70000, max_bytes))
    parts, n = [], 0
    while n < size:
        g = rs.randint(0, 12)
        m = int(min(size - n, rs.choice([200, 2000, 9000, 40000, 70000, 200000])))
        s = int(rs.randint(1, 1 << 20))
        if g == 0:
            p = corpus.text_like(m, s)
        elif g == 1:
            p = corpus.noise(m, s)
        elif g == 2:
            p = corpus.constant(m, int(rs.randint(0, 256)))
        elif g == 3:
            p = corpus.periodic(m, int(rs.choice([1, 2, 3, 7, 100, 258, 259, 1000])), s)
        elif g == 4:
            p = corpus.indented(m, s)
        elif g == 5:
            p = corpus.duplicated(m, s, int(rs.choice([50, 400, 1500, 5000])))
        elif g == 6:
            p = corpus.table_like(m, s)
        elif g == 7:
            p = corpus.json_like(m, s)
        elif g == 8:
            p = corpus.selftest_data(m, s, int(rs.choice([2, 3, 15, 64, 256])), float(rs.choice([0.0, 0.3, 0.5, 0.9])))
        elif g == 9:
            p = corpus.sparse_ones(m, s, int(rs.choice([20, 200, 3000])))
        elif g == 10 and parts:
            q = parts[int(rs.randint(0, len(parts)))]   # an earlier piece again: long-distance copies
            p = q[:m]
        else:
            p = corpus.text_like_fast(max(m, 64), s)[:m]
        p = np.ascontiguousarray(p, dtype=np.uint8)[:m]
        if len(p) == 0:
            p = np.zeros(1, dtype=np.uint8)
        parts.append(p)
        n += len(p)
    d = np.concatenate(parts)[:size]
    flags = int(rs.choice([0, 1, 2]))
    bs = int(rs.choice([0, 32768, 65536, 65536, 131072]))
    return d, flags, bs


def ref_one(args):
    seed, k, max_bytes = args
    import zlibs
    d, flags, bs = make_case(seed, k, max_bytes)
    out = zlibs.Ref().memory_compress(d, flags, bs)
    return k, zlib.crc32(out) if out is not None else None, len(out) if out is not None else -1, d.tobytes(), flags, bs   # (the generators are slow: the input travels back too)


def ref_file(args):
    seed, k = args
    import zlibs
    rs = np.random.RandomState((seed * 7919 + k) & 0x7fffffff)
    n = int(rs.choice([1, 2, 3, 17, 100, 1000, 4095, 4096])) if rs.randint(0, 4) == 0 else int(rs.randint(1, 4097))
    g = rs.randint(0, 6)
    s_ = int(rs.randint(1, 1 << 20))
    d = [corpus.json_like, corpus.text_like, corpus.noise, corpus.indented, corpus.table_like][g % 5](n, s_) if g < 5 else corpus.constant(n, int(rs.randint(0, 256)))
    d = np.ascontiguousarray(d, dtype=np.uint8)[:n]
    out = zlibs.Ref().memory_compress(d, 0, 32768)
    return k, d.tobytes(), out


def files_mode(nfiles, seed, batch=4096):
    """files mode (zultra_hip_compress_files, BASELINE configuration 5): inputs of 1 .. 4096 bytes, one raw deflate stream each, in batches of 4096"""
    t0 = time.time()
    with mp.get_context("fork").Pool(min(64, os.cpu_count() or 1)) as pool:
        refs = dict((r[0], r[1:]) for r in pool.imap_unordered(ref_file, [(seed, k) for k in range(nfiles)], chunksize=64))
    import zultra_amd
    L = zultra_amd.lib()
    ctx = L.files_context(4096, batch)
    for b0 in range(0, nfiles, batch):
        ks = list(range(b0, min(nfiles, b0 + batch)))
        files = [np.frombuffer(refs[k][0], dtype=np.uint8) for k in ks]
        sizes = [len(f) for f in files]
        offs = np.cumsum([0] + sizes[:-1])
        fo = ctx.compress_files(np.concatenate(files), offs, sizes)
        stream = ctx.stream_read(int(fo[-1]))
        for i, k in enumerate(ks):
            got = stream[int(fo[i]):int(fo[i + 1])].tobytes()
            if got != refs[k][1]:
                print("DIFFERENT: files mode, seed %d file %d (%d bytes): %d bytes against the reference's %d" % (seed, k, sizes[i], len(got), len(refs[k][1])))
                sys.exit(1)
    ctx.close()
    print("fuzz_gpu --files: %d inputs of 1 .. 4096 bytes (seed %d) in batches of %d, every stream identical to the compiled reference's; %.1f s" % (nfiles, seed, batch, time.time() - t0))


def stream_mode(ncases, seed, max_bytes):
    """the streaming API (zultra_stream_compress) fed in random chunks — 1 byte to a few hundred KB, now and then an empty one — against the
    reference's one-shot output for the same input, flags and max-block size"""
    t0 = time.time()
    with mp.get_context("fork").Pool(min(64, os.cpu_count() or 1)) as pool:
        refs = dict((r[0], r[1:]) for r in pool.imap_unordered(ref_one, [(seed, k, max_bytes) for k in range(ncases)], chunksize=2))
    import zultra_amd
    L = zultra_amd.lib()
    total = 0
    for k in range(ncases):
        c, n, raw, flags, bs = refs[k]
        d = np.frombuffer(raw, dtype=np.uint8)
        rs = np.random.RandomState(seed * 31 + k)
        st = L.stream(flags, bs)
        out, pos = bytearray(), 0
        while True:
            m = int(rs.choice([0, 1, 7, 4096, 16384, 65536, 100000, 300000])) if rs.randint(0, 3) else int(rs.randint(1, 70000))
            chunk = d[pos:pos + m]
            pos += len(chunk)
            last = pos >= len(d)
            rc, o = st.compress(chunk, last, out_chunk=int(rs.choice([1 << 12, 1 << 16, 1 << 20])))
            out += o
            if last:
                break
        st.end()
        total += len(d)
        if len(out) != n or zlib.crc32(bytes(out)) != c:
            print("DIFFERENT: streaming API, seed %d case %d: %d bytes, flags %d, max block %d: %d bytes against the reference's %d" % (seed, k, len(d), flags, bs, len(out), n))
            sys.exit(1)
    print("fuzz_gpu --stream: %d cases (seed %d), %.1f MB fed in random chunks, all byte-identical to the compiled reference; %.1f s" % (ncases, seed, total / 1e6, time.time() - t0))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--files":
        return files_mode(int(sys.argv[2]) if len(sys.argv) > 2 else 20000, int(sys.argv[3]) if len(sys.argv) > 3 else 1, int(sys.argv[4]) if len(sys.argv) > 4 else 4096)
    if len(sys.argv) > 1 and sys.argv[1] == "--stream":
        return stream_mode(int(sys.argv[2]) if len(sys.argv) > 2 else 200, int(sys.argv[3]) if len(sys.argv) > 3 else 1, int(sys.argv[4]) if len(sys.argv) > 4 else 1_500_000)
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    max_bytes = int(sys.argv[3]) if len(sys.argv) > 3 else 3_000_000
    t0 = time.time()
    with mp.get_context("fork").Pool(min(64, os.cpu_count() or 1)) as pool:   # before any GPU call of this process
        refs = dict((r[0], r[1:]) for r in pool.imap_unordered(ref_one, [(seed, k, max_bytes) for k in range(ncases)], chunksize=2))
    t1 = time.time()
    import zultra_amd
    L = zultra_amd.lib()
    total = 0
    for k in range(ncases):
        c, n, raw, flags, bs = refs[k]
        d = np.frombuffer(raw, dtype=np.uint8)
        got = L.memory_compress(d, flags, bs)
        total += len(d)
        if got is None or len(got) != n or zlib.crc32(got) != c:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            path = os.path.join(ROOT, "gpurun_out", "fuzz_fail_%d_%d.bin" % (seed, k))
            d.tofile(path)
            print("DIFFERENT: seed %d case %d: %d bytes, flags %d, max block %d: %s bytes against the reference's %d -> %s" % (
                seed, k, len(d), flags, bs, None if got is None else len(got), n, path))
            sys.exit(1)
    print("fuzz_gpu: %d cases (seed %d), %.1f MB, all byte-identical to the compiled reference; reference pool %.1f s, device %.1f s" % (
        ncases, seed, total / 1e6, t1 - t0, time.time() - t1))


if __name__ == "__main__":
    main()
