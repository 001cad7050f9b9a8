"""CPU: the corpora bench.py shards across ranks form ONE stream — rank r's 32 KiB of history are the last 32 KiB of rank r-1's
shard — and the bulk generators of tests/gen/zgen.c are functions of (seed, unit index) only, so any rank can produce any range."""
import os
import sys
import zlib

import numpy as np

import corpus

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_config4_segments_depend_on_their_index_only():
    a = corpus.mixed_config4(0, 20)
    b = corpus.mixed_config4(7, 3)
    assert np.array_equal(a[7 << 20: 10 << 20], b)
    assert not np.array_equal(corpus.mixed_config4(2, 1, seed=1), a[2 << 20: 3 << 20])   # (segment 0 has an alphabet of one symbol)
    ratios = [len(zlib.compress(a[k << 20:(k + 1) << 20].tobytes(), 1)) / (1 << 20) for k in range(16)]
    assert ratios[14] > 0.99 and ratios[15] < 0.01          # every 16th segment noise (stored fallback), every 16th one repeated byte
    assert len({round(r, 2) for r in ratios}) >= 8            # the self-test grid in between: a spread of entropies


def test_json_files_are_independent_and_json_like():
    f = corpus.json_files(100, 8)
    g = corpus.json_files(103, 2)
    assert np.array_equal(f[3 * 4096: 5 * 4096], g)
    text = bytes(f[:4096])
    assert text.startswith(b'{"id":') and b'"user":"' in text and b'"tags":[' in text and len(text) == 4096


def test_bench_shards_form_one_stream():
    import bench
    base = corpus.text_like(300000, 3)
    cyc = bench.CyclicCorpus("t", base)
    for cls, size in ((cyc, 131072), (bench.SyntheticText(), 100000), (bench.MixedConfig4(), 2 << 20)):
        prev_shard = None
        for rank in range(3):
            lead, shard = cls.shard(rank, size)
            assert len(shard) == size
            if rank == 0:
                assert lead is None
            else:
                assert len(lead) == 32768 and np.array_equal(lead, prev_shard[-32768:]), (type(cls).__name__, rank)
            prev_shard = shard
    # the cyclic corpus wraps around its base
    lead, shard = cyc.shard(3, 131072)
    assert np.array_equal(shard[:10], base[(3 * 131072) % 300000:][:10])


def test_frame_and_inflate_check():
    import bench
    d = corpus.text_like(50000, 9)
    raw = zlib.compressobj(9, zlib.DEFLATED, -15)
    body = raw.compress(d.tobytes()) + raw.flush()
    gz = bench.frame(None, 2, body, zlib.crc32(d.tobytes()), len(d))
    assert bench.inflate_check(2, gz, d, len(d))
    zz = bench.frame(None, 1, body, zlib.adler32(d.tobytes()), len(d))
    assert bench.inflate_check(1, zz, d, len(d))
    assert not bench.inflate_check(2, gz[:-8] + b"\0" * 8, d, len(d))       # wrong CRC: zlib refuses
    assert not bench.inflate_check(0, body, d[:-1], len(d) - 1)            # wrong length


def test_strong_scaling_shards_are_one_stream_cut_at_max_blocks():
    """bench.py --scaling strong (and the strong leg beside the weak line at N > 1): the configuration's ONE stream is cut over the
    ranks at max-block boundaries (zultra_amd.sharded.shard_range); the ranks' shards concatenate to the stream, every lead is the
    32 KiB in front of its shard, ranks behind the last max-block are empty."""
    import argparse

    import bench
    from zultra_amd.sharded import shard_range
    base = corpus.text_like(250000, 4)
    cyc = bench.CyclicCorpus("t", base)
    lead, body = cyc.range(131072 + 777, 50000)
    whole = bench._cyclic(base, 0, 1_100_000)
    assert np.array_equal(body, whole[131072 + 777: 131072 + 777 + 50000]) and np.array_equal(lead, whole[131072 + 777 - 32768: 131072 + 777])
    m4 = bench.MixedConfig4()
    lead, body = m4.range((3 << 20) + 65536 * 5, (2 << 20) + 4096)
    ref = corpus.mixed_config4(2, 5)   # segments 2..6
    lo = (1 << 20) + 65536 * 5
    assert np.array_equal(body, ref[lo: lo + (2 << 20) + 4096]) and np.array_equal(lead, ref[lo - 32768: lo])
    # the cut bench.py makes: prepare_stream_config with a corpus of our own (no files of the image needed)
    orig = bench.text_corpus
    bench.text_corpus = lambda world, size: (cyc, "test corpus")
    try:
        size, bs = 1_000_000, 65536
        nb_total = (size + bs - 1) // bs
        for world in (1, 2, 5, 8, 20):
            parts = []
            for rank in range(world):
                args = argparse.Namespace(config=2, scaling="strong", block=bs, size=size, cpu_sample=1 << 20, no_cpu_baseline=True, profile_run=False)
                p = bench.prepare_stream_config(args, rank, world)
                lo, hi = shard_range(nb_total, rank, world)
                assert len(p["shard"]) == max(0, min(size, hi * bs) - lo * bs)
                if len(p["shard"]) and lo:
                    assert np.array_equal(p["lead"], whole[lo * bs - 32768: lo * bs])
                assert p["last_rank"] == max(r for r in range(world) if shard_range(nb_total, r, world)[1] > shard_range(nb_total, r, world)[0])
                parts.append(p["shard"])
            assert np.array_equal(np.concatenate(parts), whole[:size])
    finally:
        bench.text_corpus = orig
