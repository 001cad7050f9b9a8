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
