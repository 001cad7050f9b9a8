"""CPU: pin the oracle against the reference itself (oracle/_ref, compiled from /root/reference),
stage by stage and on whole streams, over seeded inputs covering the reference's self-test grid
(tool/zultra.c:529-534) and the edge cases of SURVEY.md Appendix A. Skipped where oracle/_ref is absent."""
import zlib

import numpy as np
import pytest

import corpus


def _stage_check(oracle, ref, win, prev, n, max_block=65536):
    win = np.ascontiguousarray(win, dtype=np.uint8)
    mo = oracle.find_matches(win, prev, n)
    with ref.probe(max_block, win, prev, n) as P:
        mr = P.matches()
        assert np.array_equal(mo, mr), "match rows"
        sr = P.split()
        assert oracle.split(win, mr, prev, n) == sr
        at = prev
        for e in sr:
            size = e - at
            assert oracle.costs(win, mr, prev, at, size) == P.costs(at, size)
            for dyn in (0, 1):
                ro = oracle.deflate(win, mr, prev, at, size, dyn)
                rr = P.deflate(at, size, dyn)
                assert ro[0] == 0 and ro[1] == rr[1] and ro[2] == rr[2]
                assert np.array_equal(ro[3], rr[3]) and np.array_equal(ro[4], rr[4]) and np.array_equal(ro[5], rr[5])
            at = e


def test_stages_text_with_history(oracle, ref):
    t = corpus.text_like(98304, 3)
    _stage_check(oracle, ref, t, 32768, 65536)


def test_stages_text_first_block(oracle, ref):
    _stage_check(oracle, ref, corpus.text_like(65536, 4), 0, 65536)


@pytest.mark.parametrize("alphabet", [1, 2, 15, 96, 256])
@pytest.mark.parametrize("prob", [0.0, 0.5, 0.995])
def test_stages_selftest_grid(oracle, ref, alphabet, prob):
    _stage_check(oracle, ref, corpus.selftest_data(24576, 123, alphabet, prob), 8192, 16384, 32768)


def test_stages_degenerate(oracle, ref):
    _stage_check(oracle, ref, corpus.constant(40000), 4464, 35536)
    _stage_check(oracle, ref, corpus.periodic(30000, 3), 0, 30000)
    _stage_check(oracle, ref, corpus.noise(20000, 1), 0, 20000)
    _stage_check(oracle, ref, corpus.sparse_ones(40000), 8000, 32000)
    _stage_check(oracle, ref, corpus.text_like(10, 1), 0, 10)
    _stage_check(oracle, ref, corpus.text_like(3, 1), 0, 3)
    _stage_check(oracle, ref, corpus.text_like(1, 1), 0, 1)


def _stream_check(oracle, ref, data, flags, bs, d=None):
    a = oracle.memory_compress(data, flags, bs, d)
    b = ref.memory_compress(data, flags, bs, d)
    assert a == b
    if b is not None:
        wb = {0: -15, 1: 15, 2: 31}[flags]
        dec = zlib.decompressobj(wb, zdict=bytes(d)) if (d is not None and flags != 2) else zlib.decompressobj(wb)
        assert dec.decompress(b) == bytes(np.ascontiguousarray(data, dtype=np.uint8).tobytes())


@pytest.mark.parametrize("flags,bs", [(2, 65536), (1, 32768), (0, 0)])
def test_stream_text(oracle, ref, flags, bs):
    _stream_check(oracle, ref, corpus.text_like(250000, 9), flags, bs)


def test_stream_phase_dependent_stored_fallback(oracle, ref):
    t = corpus.text_like(150000, 7)
    d = np.concatenate([t[:70000], corpus.noise(140000, 3), t[70000:]])
    for flags, bs in [(2, 65536), (0, 32768), (1, 1 << 20)]:
        _stream_check(oracle, ref, d, flags, bs)


def test_stream_dictionary(oracle, ref):
    t = corpus.text_like(140000, 7)
    _stream_check(oracle, ref, t[40000:], 1, 65536, t[:32768])
    _stream_check(oracle, ref, t[40000:90000], 0, 65536, t[100:5000])


def test_stream_small_and_ragged(oracle, ref):
    t = corpus.text_like(70000, 2)
    for n in (1, 2, 3, 4, 100, 32768, 32769, 65535, 65536, 65537):
        _stream_check(oracle, ref, t[:n], 2, 32768)
    assert oracle.memory_compress(t[:0], 2, 0) is None and ref.memory_compress(t[:0], 2, 0) is None


def test_stream_output_too_small_fails_like_reference(oracle, ref):
    t = corpus.text_like(4096, 2)
    for cap in range(0, 12):
        assert oracle.memory_compress(t, 1, 0, cap=cap) is None
        assert ref.memory_compress(t, 1, 0, cap=cap) is None


def test_fuzz_streams(oracle, ref):
    rs = np.random.RandomState(2024)
    for it in range(12):
        n = int(rs.randint(1, 120000))
        kind = it % 4
        if kind == 0:
            d = corpus.mixed(n, 100 + it)
        elif kind == 1:
            d = corpus.selftest_data(n, 200 + it, corpus.SELFTEST_ALPHABETS[it % 12], [0.0, 0.3, 0.7, 0.995][it % 4])
        elif kind == 2:
            d = corpus.text_like(n, 300 + it)
        else:
            d = np.concatenate([corpus.json_like(n // 2, it), corpus.noise(n - n // 2, it)])
        _stream_check(oracle, ref, d, int(rs.randint(0, 3)), [0, 32768, 65536][it % 3])


def test_checksums(oracle, ref):
    d = corpus.text_like(100003, 5)
    assert oracle.crc32(d) == ref.checksum(d, 2, 0)
    assert oracle.adler32(d) == ref.checksum(d, 1, 1)
