"""CPU: the oracle (oracle/zultra_oracle.c) against the golden vectors produced by the compiled reference."""
import hashlib

import numpy as np
import pytest

import golden_util as G


@pytest.mark.parametrize("name", G.stream_names())
def test_stream_matches_reference_bytes(oracle, name):
    c = G.stream_case(name)
    got = oracle.memory_compress(c["data"], c["flags"], c["max_block"], c["dictionary"])
    G.check_stream_output(c, got)


@pytest.mark.parametrize("name", G.stage_names())
def test_stages_match_reference_intermediates(oracle, name):
    e, z = G.stage_case(name)
    win, prev, n = z["win"], e["prev"], e["n"]
    m = oracle.find_matches(win, prev, n)
    assert np.array_equal(m, z["match"]), "match rows differ"
    assert oracle.split(win, z["match"], prev, n) == e["splits"]
    for k, sb in enumerate(e["subblocks"]):
        dyn, sc, dc = oracle.costs(win, z["match"], prev, sb["start"], sb["size"])
        assert (dyn, sc, dc) == (sb["is_dynamic"], sb["static_cost"], sb["dynamic_cost"])
        rc, nb, bits, best, ll, dl = oracle.deflate(win, z["match"], prev, sb["start"], sb["size"], dyn)
        assert rc == 0 and nb == sb["nbits"]
        assert np.array_equal(best, z["best%d" % k]), "final parse differs"
        assert np.array_equal(ll, z["litlen%d" % k]) and np.array_equal(dl, z["distlen%d" % k])
        assert bits == z["bits%d" % k].tobytes()
        assert hashlib.sha256(bits).hexdigest() == sb["bits_sha256"]


def test_empty_input_is_an_error(oracle):
    # libzultra.c:275: an empty input never reaches the finalized state -> zultra_memory_compress returns -1
    assert oracle.memory_compress(np.zeros(0, dtype=np.uint8), 2, 65536) is None


def test_output_buffer_too_small(oracle):
    c = G.stream_case("tiny_100")
    for cap in range(0, 12):   # tool/zultra.c:521-524 feeds 0..11 byte buffers
        assert oracle.memory_compress(c["data"], 1, 0, cap=cap) is None


def test_checksums_known_answers(oracle):
    import zlib
    d = G.stream_case("json_4k")["data"]
    assert oracle.crc32(d) == zlib.crc32(d.tobytes())
    assert oracle.adler32(d) == zlib.adler32(d.tobytes())
    assert oracle.crc32(b"123456789") == 0xCBF43926
