"""GPU (MI355X): the product library zultra_amd/libzultra_amd.so, through its C ABI, against the `checker` of tests/conftest.py —
the COMPILED REFERENCE itself (oracle/_ref/libzultra_ref.so travels to the GPU box) behind the oracle's interface, stage by stage and on
whole streams, the pinned restatement only where that file is absent (round 6; rounds 1-5 compared with the restatement) —, against the
golden vectors produced by the compiled reference, and — at full benchmark sizes — through size-independent properties (inflate round
trip, block independence). "The oracle's" in the docstrings below reads "the checker's"."""
import zlib

import numpy as np
import pytest

import corpus
import golden_util as G
from parity_util import check_window

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import zultra_amd
    L = zultra_amd.lib()            # raises if the .so is missing: no fallback
    assert L.device_count() >= 1, "no HIP device visible"
    return L


def test_wave_primitives_selfcheck(gpu):
    # zh_selftest runs the DPP/readlane reductions against plain shuffles on the device
    import ctypes as C
    f = gpu.L.zultra_hip_selftest
    f.restype = C.c_int
    assert f() == 0


@pytest.mark.parametrize("case", [
    ("text_hist", lambda: corpus.text_like(98304, 3), 32768, 65536, 65536),
    ("text_first", lambda: corpus.text_like(65536, 4), 0, 65536, 65536),
    ("text_split", lambda: corpus.text_like(60000, 21)[30000:50000], 0, 20000, 32768),
    ("mixed", lambda: corpus.mixed(98304, 5), 32768, 65536, 65536),
    ("selftest15", lambda: corpus.selftest_data(49152, 123, 15, 0.5), 16384, 32768, 32768),
    ("selftest2", lambda: corpus.selftest_data(49152, 123, 2, 0.5), 16384, 32768, 32768),
    ("selftest256", lambda: corpus.selftest_data(49152, 123, 256, 0.0), 16384, 32768, 32768),
    ("zeros", lambda: corpus.constant(70000), 4464, 65536, 65536),
    ("period3", lambda: corpus.periodic(50000, 3), 0, 50000, 65536),
    ("noise", lambda: corpus.noise(40000, 1), 0, 40000, 65536),
    ("sparse", lambda: corpus.sparse_ones(98304), 32768, 65536, 65536),
    ("tiny", lambda: corpus.text_like(10, 1), 0, 10, 32768),
    ("three", lambda: corpus.text_like(3, 1), 0, 3, 32768),
    ("one", lambda: corpus.text_like(1, 1), 0, 1, 32768),
    ("json4k", lambda: corpus.json_like(4096, 3), 0, 4096, 32768),
    ("big_block", lambda: corpus.text_like(300000, 6), 32768, 267232, 1 << 20),
    ("byte_runs", lambda: corpus.indented(98304, 11), 32768, 65536, 65536),
    ("byte_runs_big", lambda: corpus.indented(300000, 13), 32768, 267232, 1 << 20),
    ("near_copies", lambda: corpus.duplicated(98304, 3), 32768, 65536, 65536),
    ("deep_huffman", lambda: corpus.fibonacci_bytes(22), 0, 46367, 65536),
    ("near_copies_sparse_edits", lambda: corpus.duplicated(98304, 4, 5000), 32768, 65536, 65536),
    ("near_copies_split", lambda: corpus.duplicated(60000, 5, 700), 10000, 50000, 65536),
    ("table_cut", lambda: corpus.table_like(98304, 9), 32768, 65536, 65536),
    ("table_cut_big", lambda: corpus.table_like(300000, 10), 32768, 267232, 1 << 20),
], ids=lambda c: c[0])
def test_stages_vs_oracle(gpu, checker, case):
    name, gen, prev, n, bs = case
    check_window(gpu, checker, gen(), prev, n, max_block=bs, tag=name)


@pytest.mark.parametrize("case", [
    ("text_hist", lambda: corpus.text_like(98304, 3), 32768, 65536, 65536),
    ("byte_runs", lambda: corpus.indented(98304, 11), 32768, 65536, 65536),
    ("near_copies", lambda: corpus.duplicated(98304, 3), 32768, 65536, 65536),
    ("table_cut_big", lambda: corpus.table_like(300000, 10), 32768, 267232, 1 << 20),
], ids=lambda c: c[0])
def test_long_pieces_stay_on_the_quads(gpu, checker, monkeypatch, case):
    """Runs of fewer tasks than CUs send tasks with a barrier-free piece above 256 positions to the chain kernel (ZULTRA_HIP_COOP_SMALL);
    the same windows with the bound of large batches (pieces of up to 1536 positions on the quads of zh_parse_lanes)."""
    name, gen, prev, n, bs = case
    monkeypatch.setenv("ZULTRA_HIP_COOP_SMALL", "1536")
    check_window(gpu, checker, gen(), prev, n, max_block=bs, tag=name + "/coop1536")


def test_fuzz_stitched_inputs_vs_oracle(gpu, checker):
    """A short run of tools/fuzz_gpu.py's generator (inputs stitched together from the corpora: text, byte runs, near copies, tables, noise,
    constant and periodic stretches, earlier pieces again), the three framings, 32 KiB / 64 KiB / 128 KiB / default max-blocks: every stream
    equals the oracle's. (The soak run of that tool against the compiled reference is in profiles/r04_fuzz_gpu.txt.)"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_gpu
    total = 0
    for k in range(25):
        d, flags, bs = fuzz_gpu.make_case(77, k, 400000)
        got = gpu.memory_compress(d, flags, bs)
        assert got == checker.memory_compress(d, flags, bs), (k, len(d), flags, bs)
        total += len(d)
    assert total > 500_000


def test_settled_subblocks_keep_their_parse(gpu, checker):
    """Sub-blocks whose code lengths have reached a fixed point of the reference's four-pass loop (blockdeflate.c:874-901) are not parsed
    again (zh_sb_build_one, st->settled): whole-block chains of a constant byte settle after the first pass, noise after the second, and in
    a batch of the mixed corpus some sub-blocks settle and others never do. Stages and stream are the oracle's either way."""
    st = {}
    check_window(gpu, checker, corpus.constant(70000), 4464, 65536, max_block=65536, tag="zeros", stats_out=st)
    assert st["settled_passes"] == 3 * st["subblocks"], st
    st = {}
    check_window(gpu, checker, corpus.noise(40000, 1), 0, 40000, max_block=65536, tag="noise", stats_out=st)
    assert st["settled_passes"] >= 1, st
    d = corpus.mixed_config4(0, 6)
    ctx = gpu.context(65536, len(d) // 65536)
    try:
        ctx.compress_blocks(d, [(65536 * b - (32768 if b else 0), 32768 if b else 0, 65536) for b in range(len(d) // 65536)])
        st = ctx.stats()
    finally:
        ctx.close()
    assert 0 < st["settled_passes"] < 3 * st["subblocks"], st
    got = gpu.memory_compress(d, 2, 65536)
    assert got == checker.memory_compress(d, 2, 65536)


@pytest.mark.parametrize("wide", ["1", "1000000", "whole"], ids=["as_waves_of_zh_parse_segments", "as_jobs_of_zh_parse_chain", "few_and_short_stay_whole"])
def test_chain_tasks_are_cut_into_speculative_segments(gpu, checker, monkeypatch, wide):
    """zh_parse.h: barrier-free runs of table-like text are parsed as segments started 1024 positions early — as waves of
    zh_parse_segments or as jobs of zh_parse_chain, by the number of segments in the run (ZULTRA_HIP_SEG_WIDE); the parse is the
    oracle's either way and most of the cuts must verify."""
    monkeypatch.setenv("ZULTRA_HIP_SEG_WIDE", "1000000" if wide == "whole" else wide)
    monkeypatch.setenv("ZULTRA_HIP_SEG_WHOLE", "1000000" if wide == "whole" else "0")
    check_window(gpu, checker, corpus.table_like(98304, 9), 32768, 65536, 65536, tag="table_cut/" + wide)
    check_window(gpu, checker, corpus.table_like(300000, 10), 32768, 267232, 1 << 20, tag="table_cut_big/" + wide)
    ctx = gpu.context(65536, 4)
    try:
        data = corpus.table_like(4 * 65536, 12)
        ctx.compress_blocks(data, [(65536 * b, 0, 65536) for b in range(4)])
        st = ctx.stats()
    finally:
        ctx.close()
    cuts = st["cut_segments"] - st["cut_tasks"]
    assert st["cut_tasks"] >= 1 and cuts >= 8, st
    assert st["cut_redone"] < 2 * cuts, st   # of 4 * cuts checks


def test_large_max_block_is_cut_into_matchfinder_segments(gpu, checker):
    # windows of 20000 + 80000 and 32768 + 200000 bytes exceed the 96 KiB LDS window: two / four segments, all but the last
    # with 258 bytes of look-ahead; byte runs and ordinary text straddle the cuts
    d = np.concatenate([corpus.text_like(60000, 31), corpus.indented(25000, 5), corpus.text_like(15000, 32)])
    check_window(gpu, checker, d, 20000, 80000, max_block=131072, tag="segments2")
    d = np.concatenate([corpus.indented(70000, 7), corpus.text_like(100000, 33), corpus.indented(62768, 8)])
    check_window(gpu, checker, d, 32768, 200000, max_block=262144, tag="segments4")


@pytest.mark.parametrize("name", G.stream_names())
def test_golden_streams(gpu, name):
    c = G.stream_case(name)
    G.check_stream_output(c, gpu.memory_compress(c["data"], c["flags"], c["max_block"], c["dictionary"]))


@pytest.mark.parametrize("flags,bs", [(2, 65536), (1, 32768), (0, 0)])
def test_stream_vs_oracle_multiblock(gpu, checker, flags, bs):
    d = np.concatenate([corpus.text_like(300000, 9), corpus.noise(90000, 2), corpus.mixed(200000, 3)])
    got = gpu.memory_compress(d, flags, bs)
    assert got == checker.memory_compress(d, flags, bs)


def test_fuzz_streams_vs_oracle(gpu, checker):
    rs = np.random.RandomState(77)
    for it in range(10):
        n = int(rs.randint(1, 150000))
        kind = it % 4
        if kind == 0:
            d = corpus.mixed(n, 500 + it)
        elif kind == 1:
            d = corpus.selftest_data(n, 600 + it, corpus.SELFTEST_ALPHABETS[it % 12], [0.0, 0.3, 0.7, 0.995][it % 4])
        elif kind == 2:
            d = corpus.text_like(n, 700 + it)
        else:
            d = np.concatenate([corpus.json_like(n // 2, it), corpus.noise(n - n // 2, it)])
        flags, bs = int(rs.randint(0, 3)), [0, 32768, 65536][it % 3]
        assert gpu.memory_compress(d, flags, bs) == checker.memory_compress(d, flags, bs), (it, n, flags, bs)


@pytest.mark.parametrize("wide", ["1", "1000000", "whole"], ids=["as_waves_of_zh_parse_segments", "as_jobs_of_zh_parse_chain", "few_and_short_stay_whole"])
def test_fuzz_streams_with_long_barrier_free_runs(gpu, checker, monkeypatch, wide):
    """Whole streams made of the kinds of data that produce chains and cut tasks (tables, records, near-copies, byte runs,
    periodic data) next to data that produces none, under each way of parsing the segments: the bytes are the oracle's."""
    monkeypatch.setenv("ZULTRA_HIP_SEG_WIDE", "1000000" if wide == "whole" else wide)
    monkeypatch.setenv("ZULTRA_HIP_SEG_WHOLE", "1000000" if wide == "whole" else "0")
    monkeypatch.setenv("ZULTRA_HIP_CACHE", "0")   # the thresholds are read when a context is created
    gpu.L.zultra_release_cached_contexts()
    rs = np.random.RandomState(4242)
    makers = [lambda n, sd: corpus.table_like(n, sd), lambda n, sd: corpus.duplicated(n, sd, int(rs.randint(40, 3000))),
              lambda n, sd: corpus.json_like(n, sd), lambda n, sd: corpus.indented(n, sd), lambda n, sd: corpus.periodic(n, 1 + sd % 7),
              lambda n, sd: corpus.noise(n, sd), lambda n, sd: corpus.text_like(n, sd), lambda n, sd: corpus.constant(n, 32 + sd % 64)]
    for it in range(4):
        parts = []
        for _ in range(int(rs.randint(3, 8))):
            parts.append(makers[int(rs.randint(0, len(makers)))](int(rs.randint(2000, 160000)), int(rs.randint(1, 1000))))
        d = np.concatenate(parts)
        flags, bs = int(rs.randint(0, 3)), [65536, 32768, 0, 262144][it % 4]
        assert gpu.memory_compress(d, flags, bs) == checker.memory_compress(d, flags, bs), (wide, it, len(d), flags, bs)


def test_streaming_api_chunking(gpu, checker):
    d = corpus.text_like(500000, 12)
    want = checker.memory_compress(d, 2, 65536)
    for chunk in (16384, 100000, 500000):
        s = gpu.stream(2, 65536)
        out = bytearray()
        pos, st = 0, 0
        while pos < len(d):
            part = d[pos:pos + chunk]
            pos += len(part)
            st, b = s.compress(part, finalize=(pos >= len(d)), out_chunk=16384)
            out += b
        assert st == 1
        s.end()
        assert bytes(out) == want


def test_device_scan_equals_the_host_planner_at_every_phase(gpu):
    """The stitcher's device scan (zh_stitch_scan) against the serial host planner at all eight start phases, and the phase table a rank
    hands its neighbours: the emulator suite's check (test_emu_parity._scan_vs_planner) on the real kernels."""
    from test_emu_parity import _scan_vs_planner
    _scan_vs_planner(gpu)


def test_strided_overflow_forms_of_the_per_item_kernels(gpu, checker, monkeypatch):
    """ZULTRA_HIP_GRID_CAP = 1 / 3: nearly every sub-block and task goes through the <true> forms of the per-sub-block / per-task kernels (the emulator
    suite's check on the real kernels, with 41 sub-blocks in one max-block)."""
    from test_emu_parity import _overflow_forms
    _overflow_forms(gpu, checker, monkeypatch, (32768, 32, ("1", "3"), ((2, 32768), (0, 65536))))


def test_a_run_that_outgrows_its_grids_is_run_again(gpu, checker, monkeypatch):
    """No <true> overflow forms for a run whose counterpart in the last batch stayed inside the grids; a run that outgrows them is void and the batch is run again (the
    emulator suite's check on the real kernels, with grids capped at 64)."""
    from test_emu_parity import _outgrown_grids
    _outgrown_grids(gpu, checker, monkeypatch, 64, corpus.text_like(100000, 5), np.concatenate([corpus.text_like(400000, 6), corpus.table_like(100000, 3)]), 65536)


def test_chains_turn_up_after_a_batch_without_any(gpu, checker):
    """The chain grid of a run follows what the context's last batch listed (a few workgroups after a batch without chains): 13 MiB in three runs, without chains, then
    with, then without, in several orders — the emulator suite's check on the real kernels, with what the stats say about chain kernels and reruns."""
    from test_emu_parity import _chains_after_none
    _chains_after_none(gpu, checker, 13 << 20, 65536, "ppccpcpp")


@pytest.mark.parametrize("runs", ["1", "2", "3"])
def test_a_run_needs_no_host_decision(gpu, checker, monkeypatch, runs):
    """Sub-block counts, task counts, chains and cut tasks are summed up on the device (zh_plan_subblocks, zh_list_huge) and every later kernel takes
    its bounds from the run's counters; the host sizes grids from the input bytes alone. A batch whose max-blocks split very unevenly — one max-block of
    forty short stretches of different statistics next to max-blocks that do not split at all, then 12 MiB of text — as one, two and three runs
    (ZULTRA_HIP_STREAMS; the run count is asserted): the bytes are the checker's."""
    parts = [corpus.text_like(65536, 3)]
    rng = np.random.default_rng(5)
    many = np.concatenate([corpus.selftest_data(1600, 100 + k, int(rng.integers(2, 200)), float(rng.uniform(0.0, 0.9))) for k in range(41)])[:65536]
    parts += [many, corpus.constant(65536, 7), corpus.duplicated(65536, 4, 900), corpus.noise(65536, 9), corpus.indented(65536, 2)] * 2
    parts += [corpus.text_like_fast(12 << 20, 77)]   # (a batch is cut into as many runs as it has multiples of 4 MiB, at most ZULTRA_HIP_STREAMS: 13 MiB for three)
    d = np.concatenate(parts)
    monkeypatch.setenv("ZULTRA_HIP_STREAMS", runs)
    want = checker.memory_compress(d, 2, 65536)
    assert gpu.memory_compress(d, 2, 65536) == want
    # ... and the device layer did cut the batch into that many runs (zh_compact_results lays their descriptors end to end)
    bs = 65536
    nb = (len(d) + bs - 1) // bs
    ctx = gpu.context(bs, nb)
    try:
        ctx.compress_blocks(d, [(b * bs - (32768 if b else 0), 32768 if b else 0, min(bs, len(d) - b * bs)) for b in range(nb)])
        assert ctx.stats()["runs"] == int(runs), ctx.stats()
    finally:
        ctx.close()


def test_token_chain_chunk_size_follows_the_batch(gpu, checker):
    """A batch of at most 32 max-blocks follows its token chain in chunks of 2048 positions, a larger one in chunks of 16 Ki (zh_split.h: the chunk size is a
    kernel argument since round 5). The same bytes at 32 and at 33 max-blocks — data with barrier-free stretches longer than a small chunk, matches that cross
    chunk boundaries, and a ragged last block — are the oracle's streams."""
    base = np.concatenate([corpus.text_like(40000, 5), corpus.duplicated(70000, 6, 700), corpus.constant(9000, 3), corpus.indented(50000, 8),
                           corpus.periodic(30000, 7, 2), corpus.json_like(60000, 4)])
    d = np.resize(base, 33 * 32768 - 5000)
    for nblocks in (32, 33):
        part = d[: nblocks * 32768 - 5000]
        assert gpu.memory_compress(part, 1, 32768) == checker.memory_compress(part, 1, 32768), nblocks


def test_streaming_output_cadence(gpu, checker, monkeypatch):
    """libzultra.c:424-462 publishes output after every max-block; the device build collects max-blocks into batches, but a caller that
    feeds small pieces with ZULTRA_CONTINUE sees output once ZULTRA_HIP_FLUSH_BYTES of full blocks are staged (default 4 MiB), not
    only when the 64 MiB staging area is full or at ZULTRA_FINALIZE. The bytes are the reference's either way."""
    d = corpus.text_like(9 * 65536 + 1000, 17)
    want = checker.memory_compress(d, 2, 65536)
    for flush, early in (("131072", True), (None, False), ("0", False)):   # (default 4 MiB: more than this input)
        if flush is None:
            monkeypatch.delenv("ZULTRA_HIP_FLUSH_BYTES", raising=False)
        else:
            monkeypatch.setenv("ZULTRA_HIP_FLUSH_BYTES", flush)
        s = gpu.stream(2, 65536)
        out = bytearray()
        before_final = 0
        pos = 0
        while pos < len(d):
            part = d[pos:pos + 16384]
            pos += len(part)
            st, b = s.compress(part, finalize=(pos >= len(d)), out_chunk=65536)
            out += b
            if pos < len(d):
                before_final = len(out)
        s.end()
        assert bytes(out) == want
        assert (before_final > 10) == early, (flush, before_final)   # (10 = the gzip header)


def test_errors(gpu):
    t = corpus.text_like(100, 1)
    assert gpu.memory_compress(t[:0], 2, 0) is None
    for cap in range(0, 12):
        assert gpu.memory_compress(t, 1, 0, cap=cap) is None


def test_full_size_properties(gpu, checker):
    """Benchmark-sized input (the oracle would take minutes): inflate round trip, size sanity versus zlib -9, and
    block independence — the same max-blocks compressed in two different batch splits give identical bits."""
    d = corpus.text_like(24 << 20, 31)
    out = gpu.memory_compress(d, 2, 65536)
    assert out is not None
    assert zlib.decompress(out, 31) == d.tobytes()
    z9 = len(zlib.compress(d.tobytes(), 9))
    assert len(out) < z9, (len(out), z9)
    # streaming in 1 MiB chunks goes through different batch boundaries; the bytes must not change
    s = gpu.stream(2, 65536)
    acc = bytearray()
    for pos in range(0, len(d), 1 << 20):
        st, b = s.compress(d[pos:pos + (1 << 20)], finalize=(pos + (1 << 20) >= len(d)), out_chunk=1 << 20)
        acc += b
    s.end()
    assert bytes(acc) == out
    # spot-check a few max-blocks of the big input against the oracle, stage by stage
    for blk in (0, 7, 200):
        lo = blk * 65536
        prev = 32768 if blk else 0
        check_window(gpu, checker, d[lo - prev: lo + 65536], prev, 65536, max_block=65536, tag="blk%d" % blk)


@pytest.mark.parametrize("runs", ["default", "3"])
def test_files_mode_graph_replay_vs_oracle(gpu, checker, monkeypatch, runs):
    """BASELINE configuration 5 in miniature: many small JSON-like inputs, one raw deflate stream each, the kernel sequence
    captured in a hipGraph on the first batch and replayed on the second (different contents, same batch shape) — as one run of
    inputs (the default for so few) and as three staggered runs forked inside the graph."""
    if runs != "default":
        monkeypatch.setenv("ZULTRA_HIP_STREAMS", runs)
    nfiles = 300
    ctx = gpu.files_context(4096, nfiles)
    try:
        for batch in range(2):
            rs = np.random.RandomState(50 + batch)
            sizes = [4096] * 200 + [int(x) for x in rs.randint(1, 4097, size=nfiles - 200)]
            files = [corpus.json_like(n, 1000 * batch + k) if k % 7 else corpus.noise(n, k) for k, n in enumerate(sizes)]
            data = np.concatenate(files)
            offs = np.cumsum([0] + sizes[:-1])
            fo = ctx.compress_files(data, offs, sizes)
            assert ctx.stats()["runs"] == (1 if runs == "default" else int(runs))
            stream = ctx.stream_read(int(fo[-1]))
            crcs = ctx.block_crc32()
            for k in range(0, nfiles, 1 if batch == 0 else 3):
                got = stream[int(fo[k]):int(fo[k + 1])].tobytes()
                assert zlib.decompress(got, -15) == files[k].tobytes(), (batch, k)
                if k % 5 == 0:
                    assert got == checker.memory_compress(files[k], 0, 32768), (batch, k)
                assert gpu.crc32_append(0, crcs[k], sizes[k]) == zlib.crc32(files[k].tobytes())
            for k, (a, bw) in enumerate(ctx.block_adler32()):
                assert gpu.adler32_append(1, a, bw, sizes[k]) == zlib.adler32(files[k].tobytes()), (batch, k)
    finally:
        ctx.close()


def test_files_mode_large_batches_are_two_runs_of_graphs(gpu, checker):
    """A files batch of 8192 inputs or more runs as two staggered runs, each replayed from two captured graphs (zh_run_files); two sets of
    graphs are kept, so a caller's full batches and its last, shorter one alternate without re-capturing. Three batches — full, short,
    full with other contents: every stream inflates to its input, every 40th equals the oracle's."""
    ctx = gpu.files_context(2048, 9000)
    try:
        for batch, nfiles in enumerate((9000, 8500, 9000)):
            rs = np.random.RandomState(70 + batch)
            sizes = [int(x) for x in rs.randint(1, 1500, size=nfiles)]
            blob = corpus.json_like(sum(sizes) + 16, 900 + batch)
            offs = np.cumsum([0] + sizes[:-1])
            fo = ctx.compress_files(blob, offs, sizes)
            assert ctx.stats()["runs"] == 2
            stream = ctx.stream_read(int(fo[-1]))
            for k in range(nfiles):
                got = stream[int(fo[k]):int(fo[k + 1])].tobytes()
                want = blob[int(offs[k]):int(offs[k]) + sizes[k]]
                assert zlib.decompress(got, -15) == want.tobytes(), (batch, k)
                if k % 40 == 0:
                    assert got == checker.memory_compress(want, 0, 32768), (batch, k)
    finally:
        ctx.close()


def _edge_inputs():
    """Every size 1..160 and a spread of larger ones, over alphabets of 1, 2, 3 and 5 symbols and a run-heavy mix: window
    ends, the last five positions of a window (which the 6-gram order does not hold), byte runs of every residue, and
    all the short-length record rules of the matchfinder in many combinations."""
    rs = np.random.RandomState(7)
    out = []
    for n in list(range(1, 161)) + [200, 255, 256, 257, 258, 259, 260, 300, 511, 512, 513, 777, 1024, 2047, 4095, 4096]:
        for k in (1, 2, 3, 5):
            out.append(rs.randint(97, 97 + k, size=n).astype(np.uint8))
        runs = bytearray()
        while len(runs) < n:
            runs += bytes([int(rs.randint(97, 100))]) * int(rs.randint(1, 12))
        out.append(np.frombuffer(bytes(runs[:n]), dtype=np.uint8).copy())
    return out


def test_edge_sizes_and_tiny_alphabets_vs_oracle(gpu, checker):
    files = _edge_inputs()
    sizes = [len(f) for f in files]
    ctx = gpu.files_context(4096, len(files))
    try:
        data = np.concatenate(files)
        offs = np.cumsum([0] + sizes[:-1])
        fo = ctx.compress_files(data, offs, sizes)
        stream = ctx.stream_read(int(fo[-1]))
        for k, f in enumerate(files):
            got = stream[int(fo[k]):int(fo[k + 1])].tobytes()
            assert got == checker.memory_compress(f, 0, 32768), (k, sizes[k])
    finally:
        ctx.close()
    # the same strings as the tail of a window with history: the end-of-window rules with earlier occurrences in reach
    hist = np.concatenate(files[-40:])[-30000:]
    for f in files[100:400:7]:
        d = np.concatenate([hist, f, f])
        check_window(gpu, checker, d, len(hist), 2 * len(f), max_block=32768, tag="edge_tail_%d" % len(f))


def _image_files(pattern, limit):
    import glob
    buf = bytearray()
    for f in sorted(glob.glob(pattern, recursive=True)):
        try:
            buf += open(f, "rb").read()
        except OSError:
            continue
        if len(buf) >= limit:
            break
    return np.frombuffer(bytes(buf[:limit]), dtype=np.uint8).copy()


@pytest.mark.parametrize("kind,flags,bs", [("pysrc", 2, 65536), ("x86", 1, 32768), ("pysrc_1m", 0, 0)])
def test_real_data_streams_vs_oracle(gpu, checker, kind, flags, bs):
    """Real text and a real executable from this image (same files on the GPU box): byte runs, long repeats, long
    barrier-free stretches — the inputs the synthetic corpora are kind to. Whole streams must equal the oracle's."""
    if kind == "x86":
        d = _image_files("/usr/bin/python3*", 2_500_000)
    else:
        d = _image_files("/usr/lib/python3*/**/*.py", 6_000_000)[2_000_000:6_000_000 if kind == "pysrc" else 4_500_000]
    if len(d) < 1_000_000:
        pytest.skip("image files not available")
    got = gpu.memory_compress(d, flags, bs)
    assert got == checker.memory_compress(d, flags, bs)


def test_full_size_properties_mixed_zlib_32k(gpu, checker):
    """BASELINE configurations 3 and 4 in spirit: zlib framing with 32 KiB max-blocks over the mixed-entropy corpus (self-test
    grid segments, noise -> stored sub-blocks, constant runs, text). Inflate round trip with zlib's own Adler-32 check, and a
    stage-by-stage spot check of a few max-blocks against the oracle."""
    d = np.concatenate([corpus.mixed(1 << 22, 11 + k) for k in range(4)])   # 16 MiB
    out = gpu.memory_compress(d, 1, 32768)
    assert out is not None
    assert zlib.decompress(out, 15) == d.tobytes()
    for blk in (1, 100, 333, 511):
        lo = blk * 32768
        check_window(gpu, checker, d[lo - 32768: lo + 32768], 32768, 32768, max_block=32768, tag="mixed_blk%d" % blk)


def test_two_mib_max_block_of_real_text_stage_by_stage(gpu, checker):
    # the largest max-block the API allows (libzultra.c:91), source code: 128 chunks of the barrier / token kernels, 33
    # matchfinder segments, the 16-wave splitter, tasks with barrier-free runs — every stage against the oracle
    d = _image_files("/usr/lib/python3*/**/*.py", 3_300_000)
    if len(d) < 3_300_000:
        pytest.skip("not enough Python sources in this image")
    check_window(gpu, checker, d[1_100_000:1_100_000 + 32768 + (2 << 20)], 32768, 2 << 20, max_block=2 << 20, tag="pysrc_2MiB")


@pytest.mark.parametrize("kind,flags,bs", [("x86", 1, 2 << 20), ("mixed", 2, 0), ("near_copies", 0, 262144), ("json", 2, 1 << 20)])
def test_large_max_blocks_streams_vs_oracle(gpu, checker, kind, flags, bs):
    """Max-blocks of 256 KiB .. 2 MiB (the reference's default is 1 MiB): many matchfinder segments and token-chain chunks
    per block, sub-blocks of a megabyte and more (deep unlimited Huffman trees in the cost estimates), the 16-wave splitter."""
    if kind == "x86":
        d = _image_files("/usr/lib/x86_64-linux-gnu/*.so*", 3_000_000)
        if len(d) < 3_000_000:
            pytest.skip("not enough shared objects in this image")
    elif kind == "mixed":
        d = corpus.mixed(3 << 20, 77)
    elif kind == "near_copies":
        d = corpus.duplicated(3_000_000, 3, 3000)
    else:
        d = corpus.json_like(3_000_000, 9)
    got = gpu.memory_compress(d, flags, bs)
    assert got == checker.memory_compress(d, flags, bs)


def test_config4_one_gib_shard(gpu, checker):
    """BASELINE configuration 4, one GPU's share: 1 GiB (16 384 max-blocks of 64 KiB) of the mixed-entropy corpus of
    tests/gen/zgen.c as ONE batch. Inflate round trip of the whole gzip stream (zlib checks the
    CRC-32 folded from the device's per-block values), and eight random max-blocks stage by stage against the oracle."""
    d = corpus.mixed_config4(0, 1024)
    n, bs = len(d), 65536
    nb = n // bs
    blocks = [(b * bs - (32768 if b else 0), 32768 if b else 0, bs) for b in range(nb)]
    ctx = gpu.context(bs, nb)
    try:
        ctx.compress_blocks(d, blocks)   # (host buffer: torch cannot initialise its own HIP runtime in a process where the library did first)
        crc = gpu.crc32_append_many(0, ctx.block_crc32(), np.full(nb, bs, dtype=np.uint32))
        end_bit, _ = ctx.stitch_device(nb - 1, phase=0)
        body = ctx.stream_read((end_bit + 7) // 8).tobytes()
        st = ctx.stats()
        assert st["blocks"] == nb and st["positions"] == n
    finally:
        ctx.close()
    gz = bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 2, 255]) + body + int(crc).to_bytes(4, "little") + int(n & 0xffffffff).to_bytes(4, "little")
    dec = zlib.decompressobj(31)
    pos = 0
    for off in range(0, len(gz), 16 << 20):
        out = dec.decompress(gz[off:off + (16 << 20)])
        assert out == d[pos:pos + len(out)].tobytes(), "inflated bytes differ near %d" % pos
        pos += len(out)
    assert dec.eof and pos == n
    rs = np.random.RandomState(4)
    for blk in sorted(int(x) for x in rs.randint(1, nb, size=8)):
        lo = blk * bs
        check_window(gpu, checker, d[lo - 32768: lo + bs], 32768, bs, max_block=bs, tag="config4_blk%d" % blk)


def test_files_context_rejects_inputs_above_its_declared_size(gpu):
    """A files context never splits an input; from 8192 bytes on the reference's splitter may (blockdeflate.c:646), so an
    input larger than the size the context was created for must be refused, not compressed differently."""
    import zultra_amd
    ctx = gpu.files_context(8191, 4)
    try:
        d = np.concatenate([corpus.text_like(4000, 3), corpus.noise(4192, 5)])
        with pytest.raises(zultra_amd.ZultraError):
            ctx.compress_files(d, [0], [8192])
        fo = ctx.compress_files(d, [0], [8191])
        assert fo[-1] > 0
    finally:
        ctx.close()


def test_two_threads_compress_different_streams(gpu, checker):
    """The reference is re-entrant (no globals): two host threads with a stream each must not disturb one another."""
    import threading
    inputs = [np.concatenate([corpus.text_like(400000, 41), corpus.noise(70000, 6)]), corpus.mixed(450000, 42)]
    params = [(2, 65536), (1, 32768)]
    want = [checker.memory_compress(d, f, b) for d, (f, b) in zip(inputs, params)]
    got = [[None] * 3, [None] * 3]

    def work(i):
        for rep in range(3):
            got[i][rep] = gpu.memory_compress(inputs[i], params[i][0], params[i][1])

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i in range(2):
        for rep in range(3):
            assert got[i][rep] == want[i], (i, rep)


@pytest.mark.parametrize("runs", ["1", "2", "6"])
def test_number_of_staggered_runs_does_not_change_the_bytes(gpu, monkeypatch, runs):
    """zh_device.hip cuts a batch into ZULTRA_HIP_STREAMS staggered runs (default 2), each with its own block of device counters
    (ZH_CNT_*): any number of runs (default: three, four for batches of 256 MiB and more) yields the same stream. 24 MiB of table-like and repetitive text: chains and cut tasks in
    every run."""
    data = np.concatenate([corpus.table_like(8 << 20, 3), corpus.duplicated(8 << 20, 4, 900), corpus.text_like_fast(8 << 20, 5)])
    monkeypatch.setenv("ZULTRA_HIP_CACHE", "0")   # a fresh context per call: the number of runs is fixed when it is created
    gpu.L.zultra_release_cached_contexts()
    want = gpu.memory_compress(data, 2, 65536)
    assert zlib.decompress(want, 31) == data.tobytes()
    monkeypatch.setenv("ZULTRA_HIP_STREAMS", runs)
    assert gpu.memory_compress(data, 2, 65536) == want


def test_bench_refuses_more_gpus_than_present(gpu):
    """`bench.py --gpus N` must never report a smaller job as N: on a box with fewer GPUs it fails loudly."""
    import os
    import subprocess
    import sys

    n = gpu.device_count()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n + 1), "--steps", "1", "--warmup", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode != 0
    assert b"n_gpus" not in p.stdout
    assert b"GPU(s) visible" in p.stderr
    # a launcher whose WORLD_SIZE disagrees with --gpus is an error too
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode != 0 and b"WORLD_SIZE" in p.stderr


def test_bench_config1_known_answer():
    """BASELINE configuration 1 through bench.py: the image's bootstrap.min.js -> 10 523 bytes of raw deflate (golden from the
    compiled reference), on the GPU."""
    import json
    import os
    import subprocess
    import sys
    if not os.path.exists(corpus.BOOTSTRAP_JS):
        pytest.skip("bootstrap.min.js is not in this image")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert line["known_answer_ok"] and line["compressed_bytes"] == 10523 and line["gzip_bytes"] == 10541
    assert "roofline" in line and "cpu_baseline" in line


def test_cached_contexts_are_released_on_request(gpu):
    """zultra_stream_end keeps the device context of a finished stream for the next one; zultra_release_cached_contexts()
    frees what is kept (the reference frees everything in zultra_stream_end, libzultra.c:521-565)."""
    import ctypes as C
    f = gpu.L.zultra_release_cached_contexts
    f.restype = C.c_int
    f()
    d = corpus.text_like(200000, 3)
    assert gpu.memory_compress(d, 2, 65536) is not None
    assert f() == 1          # the stream's context was cached ...
    assert f() == 0          # ... and is gone now
    g = gpu.L.zultra_hip_context_bytes
    g.argtypes = [C.c_uint32, C.c_uint32]
    g.restype = C.c_size_t
    ctx = gpu.context(65536, 64)
    try:
        info = (C.c_int(), C.c_uint32(), C.c_uint32(), C.c_size_t())
        gpu.L.zultra_hip_ctx_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_size_t)]
        gpu.L.zultra_hip_ctx_info(ctx.h, *[C.byref(x) for x in info])
        assert (info[1].value, info[2].value) == (65536, 64)
        est = g(65536, 64)
        assert 0.95 * info[3].value <= est <= 1.10 * info[3].value, (est, info[3].value)   # the estimate follows the real layout
    finally:
        ctx.close()


def test_api_state_machine_on_the_device(gpu, checker):
    """The stream API's corner conventions, on the GPU build (the emulator suite checks the same): ZULTRA_STREAM_END once, any
    further call -5 (libzultra.c:204-205,504-507); zultra_stream_set_dictionary only before the first compress (libzultra.c:180);
    zultra_memory_bound equal to the reference's formula (libzultra.c:576-587)."""
    t = corpus.text_like(5000, 2)
    s = gpu.stream(2, 0)
    st, out = s.compress(t, True)
    assert st == 1 and out == checker.memory_compress(t, 2, 0)          # ZULTRA_STREAM_END with the whole stream delivered
    st2, out2 = s.compress(t[:0], True)
    assert st2 == -5 and out2 == b""                                    # ZULTRA_ERROR_COMPRESSION from then on
    s.end()
    s = gpu.stream(0, 0)
    s.compress(t, False)
    assert s.set_dictionary(t) == -5
    s.end()
    for n in (0, 1, 65535, 65536, 10 ** 6, (1 << 31) + 5):
        for flags in (0, 1, 2):
            for bs in (0, 32768, 65536, 1 << 22):
                assert gpu.memory_bound(n, flags, bs) == checker.memory_bound(n, flags, bs)


@pytest.mark.parametrize("how", ["env_0_0", "api_0_0_0", "one_lane_two_lanes_stream_api_large_input"])
def test_memory_compress_over_device_lanes(gpu, checker, monkeypatch, how):
    """zultra_memory_compress over several device contexts (libzultra.cpp: lanes; ZULTRA_HIP_DEVICES / zultra_set_devices): shards of
    max-blocks compressed side by side by one host thread and one context each — here all on device 0 — and stitched in stream
    order at the bit phase the stream has reached, with a stored sub-block right behind every cut. The bytes are those of the
    one-stream path (and of the oracle)."""
    import ctypes as C
    bs = 65536
    if how == "one_lane_two_lanes_stream_api_large_input":
        d = corpus.real_text(56 << 20)
        d[28 << 20:(28 << 20) + 40000] = corpus.noise(40000, 11)
        monkeypatch.setenv("ZULTRA_HIP_MEMORY_LANES", "0")              # through the stream API, as the reference does
        want = gpu.memory_compress(d, 2, bs)
        monkeypatch.setenv("ZULTRA_HIP_MEMORY_LANES", "2")              # two contexts on the device
        got2 = gpu.memory_compress(d, 2, bs)
        monkeypatch.delenv("ZULTRA_HIP_MEMORY_LANES")
        got = gpu.memory_compress(d, 2, bs)                             # default: one lane, staged and uploaded run by run
        assert got == want and got2 == want and zlib.decompress(got, 31) == d.tobytes()
        return
    d = corpus.text_like_fast(24 * bs + 1234, 77)
    lanes = 2 if how == "env_0_0" else 3
    for k in range(1, lanes):
        cut = ((24 * k + lanes - 1) // lanes) * bs if lanes == 3 else 12 * bs
        d[cut:cut + 30000] = corpus.noise(30000, cut)
    want = checker.memory_compress(d, 1, bs)
    if how == "env_0_0":
        monkeypatch.setenv("ZULTRA_HIP_DEVICES", "0,0")
        got = gpu.memory_compress(d, 1, bs)
    else:
        f = gpu.L.zultra_set_devices
        f.argtypes = [C.POINTER(C.c_int), C.c_int]
        f.restype = C.c_int
        assert f((C.c_int * 1)(gpu.device_count()), 1) == -1            # not a visible device
        assert f((C.c_int * 3)(0, 0, 0), 3) == 3
        try:
            got = gpu.memory_compress(d, 1, bs)
        finally:
            assert f(None, 0) == 0
    assert got == want
    assert zlib.decompress(got, 15) == d.tobytes()


def test_streams_get_hardware_queues_of_their_own(tmp_path):
    """A process that initialised HIP with few hardware queues (GPU_MAX_HW_QUEUES=2 here; the runtime's default of four when an application touches HIP before the
    library is loaded) puts several of a context's streams on one queue, where they run one after the other. The context finds out at creation — its streams spin
    side by side for 0.2 ms and it looks at who ran when (zh_spread_streams) — and replaces the ones that shared; the bytes are the checker's either way."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    worker = r'''
import os, sys
import numpy as np
import torch
torch.zeros(1).cuda()                      # HIP is up before the library is loaded: its GPU_MAX_HW_QUEUES hint comes too late
sys.path.insert(0, %(root)r); sys.path.insert(0, %(here)r)
import corpus, zlibs, zultra_amd
d = np.concatenate([corpus.text_like_fast(9 << 20, 3), corpus.indented(4 << 20, 5)])
got = zultra_amd.lib().memory_compress(d, 2, 65536)
want = (zlibs.RefStages() if zlibs.have_ref() else zlibs.Oracle()).memory_compress(d, 2, 65536)
assert got == want
print("bytes equal")
''' % {"root": root, "here": here}
    script = tmp_path / "queues.py"
    script.write_text(worker)
    env = dict(os.environ, GPU_MAX_HW_QUEUES="2", ZULTRA_HIP_SPREAD_STREAMS="2")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "bytes equal" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
    import re
    m = re.findall(r"zultra_amd: (\d+) of (\d+) streams replaced", r.stderr)
    assert m and any(int(a) > 0 for a, _ in m), r.stderr[-1500:]   # with two queues for eight streams some must have shared


def test_sharded_assembly_at_world_two_with_the_real_kernels(tmp_path):
    """zultra_amd.sharded.assemble — the N > 1 path of bench.py — with the REAL kernels: two rank processes, both on GPU 0, the
    collectives over gloo (RCCL refuses two ranks on one device). A stored sub-block sits right behind the cut; rank 0's stream
    equals the oracle's single stream."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    worker = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r); sys.path.insert(0, %(here)r)
import corpus, zlibs, zultra_amd
from zultra_amd import sharded
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
L = zultra_amd.lib()
bs = 65536
data = corpus.text_like_fast(16 * bs + 7000, 5)
data[8 * bs:8 * bs + 30000] = corpus.noise(30000, 8)          # stored sub-block right behind the cut between the ranks
n = len(data)
nb = (n + bs - 1) // bs
lo, hi = sharded.shard_range(nb, rank, world)
first = lo * bs - (32768 if lo else 0)
blocks = [(b * bs - (32768 if b else 0) - first, 32768 if b else 0, min(bs, n - b * bs)) for b in range(lo, hi)]
ctx = L.context(bs, hi - lo, device=0)
ctx.compress_blocks(data[first:min(n, hi * bs)], blocks)
stream, info = sharded.assemble(L, ctx, bs, dist, torch, torch.device("cpu"), (nb - 1 - lo) if hi == nb else -1)
if rank == 0:
    want = (zlibs.RefStages() if zlibs.have_ref() else zlibs.Oracle()).memory_compress(data, 0, bs)   # the compiled reference where it travelled
    assert stream.tobytes() == want, (len(stream), len(want))
    print("SHARDED_GPU_OK", len(want), info["start_phase"])
dist.destroy_process_group()
'''
    script = tmp_path / "worker.py"
    script.write_text(worker % {"root": root, "here": here})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=900)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "SHARDED_GPU_OK" in outs[0], outs[0]


def test_sharded_assembly_under_a_one_rank_rccl_group(tmp_path):
    """The collectives of the N > 1 path (zultra_amd.sharded.assemble: all_gather of the phase tables, gather of the first bytes, the
    slice views of rank 0's stream buffer) on DEVICE tensors over RCCL — a process group of one rank with backend "nccl", which is all a
    one-GPU box can host (RCCL refuses two ranks on one device). force_collectives takes the world of one through the same calls as
    N > 1. The stream equals the oracle's; a stored sub-block sits in the middle."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    worker = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r); sys.path.insert(0, %(here)r)
import corpus, zlibs, zultra_amd
from zultra_amd import sharded
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
one = torch.ones(1, dtype=torch.int64, device=dev)
dist.all_reduce(one)
assert int(one.item()) == 1
L = zultra_amd.lib()
bs = 65536
data = corpus.text_like_fast(12 * bs + 5000, 5)
data[6 * bs:6 * bs + 30000] = corpus.noise(30000, 8)
n = len(data)
nb = (n + bs - 1) // bs
blocks = [(b * bs - (32768 if b else 0), 32768 if b else 0, min(bs, n - b * bs)) for b in range(nb)]
d_data = torch.from_numpy(data).to(dev)
ctx = L.context(bs, nb, device=0)
ctx.compress_blocks(d_data.data_ptr(), blocks, data_on_device=True, data_size=d_data.numel())
stream, info = sharded.assemble(L, ctx, bs, dist, torch, dev, nb - 1, extra=np.array([7, n], dtype=np.int64), force_collectives=True)
want = (zlibs.RefStages() if zlibs.have_ref() else zlibs.Oracle()).memory_compress(data, 0, bs)   # the compiled reference where it travelled
assert stream.tobytes() == want, (len(stream), len(want))
assert [int(x) for x in info["extras"][0]] == [7, n] and info["start_phase"] == 0 and "collective_ms" in info
print("RCCL_ONE_RANK_OK", len(want), round(info["collective_ms"], 3))
dist.destroy_process_group()
'''
    script = tmp_path / "worker.py"
    script.write_text(worker % {"root": root, "here": here})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29543", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = p.stdout.decode()
    assert p.returncode == 0 and "RCCL_ONE_RANK_OK" in out, out


def test_reference_selftest_grid_sampled(gpu, checker):
    """A sample of the grid the reference tool's self-test walks (tool/zultra.c:529-534: twelve alphabet sizes x match probabilities
    0 .. 0.995 x sizes 16 384 .. 131 072; the full grid is test_reference_cli_full_selftest_on_device, 13 minutes): every alphabet x
    three probabilities x the two end sizes, same construction of the data (tests/corpus.py: selftest_data), each case through
    zultra_memory_compress of the device library — the bytes are the oracle's, and zlib inflates them back."""
    n_cases = 0
    for a_i, alphabet in enumerate((1, 2, 3, 15, 30, 56, 96, 137, 178, 191, 255, 256)):
        for p_i, prob in enumerate((0.0, 0.5, 0.995)):
            for size in (16384, 131072):
                d = corpus.selftest_data(size, 1000 + 37 * a_i + 7 * p_i + (size >> 14), alphabet, prob)
                got = gpu.memory_compress(d, 2, 0)
                want = checker.memory_compress(d, 2, 0)
                assert got is not None and got == want, "alphabet %d probability %.3f size %d: %d bytes, oracle %d" % (alphabet, prob, size, len(got or b""), len(want))
                assert zlib.decompress(got, 31) == d.tobytes()
                n_cases += 1
    assert n_cases == 72


def test_many_sub_blocks_in_one_max_block(gpu, checker):
    """The splitter's cap (blockdeflate.c:643-647: 63 interior splits, depth 6, 8192 bytes): a 2 MiB max-block of 64 stretches of
    32 KiB, each over 16 byte values of its own that fall into one bin of the splitter's 18-bin statistics (blockdeflate.c:684-703:
    literal bin = ((b >> 4) & 0xc) | (b & 3)), neighbours in different bins — the splitter cuts it into 41 sub-blocks; splits,
    costs, parse and bits equal the oracle's."""
    parts = []
    for k in range(64):
        r = corpus.noise(32768, 500 + k)
        parts.append(((((k >> 2) & 3) << 6) | ((r & 15) << 2) | (k & 3)).astype(np.uint8))
    d = np.concatenate(parts)
    st = {}
    check_window(gpu, checker, d, 0, len(d), max_block=2 << 20, tag="many_splits", stats_out=st)
    assert st["subblocks"] >= 32, st


def test_stream_memory_comes_from_the_callers_allocator(gpu, checker):
    """libzultra.h:88-90 / libzultra.c:59-71,94-147: a stream's own memory goes through the caller's zalloc / zfree — the
    compressor state and its per-max-block arrays; the device context (device memory, pinned staging) belongs to the backend.
    A counting allocator sees every one of these allocations freed by zultra_stream_end, and none after it."""
    import ctypes as C
    from zultra_amd._ffi import ZALLOC_T, ZFREE_T
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]
    libc.free.argtypes = [C.c_void_p]
    live, log = {}, []

    def za(opaque, items, size):
        p = libc.malloc(items * size)
        live[p] = items * size
        log.append(items * size)
        return p

    def zf(opaque, p):
        assert p in live, "zfree of memory zalloc never returned"
        del live[p]
        libc.free(p)

    zalloc, zfree = ZALLOC_T(za), ZFREE_T(zf)
    d = corpus.text_like(50000, 6)
    s = gpu.stream(2, 32768, zalloc, zfree)
    assert len(log) >= 5 and len(live) == len(log)      # the state and four per-max-block arrays
    n_init = len(log)
    st, out = s.compress(d, True)
    assert st == 1 and out == checker.memory_compress(d, 2, 32768)
    assert len(log) == n_init                           # compressing allocates nothing more on the host side of the stream
    s.end()
    assert not live
