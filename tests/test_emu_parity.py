"""CPU (no GPU needed): the product's own kernel sources, compiled against the lock-step wave emulator in
tests/emu/, checked stage by stage against the oracle and on whole streams against the golden vectors.
This exercises the kernel logic and the host layer (stream state machine, batching, stitcher); the `-m gpu`
tests repeat the same checks on the real MI355X build."""
import os
import sys
import zlib

import numpy as np
import pytest

import corpus
import golden_util as G
from parity_util import check_window

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))


@pytest.fixture(scope="module")
def emu():
    import build_emu
    from zultra_amd._ffi import Lib
    return Lib(build_emu.build())


def test_emu_is_not_the_product_library(emu):
    import zultra_amd
    assert os.path.realpath(emu.path) != os.path.realpath(zultra_amd.LIB_PATH)
    assert "tests/emu/_build" in emu.path


@pytest.mark.parametrize("case", [
    ("text", lambda: corpus.text_like(9000, 3), 1000, 8000),
    ("text_split", lambda: corpus.text_like(60000, 21)[30000:42000], 0, 12000),
    ("selftest", lambda: corpus.selftest_data(6000, 77, 15, 0.5), 1000, 5000),
    ("binary2", lambda: corpus.selftest_data(5000, 5, 2, 0.3), 0, 5000),
    ("zeros", lambda: corpus.constant(3000), 500, 2500),
    ("period3", lambda: corpus.periodic(2500, 3), 0, 2500),
    ("noise", lambda: corpus.noise(3000, 1), 0, 3000),
    ("tiny", lambda: corpus.text_like(10, 1), 0, 10),
    ("three", lambda: corpus.text_like(3, 1), 0, 3),
    ("one", lambda: corpus.text_like(1, 1), 0, 1),
    ("end_clamp", lambda: np.concatenate([corpus.noise(300, 2), corpus.constant(700, 65)]), 0, 1000),
    ("byte_runs", lambda: corpus.indented(7000, 11), 2000, 5000),
    ("byte_runs_first", lambda: corpus.indented(3000, 12), 0, 3000),
    ("near_copies", lambda: corpus.duplicated(14000, 3, 1500), 2000, 12000),
    ("deep_huffman", lambda: corpus.fibonacci_bytes(19), 0, 10945),
], ids=lambda c: c[0])
def test_stages_vs_oracle(emu, oracle, case):
    name, gen, prev, n = case
    check_window(emu, oracle, gen(), prev, n, tag=name)


@pytest.mark.parametrize("case", [
    ("text", lambda: corpus.text_like(9000, 3), 1000, 8000),
    ("byte_runs", lambda: corpus.indented(7000, 11), 2000, 5000),
    ("near_copies", lambda: corpus.duplicated(14000, 3, 1500), 2000, 12000),
    ("deep_huffman", lambda: corpus.fibonacci_bytes(19), 0, 10945),
], ids=lambda c: c[0])
def test_long_pieces_stay_on_the_quads(emu, oracle, monkeypatch, case):
    """A run of fewer tasks than CUs lists every task with a barrier-free piece of more than 256 positions for the chain kernel
    (ZULTRA_HIP_COOP_SMALL: one call on a few max-blocks waits for its longest chain of steps); large batches keep pieces of up to
    ZH_COOP_MIN = 1536 positions on the quads of zh_parse_lanes. The same windows with the large-batch bound."""
    name, gen, prev, n = case
    monkeypatch.setenv("ZULTRA_HIP_COOP_SMALL", "1536")
    check_window(emu, oracle, gen(), prev, n, tag=name + "/coop1536")


@pytest.mark.parametrize("case", [
    ("text", "256", lambda: corpus.text_like(9000, 3), 1000, 8000),
    ("binary2", "256", lambda: corpus.selftest_data(5000, 5, 2, 0.3), 0, 5000),
    ("byte_runs", "256", lambda: corpus.indented(7000, 11), 2000, 5000),
    ("zeros", "16", lambda: corpus.constant(3000), 500, 2500),
    ("end_clamp", "16", lambda: np.concatenate([corpus.noise(300, 2), corpus.constant(700, 65)]), 0, 1000),
], ids=lambda c: c[0] + "/cap" + c[1])
def test_matchfinder_chunks_and_oversized_classes(emu, oracle, monkeypatch, case):
    """zh_mf_group_lds.h with chunks of 256 elements: a window of a few KB is then several chunks of the bigram order, refined in LDS one
    after the other, and every bigram class above 256 entries (byte runs, the four classes of a two-symbol alphabet) goes through the
    passes in HBM on its range; with 16, nearly every class does. (By default such windows are one chunk: the whole-window path.)"""
    name, cap, gen, prev, n = case
    monkeypatch.setenv("ZULTRA_HIP_MF_CAP", cap)
    check_window(emu, oracle, gen(), prev, n, tag=name + "/cap" + cap)


@pytest.mark.parametrize("case", [
    ("zeros", lambda: corpus.constant(3000), 500, 2500, 3),
    ("noise", lambda: corpus.noise(3000, 1), 0, 3000, 2),
    ("selftest", lambda: corpus.selftest_data(6000, 77, 15, 0.5), 1000, 5000, 1),
    ("text", lambda: corpus.text_like(9000, 3), 1000, 8000, 0),
], ids=lambda c: c[0])
def test_settled_subblocks_keep_their_parse(emu, oracle, case):
    """A sub-block whose rebuilt code lengths (unused symbols at 9 / 6 bits) equal the ones its parse pass priced with has reached a
    fixed point of the reference's loop (blockdeflate.c:874-901): the remaining passes would reproduce the parse, so the parse
    kernels skip it (zh_sb_build_one, st->settled) and only the code builds run. One max-block each that settles after pass 0, 1,
    2 and never: the stages still equal the oracle's, and the number of passes not run is what the fixed point implies."""
    name, gen, prev, n, skipped = case
    st = {}
    check_window(emu, oracle, gen(), prev, n, tag=name, stats_out=st)
    assert st["subblocks"] == 1 and st["settled_passes"] == skipped, st


@pytest.mark.parametrize("wide", ["1", "1000000", "whole"], ids=["as_waves_of_zh_parse_segments", "as_jobs_of_zh_parse_chain", "few_and_short_stay_whole"])
def test_chain_segments_are_both_accepted_and_redone(emu, oracle, monkeypatch, wide):
    """The speculative segments of long barrier-free tasks (zh_parse.h; the emulator build cuts every 512 positions with a
    288-position warm-up), parsed either way the host may choose (ZULTRA_HIP_SEG_WIDE = the number of segments in a run from
    which they go to zh_parse_segments): on table-like data some cuts verify and some segments are parsed again, and the parse
    is the reference's either way."""
    monkeypatch.setenv("ZULTRA_HIP_SEG_WIDE", "1000000" if wide == "whole" else wide)
    monkeypatch.setenv("ZULTRA_HIP_SEG_WHOLE", "1000000" if wide == "whole" else "0")
    st = {}
    check_window(emu, oracle, corpus.table_like(26000, 9), 0, 26000, tag="table_cut/" + wide, stats_out=st)
    cuts = st["cut_segments"] - st["cut_tasks"]
    assert st["cut_tasks"] >= 1 and cuts >= 8
    if wide == "whole":
        assert st["cut_redone"] == 0, st
    else:
        assert 0 < st["cut_redone"] < 4 * cuts, st


@pytest.mark.parametrize("demote", ["0", "1"])
def test_cut_tasks_that_keep_failing_become_whole_chains(emu, oracle, monkeypatch, demote):
    """A cut task with ZULTRA_HIP_DEMOTE (default 2) or more failed cuts in one pass is handed to zh_parse_chain as one chain for the passes
    left (its checker parses failed segments again one after the other, on one row); zh_parse_segments skips it from then on. With 1 every
    task with a failed cut goes, with 0 none: the stages are the oracle's each time."""
    monkeypatch.setenv("ZULTRA_HIP_SEG_WIDE", "1")
    monkeypatch.setenv("ZULTRA_HIP_SEG_WHOLE", "0")
    monkeypatch.setenv("ZULTRA_HIP_DEMOTE", demote)
    st = {}
    check_window(emu, oracle, corpus.table_like(26000, 9), 0, 26000, tag="table_cut/demote" + demote, stats_out=st)
    assert st["cut_tasks"] >= 1 and st["cut_redone"] > 0, st
    if demote == "0":
        assert st["cut_demoted"] == 0, st
    elif demote == "1":
        assert 1 <= st["cut_demoted"] <= st["cut_tasks"], st


@pytest.mark.parametrize("block", [65536])
def test_chain_lists_of_the_three_length_classes_do_not_run_into_each_other(emu, oracle, block):
    """zh_list_huge files whole chain tasks in three length classes. Records of 1800 zeros + 8 random bytes make most task
    slots of the run short chains (barrier-free runs of ~1800, periodic: never cut into segments) and one record of 7000 zeros
    a long one: with the classes packed into one list of `cap` entries the long and the short class overlapped once more than
    half the slots were listed — a task was then parsed twice and another never (round 2: 65 546 stored bytes instead of 476, and
    at 32 KiB a stream that inflated to other bytes)."""
    rng = np.random.default_rng(7)
    recs = []
    n = 0
    k = 0
    while n < 65536:
        z = 7000 if k == 9 else 1800
        recs += [np.zeros(z, dtype=np.uint8), rng.integers(1, 256, 8, dtype=np.uint8)]
        n += z + 8
        k += 1
    d = np.concatenate(recs)[:65536]
    got = emu.memory_compress(d, 2, block)
    want = oracle.memory_compress(d, 2, block)
    assert got is not None and zlib.decompress(got, 31) == d.tobytes()
    assert got == want, "%d bytes, oracle %d" % (len(got), len(want))


def test_memory_compress_over_two_device_lanes(emu, oracle, monkeypatch):
    """ZULTRA_HIP_DEVICES=0,0: zultra_memory_compress cuts the input into two shards of max-blocks, a host thread and a device
    context each (here: on the emulator's one device, the kernels taking turns), and stitches the shards in stream order at the
    bit phase the stream has reached — a stored sub-block sits right behind the cut. Same bytes as the one-stream path."""
    d = corpus.text_like(2 * 32768 + 300, 9)           # three max-blocks: two for the first lane, one for the second
    d[32768:32768 + 12000] = corpus.noise(12000, 4)
    d[2 * 32768:] = corpus.noise(300, 5)
    want = oracle.memory_compress(d, 2, 32768)
    monkeypatch.setenv("ZULTRA_HIP_DEVICES", "0,0")
    got = emu.memory_compress(d, 2, 32768)
    assert got == want


@pytest.mark.parametrize("name", ["tiny_100", "one_byte", "two_bytes", "json_4k", "json_4k_b"])
def test_golden_streams(emu, name):
    c = G.stream_case(name)
    G.check_stream_output(c, emu.memory_compress(c["data"], c["flags"], c["max_block"], c["dictionary"]))


def test_multiblock_stream_bit_carry_and_stored_fallback(emu, oracle):
    # 2 max-blocks of 32 KiB: text | short noise tail (stored, padding depends on the bit phase the text left), gzip framing
    # (the three-block text | stored | text case runs on the GPU: test_gpu_parity.py, __graft_entry__.smoke)
    t = corpus.text_like(40000, 5)
    d = np.concatenate([t[:32768], corpus.noise(3000, 3)])
    got = emu.memory_compress(d, 2, 32768)
    assert got == oracle.memory_compress(d, 2, 32768)
    assert zlib.decompress(got, 31) == d.tobytes()


def _scan_vs_planner(lib, more=True, bs=32768):
    """The device scan of the stitcher (zh_stitch_scan: transfer tables bit phase -> bits added, composed over the batch) against the serial
    host planner (zh_stitch_plan, behind zultra_hip_stitch), at every one of the eight start phases: same bytes, same end bit, same new phase;
    and the eight-entry phase table a rank hands its neighbours (zultra_hip_stitch_phase_table) = the eight end bits."""
    import ctypes as C
    t = corpus.text_like(3 * bs, 11)
    d = t.copy()
    d[bs + 100: bs + 9000] = corpus.noise(8900, 1)        # a stored sub-block inside the second max-block
    d[2 * bs: 2 * bs + 700] = corpus.noise(700, 2)        # ... at the head of the third
    d = np.concatenate([d, corpus.noise(333, 3)])         # a short stored max-block at the end
    nb = (len(d) + bs - 1) // bs
    blocks = [(b * bs - (bs if b else 0), bs if b else 0, min(bs, len(d) - b * bs)) for b in range(nb)]
    ctx = lib.context(bs, nb)
    try:
        # (round 6) the stitch at phase 0 goes out WITH the batch (zultra_hip_stitch_with_batch): the first stitch_device call of the loop below returns its
        # result without a launch, the seven others — other phases — stitch again
        ctx.stitch_with_batch(nb - 1, phase=0)
        ctx.compress_blocks(d, blocks)
        subs, _, cnt = ctx.subblocks()
        assert cnt > nb   # the splitter cut at least one max-block
        offs = [b * bs for b in range(nb)]
        ends = (C.c_uint64 * 8)()
        failed = C.c_uint32(7)
        lib.L.zultra_hip_stitch_phase_table.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
        assert lib.L.zultra_hip_stitch_phase_table(ctx.h, ends, C.byref(failed)) == 0 and failed.value == 0
        stored = set()
        for ph in range(8):
            from zultra_amd._ffi import BitState
            want, st = ctx.stitch(d, offs, bs, nb - 1, state=BitState(0, ph), finish=False)
            end_bit, nacc = ctx.stitch_device(nb - 1, phase=ph)
            assert nacc == st.nacc and (end_bit + 7) // 8 == len(want) + (1 if st.nacc else 0), (ph, end_bit, len(want), st.nacc)
            got = ctx.stream_read((end_bit + 7) // 8)
            full = len(want)
            assert got[:full].tobytes() == want[:full], "phase %d" % ph
            if st.nacc:
                assert int(got[full]) == st.acc & 0xff, "phase %d partial byte" % ph
            assert int(ends[ph]) == end_bit, (ph, int(ends[ph]), end_bit)
            stored.add(end_bit - ph)
        assert len(stored) > 1   # the shard's bit length does depend on the phase it starts at (stored sub-blocks pad to a byte)
        # a stitch that goes out WITH its batch (zultra_hip_stitch_with_batch, round 6): armed for one batch, its result returned by the stitch_device call
        # that follows with the same arguments — the same bytes as a stitch of its own; other arguments stitch again
        for ph in ((0, 5) if more else ()):   # (the emulator suite stops here: a batch takes it a minute)
            from zultra_amd._ffi import BitState
            want, st = ctx.stitch(d, offs, bs, nb - 1, state=BitState(0, ph), finish=False)
            ctx.stitch_with_batch(nb - 1, phase=ph)
            ctx.compress_blocks(d, blocks)
            assert ctx.timing()["stitch_ms"] > 0 or "emu" in lib.path   # (the batch's own timing carries the stitch)
            end_bit, nacc = ctx.stitch_device(nb - 1, phase=ph)
            got = ctx.stream_read((end_bit + 7) // 8)
            assert nacc == st.nacc and got[:len(want)].tobytes() == want, "with the batch, phase %d" % ph
            other = (ph + 3) & 7
            want2, st2 = ctx.stitch(d, offs, bs, nb - 1, state=BitState(0, other), finish=False)
            end2, nacc2 = ctx.stitch_device(nb - 1, phase=other)
            assert nacc2 == st2.nacc and ctx.stream_read((end2 + 7) // 8)[:len(want2)].tobytes() == want2, "restitched at phase %d" % other
            ctx.compress_blocks(d, blocks)   # (not armed any more: a stitch of its own)
            end3, _ = ctx.stitch_device(nb - 1, phase=ph)
            assert end3 == end_bit and ctx.stream_read((end3 + 7) // 8)[:len(want)].tobytes() == want
    finally:
        ctx.close()


def test_device_scan_equals_the_host_planner_at_every_phase(emu):
    _scan_vs_planner(emu, more=False, bs=16384)   # (half the bytes of the GPU form: a third of the emulator suite's longest test)


def _overflow_forms(lib, checker, monkeypatch, sizes):
    """The <true> forms of zh_sb_init / zh_sb_build / zh_list_huge / zh_post_tasks / zh_emit_tasks — a few workgroups that stride over whatever lies beyond
    the <false> grid — are reached by a run with more sub-blocks than max(4 x max-blocks, 1024) or more tasks than bytes / 2048 + 4 x max-blocks only.
    ZULTRA_HIP_GRID_CAP caps the <false> grids (here at 1 and at 3 workgroups), so that nearly every sub-block and task of a batch that splits into many
    sub-blocks goes through the strided forms: stages and streams are the checker's."""
    from parity_util import check_window
    stretch, nstretch, caps, streams = sizes
    parts = []
    for k in range(nstretch):   # stretches over byte values of their own, neighbours in different bins of the splitter's statistics: many sub-blocks
        r = corpus.noise(stretch, 500 + k)
        parts.append(((((k >> 2) & 3) << 6) | ((r & 15) << 2) | (k & 3)).astype(np.uint8))
    many = np.concatenate(parts)
    d = np.concatenate([corpus.text_like(32768, 3), many, corpus.text_like(20000, 4), corpus.noise(3000, 1)])
    for cap in caps:
        monkeypatch.setenv("ZULTRA_HIP_GRID_CAP", cap)
        st = {}
        check_window(lib, checker, many, 0, len(many), max_block=1 << 20, tag="overflow_forms/cap" + cap, stats_out=st)
        assert st["subblocks"] >= 3 and st["tasks"] > int(cap) + 2, st
        for flags, bs in streams:
            assert lib.memory_compress(d, flags, bs) == checker.memory_compress(d, flags, bs), (cap, flags, bs)


def _chains_after_none(lib, checker, n, bs, order="pc"):
    """A run whose counterpart in the context's last batch listed nothing for zh_parse_chain is enqueued without chain kernels (round 6: a stream without chains must not
    wait, in every pass, for a grid of four-wave workgroups to be scheduled and leave). When such a run lists chains after all its kernels leave at once and the host runs
    the batch again, with them (zh_run_is_void). `order`: the batches of one context, p = text without chains, c = tables and near-copies with whole and cut chain tasks.
    Every batch's bytes are the checker's; the stats say which batches ran without chain kernels and which were run again."""
    plain = corpus.text_like(n, 31)
    chains = np.concatenate([corpus.table_like(n // 2, 9), corpus.duplicated(n - n // 2, 4, 900)])
    nb = (n + bs - 1) // bs
    blocks = [(b * bs - (32768 if b else 0), 32768 if b else 0, min(bs, n - b * bs)) for b in range(nb)]
    ctx = lib.context(bs, nb)
    try:
        prev_kind, reruns = None, 0
        for k, kind in enumerate(order):
            d = plain if kind == "p" else chains
            ctx.compress_blocks(d, blocks)
            st = ctx.stats()
            end_bit, _ = ctx.stitch_device(nb - 1, phase=0)
            got = ctx.stream_read((end_bit + 7) // 8).tobytes()
            assert got == checker.memory_compress(d, 0, bs), (k, order)
            listed = st["huge_tasks"] + st["cut_tasks"]
            assert (listed == 0) == (kind == "p"), (k, order, st)
            # behind a batch without chains: no chain kernels — unless this batch has chains, then it was run again with them; behind one with chains (and for the
            # first batch of a context): always with them
            if prev_kind == "p" and kind == "c":
                reruns += 1
            assert st["batches_rerun"] == reruns, (k, order, st)
            # (run by run: a batch with chains may hold a run without any — that run goes without its chain kernels next time, and in a rerun)
            assert (st["runs_without_chain_kernels"] == st["runs"]) == (prev_kind == "p" and kind == "p"), (k, order, st)
            assert k > 0 or st["runs_without_chain_kernels"] == 0, (k, order, st)
            prev_kind = kind
    finally:
        ctx.close()


def test_chains_turn_up_after_a_batch_without_any(emu, oracle):
    _chains_after_none(emu, oracle, 24000, 32768, "pc")   # (the emulator takes a minute for the two batches)


def _outgrown_grids(lib, checker, monkeypatch, cap, small, big, bs):
    """The <true> overflow forms of the per-sub-block / per-task kernels are not launched for a run whose counterpart in the context's last batch stayed inside the
    <false> grids (round 6); the grids go into the run's counters, and a run that outgrows them is void: every later kernel leaves, the stitch writes nothing, the host
    runs the batch again with the forms (zh_run_is_void). With ZULTRA_HIP_GRID_CAP the grids are small enough to outgrow: a batch inside them, then one beyond, then the
    first again — every batch's bytes are the checker's, and exactly the second is run twice."""
    monkeypatch.setenv("ZULTRA_HIP_GRID_CAP", str(cap))
    nb = (len(big) + bs - 1) // bs
    ctx = lib.context(bs, nb)
    try:
        reruns = []
        for d in (small, big, small):
            n = len(d)
            k = (n + bs - 1) // bs
            blocks = [(b * bs - (32768 if b else 0), 32768 if b else 0, min(bs, n - b * bs)) for b in range(k)]
            ctx.stitch_with_batch(k - 1, phase=0)   # (the stitch goes out with the batch: it must write nothing for a void one)
            ctx.compress_blocks(d, blocks)
            end_bit, _ = ctx.stitch_device(k - 1, phase=0)
            assert ctx.stream_read((end_bit + 7) // 8).tobytes() == checker.memory_compress(d, 0, bs), len(reruns)
            st = ctx.stats()
            reruns.append(st["batches_rerun"])
            assert (st["tasks"] > cap) == (d is big), st
        assert reruns == [0, 1, 1], reruns
    finally:
        ctx.close()


def test_a_run_that_outgrows_its_grids_is_run_again(emu, oracle, monkeypatch):
    _outgrown_grids(emu, oracle, monkeypatch, 6, corpus.text_like(8000, 5), corpus.text_like(20000, 6), 65536)   # (the emulator takes 40 s)


def test_strided_overflow_forms_of_the_per_item_kernels(emu, oracle, monkeypatch):
    _overflow_forms(emu, oracle, monkeypatch, (12288, 4, ("1",), ()))   # (48 KB: the emulator takes a minute)


def test_dictionary_stream(emu, oracle):
    t = corpus.text_like(12000, 8)
    got = emu.memory_compress(t[4000:], 1, 32768, t[:4000])
    assert got == oracle.memory_compress(t[4000:], 1, 32768, t[:4000])
    assert zlib.decompressobj(15, zdict=t[:4000].tobytes()).decompress(got) == t[4000:].tobytes()


def test_streaming_api_chunking_does_not_change_the_bytes(emu, oracle):
    # libzultra.c:259-269: blocks are cut at nMaxBlockSize whatever the chunking; tool/zultra.c:161 feeds 16 KiB chunks
    d = corpus.text_like(36000, 3)
    want = oracle.memory_compress(d, 2, 32768)
    for chunk in (16384,):   # smaller than a max-block; chunks >= a block are covered on the GPU (test_gpu_parity.py)
        s = emu.stream(2, 32768)
        out = bytearray()
        pos = 0
        st = 0
        while pos < len(d):
            part = d[pos:pos + chunk]
            pos += len(part)
            st, b = s.compress(part, finalize=(pos >= len(d)), out_chunk=5000)
            out += b
        assert st == 1   # ZULTRA_STREAM_END
        assert s.total_in == len(d) and s.total_out == len(out)
        st2, _ = s.compress(np.zeros(0, dtype=np.uint8), True)
        assert st2 == -5   # further calls: ZULTRA_ERROR_COMPRESSION (libzultra.c:204-205)
        s.end()
        assert bytes(out) == want


def test_error_behaviour_matches_reference(emu):
    t = corpus.text_like(100, 1)
    assert emu.memory_compress(t[:0], 2, 0) is None              # empty input never finalizes (libzultra.c:275)
    for cap in range(0, 12):                                      # tool/zultra.c:521-524
        assert emu.memory_compress(t, 1, 0, cap=cap) is None
    s = emu.stream(0, 0)
    s.compress(t, False)
    assert s.set_dictionary(t) == -5                              # only before the first compress (libzultra.c:180)
    s.end()


def test_memory_bound_and_checksums(emu, oracle):
    for n in (0, 1, 65535, 65536, 10 ** 6):
        for flags in (0, 1, 2):
            for bs in (0, 32768, 65536, 1 << 22):
                assert emu.memory_bound(n, flags, bs) == oracle.memory_bound(n, flags, bs)
    d = corpus.text_like(70001, 4)
    assert emu.checksum(d, 2) == zlib.crc32(d.tobytes())
    assert emu.checksum(d, 1) == zlib.adler32(d.tobytes())


def test_files_mode_each_input_is_its_own_stream(emu, oracle):
    # BASELINE configuration 5 in miniature: small JSON-like records, one raw deflate stream each
    files = [corpus.json_like(n, 40 + k) for k, n in enumerate((4096, 1500, 1, 3000))]
    data = np.concatenate(files)
    offs = np.cumsum([0] + [len(f) for f in files[:-1]])
    ctx = emu.files_context(4096, len(files))
    try:
        fo = ctx.compress_files(data, offs, [len(f) for f in files])
        stream = ctx.stream_read(int(fo[-1]))
        for k, f in enumerate(files):
            got = stream[int(fo[k]):int(fo[k + 1])].tobytes()
            assert got == oracle.memory_compress(f, 0, 32768), k
            assert zlib.decompress(got, -15) == f.tobytes()
        for k, lin in enumerate(ctx.block_crc32()):
            assert emu.crc32_append(0, lin, len(files[k])) == zlib.crc32(files[k].tobytes())
        for k, (a, bw) in enumerate(ctx.block_adler32()):   # (the same kernel takes the Adler-32 sums: zh_crc32_small, several inputs to a workgroup)
            assert emu.adler32_append(1, a, bw, len(files[k])) == zlib.adler32(files[k].tobytes()), k
    finally:
        ctx.close()


def test_files_mode_in_staggered_runs(emu, oracle, monkeypatch):
    """A files batch runs as staggered runs of inputs like a batch of max-blocks (zh_enqueue_files): with ZULTRA_HIP_STREAMS=3 twelve
    inputs are three runs of four; the sub-block descriptors come back in batch coordinates."""
    monkeypatch.setenv("ZULTRA_HIP_STREAMS", "3")
    sizes = (4096, 1500, 1, 3000, 777, 4000, 2048, 12, 3500, 4096, 100, 2500)
    files = [corpus.json_like(n, 60 + k) if k % 3 else corpus.text_like(n, 60 + k) for k, n in enumerate(sizes)]
    data = np.concatenate(files)
    offs = np.cumsum([0] + [len(f) for f in files[:-1]])
    ctx = emu.files_context(4096, len(files))
    try:
        fo = ctx.compress_files(data, offs, [len(f) for f in files])
        assert ctx.stats()["runs"] == 3
        stream = ctx.stream_read(int(fo[-1]))
        for k, f in enumerate(files):
            assert stream[int(fo[k]):int(fo[k + 1])].tobytes() == oracle.memory_compress(f, 0, 32768), k
        for k, lin in enumerate(ctx.block_crc32()):
            assert emu.crc32_append(0, lin, len(files[k])) == zlib.crc32(files[k].tobytes())
    finally:
        ctx.close()


def test_edge_sizes_and_tiny_alphabets_vs_oracle(emu, oracle):
    # every size 1..48 over alphabets of 1, 2 and 3 symbols: window ends, the last five positions of a window (not in the
    # 6-gram order), byte runs of every residue (the GPU suite runs the full grid: test_gpu_parity.py)
    rs = np.random.RandomState(7)
    files = [rs.randint(97, 97 + k, size=n).astype(np.uint8) for n in range(1, 49) for k in (1, 2, 3)]
    sizes = [len(f) for f in files]
    ctx = emu.files_context(4096, len(files))
    try:
        fo = ctx.compress_files(np.concatenate(files), np.cumsum([0] + sizes[:-1]), sizes)
        stream = ctx.stream_read(int(fo[-1]))
        for k, f in enumerate(files):
            assert stream[int(fo[k]):int(fo[k + 1])].tobytes() == oracle.memory_compress(f, 0, 32768), (k, sizes[k])
    finally:
        ctx.close()
    hist = np.concatenate(files)[-3000:]
    for f in files[60::9]:
        check_window(emu, oracle, np.concatenate([hist, f, f]), len(hist), 2 * len(f), tag="edge_tail_%d" % len(f))


def test_stream_memory_comes_from_the_callers_allocator(emu, oracle):
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
    d = corpus.text_like(20000, 6)
    s = emu.stream(2, 32768, zalloc, zfree)
    assert len(log) >= 5 and len(live) == len(log)      # the state and four per-max-block arrays
    n_init = len(log)
    st, out = s.compress(d, True)
    assert st == 1 and out == oracle.memory_compress(d, 2, 32768)
    assert len(log) == n_init                           # compressing allocates nothing more on the host side of the stream
    s.end()
    assert not live

