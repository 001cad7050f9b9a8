"""Shared stage-by-stage parity checker: device layer (real GPU or the CPU emulator build) vs the oracle."""
import numpy as np


def check_window(lib, oracle, win, prev, n, max_block=32768, tag="", stats_out=None):
    """Run one max-block through zultra_hip_compress_blocks and compare every stage with the oracle:
    match rows, split offsets, per-sub-block costs / type / bit count / bits / final parse."""
    win = np.ascontiguousarray(win, dtype=np.uint8)
    ctx = lib.context(max_block, 1)
    try:
        ctx.compress_blocks(win, [(0, prev, n)])
        if stats_out is not None:
            stats_out.update(ctx.stats())
        mo = oracle.find_matches(win, prev, n)
        mg = ctx.matches(0)
        if not np.array_equal(mo, mg):
            bad = int(np.argwhere((mo != mg).any(axis=(1, 2)))[0][0])
            raise AssertionError("%s match rows differ first at block pos %d: oracle %s device %s" % (tag, bad, mo[bad].tolist(), mg[bad].tolist()))
        so = oracle.split(win, mo, prev, n)
        sg = ctx.splits(0)
        assert so == sg, "%s splits: oracle %s device %s" % (tag, so, sg)
        subs, _, cnt = ctx.subblocks()
        assert cnt == len(so)
        parse = ctx.parse(0)
        at = prev
        for k, e in enumerate(so):
            sb = subs[k]
            size = e - at
            dyn, sc, dc = oracle.costs(win, mo, prev, at, size)
            assert (sb.block, sb.start, sb.size) == (0, at - prev, size), tag
            assert (sb.is_dynamic, sb.static_cost, sb.dynamic_cost) == (dyn, sc, dc), "%s sub %d costs: device %s oracle %s" % (
                tag, k, (sb.is_dynamic, sb.static_cost, sb.dynamic_cost), (dyn, sc, dc))
            rc, nb, bits, best, ll, dl = oracle.deflate(win, mo, prev, at, size, dyn)
            pg = parse[at - prev: e - prev]
            if not np.array_equal(pg, best):
                bad = int(np.argmax((pg != best).any(axis=1)))
                raise AssertionError("%s sub %d final parse differs at %d: oracle %s device %s" % (tag, k, bad, best[bad].tolist(), pg[bad].tolist()))
            if sb.failed:
                # only legal when the body cannot fit its slot, i.e. the stitcher must store it anyway
                assert nb > 8 * (size + 8), "%s sub %d flagged failed with %d bits for %d bytes" % (tag, k, nb, size)
            else:
                assert sb.nbits == nb, "%s sub %d nbits %d vs %d" % (tag, k, sb.nbits, nb)
                assert ctx.subblock_bits(sb) == bits, "%s sub %d bits differ" % (tag, k)
            at = e
        return ctx.timing()
    finally:
        ctx.close()
