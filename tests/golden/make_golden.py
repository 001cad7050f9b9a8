#!/usr/bin/env python3
"""Generate tests/golden/* from the REFERENCE ITSELF (oracle/_ref/libzultra_ref.so, compiled from
/root/reference by oracle/Makefile). Run in the build container only:  python tests/golden/make_golden.py

The reference ships no golden vectors (its tests are round-trip only, tool/zultra.c:465-641), so these are
its outputs on our seeded inputs:
  * stream cases : input (seeded generator expression + sha256; small ones also zlib-packed), flags, max block size, optional dictionary -> exact compressed bytes
  * stage cases  : one window -> match rows, split offsets, per-sub-block (static cost, dynamic cost, type,
                   bit count, bits, final parse, code lengths) read through oracle/ref_probe.c
Everything stored is data (inputs and expected outputs); no reference source text is stored.
"""
import hashlib
import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import corpus  # noqa: E402
import zlibs  # noqa: E402


def stream_cases():
    """(name, generator expression over `corpus`/`np`, flags, max block, dictionary expression or None)."""
    T = "corpus.text_like(200000, 7)"
    cases = [
        ("text_gzip_64k", T, 2, 65536, None),
        ("text_zlib_32k", T + "[:150000]", 1, 32768, None),
        ("text_raw_default", T + "[:120000]", 0, 0, None),
        ("text_exact_multiple", T + "[:131072]", 2, 65536, None),
        ("tiny_100", T + "[:100]", 2, 65536, None),
        ("one_byte", T + "[:1]", 0, 65536, None),
        ("two_bytes", T + "[:2]", 1, 65536, None),
        ("mixed_gzip_64k", "corpus.mixed(160000, 11)", 2, 65536, None),
        ("text_noise_text", "np.concatenate([%s[:70000], corpus.noise(140000, 3), %s[70000:110000]])" % (T, T), 2, 65536, None),
        ("noise_stored_gt65535", "corpus.noise(100000, 5)", 0, 1 << 20, None),
        ("noise_gzip_64k", "corpus.noise(70000, 9)", 2, 65536, None),
        ("zeros", "corpus.constant(150000)", 2, 65536, None),
        ("period3", "corpus.periodic(100000, 3)", 1, 65536, None),
        ("sparse_ones", "corpus.sparse_ones(100000)", 2, 65536, None),
        ("dict_zlib", T + "[40000:140000]", 1, 65536, T + "[:32768]"),
        ("dict_raw_small", T + "[40000:100000]", 0, 65536, T + "[100:5000]"),
        ("json_4k", "corpus.json_like(4096, 3)", 2, 0, None),
        ("json_4k_b", "corpus.json_like(4096, 4)", 2, 0, None),
        # BASELINE.json configuration 1 (known answer): the image's bootstrap.min.js v3.4.1 at the default block size
        ("bootstrap_raw_default", "corpus.bootstrap_js()", 0, 0, None),
        ("bootstrap_gzip_default", "corpus.bootstrap_js()", 2, 0, None),
    ]
    for a, p in [(1, 0.0), (2, 0.5), (3, 0.3), (15, 0.5), (56, 0.7), (137, 0.9), (255, 0.995), (256, 0.0)]:
        cases.append(("selftest_a%d_p%d" % (a, int(p * 1000)), "corpus.selftest_data(40000, %d, %d, %r)" % (123 + a, a, p), 1, 32768, None))
    return cases


def stage_cases():
    t = corpus.text_like(60000, 21)
    return [
        ("stage_text", t[:8192 + 24576], 8192, 24576, 32768),
        ("stage_text_nohist", t[30000:30000 + 20000], 0, 20000, 32768),
        ("stage_mixed", corpus.mixed(40000, 5)[:36000], 4000, 32000, 32768),
        ("stage_selftest", corpus.selftest_data(20000, 77, 15, 0.5), 2000, 18000, 32768),
    ]


def pack(a):
    return zlib.compress(np.ascontiguousarray(a).tobytes(), 9)


def main():
    zlibs.build_oracle()
    R = zlibs.Ref()
    manifest = {"streams": [], "stages": []}
    for name, gen, flags, bs, dgen in stream_cases():
        data = zlibs.as_u8(eval(gen, {"corpus": corpus, "np": np}))
        d = zlibs.as_u8(eval(dgen, {"corpus": corpus, "np": np})) if dgen else None
        out = R.memory_compress(data, flags, bs, d)
        assert out is not None, name
        # inputs are regenerated from `gen` (seeded, checked against in_sha256); small ones are also stored
        ent = {"name": name, "gen": gen, "flags": flags, "max_block": bs, "n": int(len(data)),
               "in_sha256": hashlib.sha256(data.tobytes()).hexdigest(),
               "out_sha256": hashlib.sha256(out).hexdigest(), "out_len": len(out)}
        if len(pack(data)) <= 8192:
            with open(os.path.join(HERE, name + ".in.zz"), "wb") as f:
                f.write(pack(data))
            ent["in_file"] = name + ".in.zz"
        if len(out) <= 50000:
            with open(os.path.join(HERE, name + ".out"), "wb") as f:
                f.write(out)
            ent["out_file"] = name + ".out"
        if d is not None:
            ent["dict_gen"] = dgen
            ent["dict_sha256"] = hashlib.sha256(d.tobytes()).hexdigest()
        manifest["streams"].append(ent)
        print("stream", name, len(data), "->", len(out))

    for name, win, prev, n, bs in stage_cases():
        win = zlibs.as_u8(win)
        with R.probe(bs, win, prev, n) as P:
            m = P.matches()
            splits = P.split()
            subs = []
            at = prev
            blobs = {"win": win, "match": m}
            for k, e in enumerate(splits):
                size = e - at
                dyn, sc, dc = P.costs(at, size)
                rc, nb, bits, best, ll, dl = P.deflate(at, size, dyn)
                assert rc == 0
                subs.append({"start": at, "size": size, "static_cost": sc, "dynamic_cost": dc, "is_dynamic": dyn,
                             "nbits": nb, "bits_sha256": hashlib.sha256(bits).hexdigest()})
                blobs["bits%d" % k] = np.frombuffer(bits, dtype=np.uint8)
                blobs["best%d" % k] = best
                blobs["litlen%d" % k] = ll
                blobs["distlen%d" % k] = dl
                at = e
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **blobs)
        manifest["stages"].append({"name": name, "prev": prev, "n": n, "max_block": bs, "splits": splits, "subblocks": subs})
        print("stage", name, prev, n, splits)

    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    total = sum(os.path.getsize(os.path.join(HERE, x)) for x in os.listdir(HERE))
    print("total bytes in tests/golden:", total)


if __name__ == "__main__":
    main()
