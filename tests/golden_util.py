"""Loader for tests/golden (vectors produced by the compiled reference; see tests/golden/make_golden.py)."""
import hashlib
import json
import os
import zlib

import numpy as np

import corpus

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

with open(os.path.join(HERE, "manifest.json")) as _f:
    MANIFEST = json.load(_f)

_cache = {}


def _gen(expr):
    if expr not in _cache:
        _cache[expr] = np.ascontiguousarray(eval(expr, {"corpus": corpus, "np": np}), dtype=np.uint8)
    return _cache[expr]


def stream_names():
    return [e["name"] for e in MANIFEST["streams"]]


def stream_case(name):
    """-> dict(data, flags, max_block, dictionary, out (bytes or None), out_sha256, out_len)"""
    e = next(x for x in MANIFEST["streams"] if x["name"] == name)
    if "in_file" in e:
        data = np.frombuffer(zlib.decompress(open(os.path.join(HERE, e["in_file"]), "rb").read()), dtype=np.uint8).copy()
    else:
        data = _gen(e["gen"])
    assert hashlib.sha256(data.tobytes()).hexdigest() == e["in_sha256"], "input generator drifted: " + name
    d = None
    if "dict_gen" in e:
        d = _gen(e["dict_gen"])
        assert hashlib.sha256(d.tobytes()).hexdigest() == e["dict_sha256"]
    out = open(os.path.join(HERE, e["out_file"]), "rb").read() if "out_file" in e else None
    return dict(name=name, data=data, flags=e["flags"], max_block=e["max_block"], dictionary=d, out=out,
                out_sha256=e["out_sha256"], out_len=e["out_len"])


def check_stream_output(case, got):
    assert got is not None, case["name"] + ": compression failed"
    assert len(got) == case["out_len"], "%s: %d bytes, reference %d" % (case["name"], len(got), case["out_len"])
    if case["out"] is not None:
        assert got == case["out"], case["name"]
    assert hashlib.sha256(got).hexdigest() == case["out_sha256"], case["name"]


def stage_names():
    return [e["name"] for e in MANIFEST["stages"]]


def stage_case(name):
    e = next(x for x in MANIFEST["stages"] if x["name"] == name)
    z = np.load(os.path.join(HERE, name + ".npz"))
    return e, z
