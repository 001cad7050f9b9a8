"""Deterministic synthetic inputs shared by tests and bench (no files, no network).

The shapes follow SURVEY.md §8(d): the reference's own self-test generator grid (tool/zultra.c:425-463:
alphabet sizes x match probabilities, literal runs 0..127, match lengths 3..1026, offsets uniform in
history), text-like data standing in for enwik8, JSON-like 4 KiB records, plus the degenerate cases the
hot path special-cases (all-equal bytes, short periods, incompressible noise).
All generators are seeded numpy MT19937 streams, stable across numpy versions and machines.
"""
import numpy as np

SELFTEST_ALPHABETS = [1, 2, 3, 15, 30, 56, 96, 137, 178, 191, 255, 256]   # tool/zultra.c:534


def selftest_data(size, seed, n_literal_values, match_probability):
    """Same construction as the reference's generate_compressible_data, with our own PRNG."""
    rs = np.random.RandomState(seed)
    buf = np.zeros(size, dtype=np.uint8)
    if size == 0:
        return buf
    thresh = int(match_probability * 1023.0)
    buf[0] = rs.randint(0, n_literal_values)
    i = 1
    while i < size:
        if rs.randint(0, 1024) >= thresh:
            cnt = min(int(rs.randint(0, 128)), size - i)
            if cnt:
                buf[i:i + cnt] = rs.randint(0, n_literal_values, size=cnt)
                i += cnt
        else:
            ln = 3 + int(rs.randint(0, 1024))
            ln = min(ln, size - i, i)
            off = int(rs.randint(0, i - ln)) if ln < i else 0
            # overlapping copy semantics: byte by byte from i-off (off may be 0 => repeats itself)
            if off >= ln:
                buf[i:i + ln] = buf[i - off:i - off + ln]
            else:
                for k in range(ln):
                    buf[i + k] = buf[i + k - off]
            i += ln
    return buf


_WORDS = None


def _vocab(rs, n=4096):
    letters = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
    weights = np.array([12.7, 9.1, 8.2, 7.5, 7.0, 6.7, 6.3, 6.1, 6.0, 4.3, 4.0, 2.8, 2.8, 2.4, 2.4, 2.2, 2.0, 2.0, 1.9,
                        1.5, 1.0, 0.8, 0.15, 0.15, 0.1, 0.07])
    weights = weights / weights.sum()
    words = []
    for _ in range(n):
        ln = 1 + min(int(rs.geometric(0.25)), 14)
        words.append(bytes(rs.choice(letters, size=ln, p=weights)))
    return words


def text_like(size, seed=1):
    """Zipf-distributed words, sentences, paragraphs and wiki-ish markup: an enwik8 stand-in."""
    rs = np.random.RandomState(seed)
    words = _vocab(np.random.RandomState(12345))
    ranks = rs.zipf(1.25, size=size // 3 + 64)
    ranks = np.minimum(ranks - 1, len(words) - 1)
    out = bytearray()
    k = 0
    sent = 0
    while len(out) < size:
        w = words[int(ranks[k])]
        k += 1
        if sent == 0:
            w = w.capitalize()
        r = rs.randint(0, 100)
        if r < 3:
            out += b"[[" + w + b"]]"
        elif r < 5:
            out += b"'''" + w + b"'''"
        elif r < 6:
            out += b"&quot;" + w + b"&quot;"
        elif r < 7:
            out += str(int(rs.randint(0, 2100))).encode()
        else:
            out += w
        sent += 1
        if sent > 4 and rs.randint(0, 12) == 0:
            out += b". "
            sent = 0
            if rs.randint(0, 8) == 0:
                out += b"\n\n" if rs.randint(0, 4) else b"\n== " + words[int(ranks[k])] + b" ==\n"
        elif rs.randint(0, 15) == 0:
            out += b", "
        else:
            out += b" "
    return np.frombuffer(bytes(out[:size]), dtype=np.uint8).copy()


def json_like(size, seed=1):
    """JSON-ish records (config 5 of BASELINE.json), truncated/padded to ``size`` bytes."""
    rs = np.random.RandomState(seed)
    names = [w.decode() for w in _vocab(np.random.RandomState(777), 64)]
    out = bytearray()
    rid = int(rs.randint(0, 1 << 30))
    while len(out) < size:
        tags = ",".join('"%s"' % names[int(t)] for t in rs.randint(0, 64, size=int(rs.randint(0, 5))))
        out += ('{"id":%d,"user":"%s","ts":%d,"tags":[%s],"v":%.4f}\n' % (
            rid, names[int(rs.randint(0, 64))], 1700000000 + int(rs.randint(0, 1 << 24)), tags, rs.rand() * 1000)).encode()
        rid += 1 + int(rs.randint(0, 3))
    return np.frombuffer(bytes(out[:size]), dtype=np.uint8).copy()


def noise(size, seed=1):
    return np.random.RandomState(seed).randint(0, 256, size=size).astype(np.uint8)


def constant(size, value=0):
    return np.full(size, value, dtype=np.uint8)


def periodic(size, period=3, seed=1):
    pat = np.random.RandomState(seed).randint(0, 256, size=period).astype(np.uint8)
    return np.resize(pat, size)


def sparse_ones(size, seed=1, gap=200):
    """Zeros with a 1 at irregular gaps: long-but-not-maximal matches everywhere (matchfinder stress)."""
    rs = np.random.RandomState(seed)
    buf = np.zeros(size, dtype=np.uint8)
    i = 0
    while i < size:
        i += int(rs.randint(gap // 2, gap * 2))
        if i < size:
            buf[i] = 1
    return buf


def indented(size, seed=1):
    """Source-code-like lines: runs of spaces of many lengths (incl. > 258), zero padding, repeated line bodies — the byte-run
    classes of the matchfinder (zh_matchfinder.h) in every combination: same / shorter / longer earlier runs, equal runs with
    and without a continuing match."""
    rs = np.random.RandomState(seed)
    words = [b"def", b"return", b"self", b"x", b"if", b"else:", b"value", b"for i in range(n):", b"pass", b"0", b"=="]
    out = bytearray()
    while len(out) < size:
        kind = rs.randint(0, 20)
        if kind == 0:
            out += bytes(int(rs.randint(4, 600)))            # zero padding, sometimes longer than the match limit
        elif kind == 1:
            out += b" " * int(rs.randint(250, 300)) + b"\n"
        else:
            ind = int(rs.choice([0, 1, 2, 3, 4, 4, 4, 5, 7, 8, 8, 8, 12, 12, 16, 20, 40]))
            body = b" ".join(words[int(t)] for t in rs.randint(0, len(words), size=int(rs.randint(1, 5))))
            out += b" " * ind + body + (b" " * int(rs.randint(0, 6)) if rs.randint(0, 4) == 0 else b"") + b"\n"
    return np.frombuffer(bytes(out[:size]), dtype=np.uint8).copy()


def duplicated(size, seed=1, edit_gap=400):
    """Text followed by near-copies of itself (a byte changed every ~edit_gap bytes): long matches at nearly every position,
    barrier-free runs of thousands of positions — the cooperative phase of the parse (zh_parse.h) with short, 19..39 and
    long candidates, literals and sub-block ends in the mix."""
    rs = np.random.RandomState(seed)
    base = text_like(max(64, size // 3), seed + 100)
    out = [base]
    n = len(base)
    while n < size:
        c = base.copy()
        at = int(rs.randint(1, edit_gap))
        while at < len(c):
            c[at] = rs.randint(97, 123)
            at += int(rs.randint(1, 2 * edit_gap))
        k = int(rs.randint(0, 50))
        out.append(c[k:])
        n += len(c) - k
    return np.concatenate(out)[:size]


def table_like(size, seed=1):
    """Lines of a generated table — a fixed frame around a counter, a code and a few words from a small vocabulary (the codec
    tables among the bench's Python sources look like this): medium matches overlap without a gap for thousands of positions,
    but few of them are 258 long. These are the barrier-free runs whose costs forget their start quickly: the speculative
    segments of the chain parse (zh_parse_chain.h) mostly verify on such data."""
    rs = np.random.RandomState(seed)
    words = [w.upper() for w in ("letter", "capital", "small", "with", "sign", "digit", "box", "drawings", "light", "heavy", "double", "vertical",
                                 "horizontal", "arabic", "greek", "cyrillic", "latin", "acute", "grave", "shade", "block", "quotation", "mark")]
    out = bytearray()
    k = int(rs.randint(0, 4096))
    while len(out) < size:
        nw = int(rs.randint(2, 6))
        name = " ".join(words[int(rs.randint(0, len(words)))] for _ in range(nw))
        out += ("    '\\u%04x'    #  0x%02X -> %s\n" % (0x2500 + (k * 7) % 3000, k & 255, name)).encode()
        k += 1
    return np.frombuffer(bytes(out[:size]), dtype=np.uint8).copy()


def fibonacci_bytes(k=19, seed=1):
    """Bytes whose counts are the Fibonacci numbers F(1)..F(k), shuffled: the unlimited Huffman tree of such a histogram is
    a chain of depth k-1 (> 15 from k = 17: the cost estimates price code lengths the format cannot even encode). Matches
    flatten the token histogram, so this only leans on the deep-tree paths; the regression test for code lengths above 15
    in the cost estimates is the 2 MiB block of source code in test_gpu_parity.py."""
    rs = np.random.RandomState(seed)
    fib = [1, 1]
    while len(fib) < k:
        fib.append(fib[-1] + fib[-2])
    d = np.concatenate([np.full(f, 33 + i, dtype=np.uint8) for i, f in enumerate(fib)])
    rs.shuffle(d)
    return d


def mixed(size, seed=1):
    """Segments cycling through the self-test grid plus noise and constant runs (config 4 shape)."""
    rs = np.random.RandomState(seed)
    probs = [0.0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 0.995]
    parts = []
    total = 0
    k = 0
    while total < size:
        seg = min(int(rs.randint(4096, 49152)), size - total)
        sel = k % 16
        if sel == 14:
            parts.append(noise(seg, seed + k))
        elif sel == 15:
            parts.append(constant(seg, int(rs.randint(0, 256))))
        elif sel % 3 == 2:
            parts.append(text_like(seg, seed + k))
        else:
            parts.append(selftest_data(seg, seed + k, SELFTEST_ALPHABETS[k % 12], probs[(k * 7) % 11]))
        total += seg
        k += 1
    return np.concatenate(parts)[:size]


def text_like_fast(size, seed=1):
    """Vectorised variant of text_like for benchmark-sized inputs (100 MB in a few seconds): Zipf word stream with
    spaces, sentence punctuation, paragraph breaks and a sprinkle of wiki-style link / entity markup."""
    rs = np.random.RandomState(seed)
    words = _vocab(np.random.RandomState(4242), 65536)
    extra = [b"[[" + w + b"]]" for w in words[:256]] + [b"&quot;" + w + b"&quot;" for w in words[:128]] + \
            [b"'''" + w + b"'''" for w in words[:128]] + [str(y).encode() for y in range(1800, 2056)]
    table = words + extra
    maxlen = max(len(w) for w in table) + 2
    tab = np.zeros((len(table), maxlen), dtype=np.uint8)
    lens = np.zeros(len(table), dtype=np.int64)
    for i, w in enumerate(table):
        tab[i, :len(w)] = np.frombuffer(w, dtype=np.uint8)
        lens[i] = len(w)
    out = np.empty(size + 64, dtype=np.uint8)
    pos = 0
    while pos < size:
        nw = min(4_000_000, (size - pos) // 4 + 64)
        ids = rs.zipf(1.19, size=nw) - 1
        ids = np.where(ids >= len(words), rs.randint(0, len(words), size=nw), ids)
        r = rs.randint(0, 100, size=nw)
        ids = np.where(r < 7, len(words) + rs.randint(0, len(extra), size=nw), ids)
        sep = rs.randint(0, 64, size=nw)
        # separator: 0 -> ". " , 1 -> ", ", 2 -> ".\n\n" (rare: only when sep==2 and r<50), else " "
        seplen = np.where(sep == 0, 2, np.where(sep == 1, 2, np.where((sep == 2) & (r < 50), 3, 1)))
        wl = lens[ids]
        tot = wl + seplen
        offs = np.cumsum(tot) - tot
        need = int(offs[-1] + tot[-1])
        buf = np.full(need + 4, 32, dtype=np.uint8)
        for j in range(maxlen):
            m = wl > j
            if not m.any():
                break
            buf[offs[m] + j] = tab[ids[m], j]
        so = offs + wl
        m0 = sep == 0
        buf[so[m0]] = ord(".")
        m1 = sep == 1
        buf[so[m1]] = ord(",")
        m2 = (sep == 2) & (r < 50)
        buf[so[m2]] = ord(".")
        buf[so[m2] + 1] = 10
        buf[so[m2] + 2] = 10
        take = min(need, size - pos)
        out[pos:pos + take] = buf[:take]
        pos += take
    return out[:size]


# ---- bulk generators in C (tests/gen/zgen.c): BASELINE.json configurations 4 and 5 at full size --------------------
CONFIG4_SEED = 0x5EED
CONFIG4_SEGMENT = 1 << 20
_zgen = None


def _zgen_lib():
    global _zgen
    if _zgen is None:
        import ctypes as C
        import os
        import subprocess
        here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gen")
        so = os.path.join(here, "_build", "libzgen.so")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(here, "zgen.c")):
            subprocess.run(["make", "-s", "-C", here], check=True)
        L = C.CDLL(so)
        L.zgen_mixed.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
        L.zgen_mixed.restype = None
        L.zgen_json_files.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64]
        L.zgen_json_files.restype = None
        _zgen = L
    return _zgen


def mixed_config4(first_segment, nsegments, seed=CONFIG4_SEED, out=None):
    """Segments [first_segment, first_segment + nsegments) of the configuration-4 stream (1 MiB each):
    the self-test grid of alphabets x match probabilities, every 16th segment noise, every 16th one repeated byte.
    Every segment depends only on (seed, its global index): ranks generate their own shard of the one 8 GiB stream."""
    if out is None:
        out = np.empty(nsegments * CONFIG4_SEGMENT, dtype=np.uint8)
    assert out.dtype == np.uint8 and out.size >= nsegments * CONFIG4_SEGMENT and out.flags["C_CONTIGUOUS"]
    _zgen_lib().zgen_mixed(out.ctypes.data, first_segment, nsegments, seed)
    return out[: nsegments * CONFIG4_SEGMENT]


def json_files(first_file, nfiles, file_size=4096, seed=5, out=None):
    """Files [first_file, first_file + nfiles) of the configuration-5 corpus: independent JSON-like inputs of file_size bytes,
    laid end to end."""
    if out is None:
        out = np.empty(nfiles * file_size, dtype=np.uint8)
    assert out.dtype == np.uint8 and out.size >= nfiles * file_size and out.flags["C_CONTIGUOUS"]
    _zgen_lib().zgen_json_files(out.ctypes.data, first_file, nfiles, file_size, seed)
    return out[: nfiles * file_size]


def real_text(size):
    """Real text present in this image (and, identically, on the GPU box): the Python sources under /usr/lib/python3* and
    /usr/local/lib/python3*, sorted by path and concatenated, cycled to `size` bytes (windows are 32 KiB: the tiling is
    invisible to the compressor). Stands in for enwik8, which is not in the image."""
    import glob
    parts, total = [], 0
    files = sorted(glob.glob("/usr/lib/python3*/**/*.py", recursive=True)) + sorted(glob.glob("/usr/local/lib/python3*/**/*.py", recursive=True))
    for f in files:
        try:
            b = np.fromfile(f, dtype=np.uint8)
        except OSError:
            continue
        if b.size:
            parts.append(b)
            total += b.size
        if total >= size:
            break
    if not parts:
        raise RuntimeError("no Python sources found for the real-text corpus")
    d = np.concatenate(parts)
    if d.size < size:
        d = np.resize(d, size)
    return d[:size].copy()


BOOTSTRAP_JS = "/opt/conda/lib/python3.9/site-packages/notebook/static/components/bootstrap/dist/js/bootstrap.min.js"


def bootstrap_js():
    """BASELINE.json configuration 1 stand-in: the image's bootstrap.min.js v3.4.1, 39 680 bytes (the
    48 944-byte file of the reference's README is not in the image). Raises FileNotFoundError where the image lacks it."""
    return np.fromfile(BOOTSTRAP_JS, dtype=np.uint8)
