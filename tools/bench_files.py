#!/usr/bin/env python3
"""BASELINE.json configuration 5: N x 4 KiB JSON-like inputs, each its own gzip stream, on one MI355X.
Batches of independent inputs go through the files interface of the device layer (include/zultra_hip.h): one captured
hipGraph replay per batch for stages 1-3, one stitch launch, D2H of the raw streams, gzip framing on the host.
Prints one JSON line (files/s and input MB/s; input resident in HBM before the timed region)."""
import argparse
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=1 << 20)
    ap.add_argument("--batch", type=int, default=1 << 16)
    ap.add_argument("--size", type=int, default=4096)
    args = ap.parse_args()
    import torch

    import corpus
    import zultra_amd
    L = zultra_amd.lib()
    # one batch worth of distinct records, reused for every batch (generation is slow in Python; contents do not matter to the rate)
    base = corpus.json_like(args.size * 2048, 9)
    tile = np.concatenate([base] * (args.batch // 2048 + 1))[: args.batch * args.size].copy()
    tile[::args.size] = (np.arange(args.batch) & 0xff).astype(np.uint8)   # make the inputs differ
    d = torch.from_numpy(tile).cuda()
    torch.cuda.synchronize()
    offs = np.arange(args.batch, dtype=np.uint64) * args.size
    sizes = np.full(args.batch, args.size, dtype=np.uint32)
    ctx = L.files_context(args.size, args.batch)
    nb = (args.files + args.batch - 1) // args.batch
    fo = ctx.compress_files(d.data_ptr(), offs, sizes, data_on_device=True, data_size=d.numel())   # warm-up: captures the graph
    t0 = time.perf_counter()
    out_bytes = 0
    for _ in range(nb):
        fo = ctx.compress_files(d.data_ptr(), offs, sizes, data_on_device=True, data_size=d.numel())
        stream = ctx.stream_read(int(fo[-1]))
        crcs = ctx.block_crc32()
        out_bytes += int(fo[-1]) + 18 * args.batch
    dt = time.perf_counter() - t0
    # spot check: frame one input as gzip and inflate it
    k = 12345 % args.batch
    raw = stream[int(fo[k]):int(fo[k + 1])].tobytes()
    gz = bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 2, 255]) + raw + int(L.crc32_append(0, crcs[k], args.size)).to_bytes(4, "little") + int(args.size).to_bytes(4, "little")
    ok = zlib.decompress(gz, 31) == tile[k * args.size:(k + 1) * args.size].tobytes()
    t = ctx.timing()
    print(json.dumps({"metric": "files/s, 4 KiB JSON-like inputs, one gzip stream each, hipGraph replay per batch", "value": round(nb * args.batch / dt, 1),
                      "unit": "files/s", "input_MBps": round(nb * args.batch * args.size / dt / 1e6, 2), "files": nb * args.batch, "batch": args.batch,
                      "ratio": round(out_bytes / (nb * args.batch * args.size), 4), "graph_ms_per_batch": round(t["encode_ms"], 3),
                      "stitch_ms_per_batch": round(t["stitch_ms"], 3), "spot_check_inflates": bool(ok)}))
    ctx.close()


if __name__ == "__main__":
    main()
