#!/usr/bin/env python3
"""HBM traffic per kernel launch from two rocprofv3 PMC passes of `bench.py --profile-run` (FETCH_SIZE and WRITE_SIZE
need separate passes: MI355X_MICROARCH.md, rocprofv3 PMC slots). Counter values are KiB per dispatch. The kernels move
4 bytes per lane, a width the guide leaves uncalibrated, so both counters are scaled by what they report for
zh_probe_copy_dword, a streaming dword copy of known size run in the same passes.

usage: python tools/pmc_traffic.py <fetch_results.db> <write_results.db> <probe_bytes> <out.json> [config] [source note]"""
import json
import re
import sqlite3
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zultra_amd  # noqa: E402  (csrc_digest: the sources this profile was measured on)


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    out = {}
    for name, n, total in db.execute(
            "select kernel_name, count(*), sum(value) from counters_collection where counter_name=? group by kernel_name", (counter,)):
        key = re.sub(r"<.*>", "", name.split("(")[0].replace("void ", ""))
        n0, t0 = out.get(key, (0, 0.0))
        out[key] = (n0 + n, t0 + total * 1024.0)
    return out


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
probe_bytes = float(sys.argv[3])
pf, pw = fetch["zh_probe_copy_dword"], write["zh_probe_copy_dword"]
cal_f = probe_bytes / (pf[1] / pf[0])     # true bytes per reported byte
cal_w = probe_bytes / (pw[1] / pw[0])
config = int(sys.argv[5]) if len(sys.argv) > 5 else 2
note = sys.argv[6] if len(sys.argv) > 6 else "python3 bench.py --config %d --profile-run --steps 1 --warmup 1" % config
res = {"config": config, "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of: " + note,
       "calibration": {"probe": "zh_probe_copy_dword, %d bytes read and written, 4 B per lane" % int(probe_bytes),
                       "fetch_true_over_reported": round(cal_f, 4), "write_true_over_reported": round(cal_w, 4)},
       "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    if k.startswith("__amd") or k == "zh_probe_copy_dword":
        continue
    nf, bf = fetch.get(k, (0, 0.0))
    nw, bw = write.get(k, (0, 0.0))
    n = max(nf, nw, 1)
    res["kernels"][k] = {"launches": n, "fetch_bytes_per_launch_reported": round(bf / n), "write_bytes_per_launch_reported": round(bw / n),
                         "hbm_bytes_per_launch": round((bf * cal_f + bw * cal_w) / n)}
# the hipGraph of files mode (configuration 5) as one item: all its kernels of one replay
res["kernels"]["graph"] = {"launches": max(1, res["kernels"].get("zh_stitch", {}).get("launches", 1)),
                           "hbm_bytes_per_launch": round(sum(v["hbm_bytes_per_launch"] * v["launches"] for k, v in res["kernels"].items() if k != "zh_stitch") /
                                                         max(1, res["kernels"].get("zh_stitch", {}).get("launches", 1)))}
res["csrc_digest"] = zultra_amd.csrc_digest()
with open(sys.argv[4], "w") as f:
    json.dump(res, f, indent=1)
print(json.dumps(res["calibration"]))
for k, v in sorted(res["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]):
    print("%-24s launches %3d  HBM bytes/launch %14d" % (k, v["launches"], v["hbm_bytes_per_launch"]))
