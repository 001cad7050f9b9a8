#!/usr/bin/env python3
"""Issue and residency figures per kernel from a rocprofv3 SQ-counter pass of `bench.py --profile-run` (rocpd database), as the JSON that
bench.py reads back (profiles/*_sq_c<N>.json):
   valu_issue_frac   = sum of SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x cycles of a step)         (MI355X_MICROARCH.md: a wave64 VALU
                       instruction issues over 2 cycles on a SIMD-32; 256 CUs x 4 SIMDs; 2.4 GHz)
   resident waves    = SQ_WAVE_CYCLES x 4 / cycles the kernel was running (the counter ticks once per four cycles and wave), per kernel:
                       the average number of its waves on the chip while it ran (16 384 wave slots of 32 per CU; a kernel of 128
                       registers can fill 4096)
usage: python tools/sq_profile.py <sq_results.db> <out.json> <config> <steps profiled> [source note] [ms per step of the profiled run]"""
import json
import re
import sqlite3
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zultra_amd  # noqa: E402  (csrc_digest: the sources this profile was measured on)

CLOCK_HZ, SIMDS = 2.4e9, 1024
db = sqlite3.connect(sys.argv[1])
config, steps = int(sys.argv[3]), max(1, int(sys.argv[4]))
note = sys.argv[5] if len(sys.argv) > 5 else ""


def key(name):
    return re.sub(r"<.*>", "", name.split("(")[0].replace("void ", ""))


counters = {}
for name, cname, n, total in db.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
    # (the <true> and <false> instances of a kernel share a key: their counters ADD — round 5's file held whichever instance came last, zeros for five kernels)
    n0, t0 = counters.setdefault(key(name), {}).get(cname, (0, 0.0))
    counters[key(name)][cname] = (n0 + n, t0 + float(total))
dur = {}
for name, n, total in db.execute("select name, count(*), sum(duration) from kernels group by name"):
    k = key(name)
    n0, t0 = dur.get(k, (0, 0.0))
    dur[k] = (n0 + n, t0 + float(total))
res = {"config": config, "steps_profiled": steps,
       "source": "rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT of: " + note,
       "kernels": {}}
valu = 0.0
for k, c in sorted(counters.items()):
    if k.startswith("__amd") or k.startswith("zh_probe"):
        continue
    n, ns = dur.get(k, (0, 0.0))
    wc = c.get("SQ_WAVE_CYCLES", (0, 0.0))[1]
    iv = c.get("SQ_INSTS_VALU", (0, 0.0))[1]
    valu += iv
    e = {"launches_per_step": round(c.get("SQ_WAVES", (n, 0))[0] / steps, 2), "valu_insts_per_step": round(iv / steps),
         "wave_cycles_x4_per_step": round(4.0 * wc / steps), "waves_per_launch": round(c.get("SQ_WAVES", (1, 0.0))[1] / max(1, c.get("SQ_WAVES", (1, 0.0))[0]), 1)}
    if ns > 0:
        e["resident_waves_avg"] = round(4.0 * wc / (ns * 1e-9 * CLOCK_HZ), 1)
        e["kernel_ms_per_step"] = round(ns / steps / 1e6, 3)
    if wc > 0:
        e["valu_active_frac_of_wave_cycles"] = round(c.get("SQ_ACTIVE_INST_VALU", (0, 0.0))[1] / wc, 3)   # (both count quad-cycles)
        e["parked_frac_of_wave_cycles"] = round(c.get("SQ_WAIT_ANY", (0, 0.0))[1] / wc, 3)
    res["kernels"][k] = e
res["valu_insts_per_step"] = round(valu / steps)
res["valu_issue_ms_per_step"] = round(valu / steps * 2.0 / SIMDS / CLOCK_HZ * 1e3, 3)
if len(sys.argv) > 6 and float(sys.argv[6]) > 0:
    res["step_ms_profiled"] = round(float(sys.argv[6]), 3)   # the profiled run's own step: the denominator of valu_issue_frac (bench.py)
res["csrc_digest"] = zultra_amd.csrc_digest()
with open(sys.argv[2], "w") as f:
    json.dump(res, f, indent=1)
print("VALU instructions per step %.3e = %.2f ms of issue at 1024 SIMDs x 1/2 per cycle x 2.4 GHz" % (res["valu_insts_per_step"], res["valu_issue_ms_per_step"]))
for k, e in sorted(res["kernels"].items(), key=lambda kv: -kv[1]["wave_cycles_x4_per_step"]):
    print("%-22s resident waves %8s  VALU active %5s  parked %5s  VALU insts/step %.2e" % (k, e.get("resident_waves_avg"), e.get("valu_active_frac_of_wave_cycles"), e.get("parked_frac_of_wave_cycles"), e["valu_insts_per_step"]))
