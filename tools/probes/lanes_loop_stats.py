#!/usr/bin/env python3
"""Diagnostics (no GPU needed): instruction mix of the step loop of zh_parse_lanes — the batch loop of zh_lp_group, four steps of sixteen positions per round —
from the compiler's assembly. usage: python tools/probes/lanes_loop_stats.py [extra hipcc flags]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "zultra_amd", "csrc")
with tempfile.TemporaryDirectory() as td:
    asm = os.path.join(td, "dev.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-I", CSRC, "-S", "--cuda-device-only", "-o", asm,
                    os.path.join(CSRC, "zh_device.hip")] + sys.argv[1:], check=True, stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith("_Z14zh_parse_lanes")][0]
end = [i for i, l in enumerate(lines) if i > start and l.startswith(".Lfunc_end")][0]
body = lines[start:end]
labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
loops = []
for i, l in enumerate(body):
    m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))


def count(a, b):
    c = collections.Counter()
    for l in body[a:b + 1]:
        m = re.match(r"^\s+([a-z_0-9]+)", l)
        if m and not l.strip().startswith((".", ";")):
            c[m.group(1)] += 1
    return c


# the step loop: the loop with quad DPP minima that is not the whole group loop (the smallest one that holds all sixteen quad-min instructions)
cands = [(b - a, a, b) for a, b in loops if sum("quad_perm" in l for l in body[a:b + 1]) >= 16]
_, a, b = min(cands)
c = count(a, b)
cat = collections.Counter()
for o, v in c.items():
    cat["valu" if o.startswith("v_") else "salu" if o.startswith("s_") else "lds" if o.startswith("ds_") else "vmem" if o.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"] += v
regs = [l for l in body if "vgpr_count" in l or ".vgpr_count" in l]
print("step loop (four steps of sixteen positions): %d instructions: %s" % (sum(c.values()), dict(cat)))
print("   v_mov %d (of which zero %d), v_cndmask %d, s_nop %d, v_min %d, v_add %d" % (
    sum(v for o, v in c.items() if o.startswith("v_mov")), sum(1 for l in body[a:b + 1] if re.match(r"\s+v_mov_b32_e32 v\d+, 0$", l)),
    sum(v for o, v in c.items() if o.startswith("v_cndmask")), c["s_nop"], sum(v for o, v in c.items() if o.startswith("v_min")), sum(v for o, v in c.items() if o.startswith("v_add"))))
