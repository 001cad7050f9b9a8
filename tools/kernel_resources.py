#!/usr/bin/env python3
"""Prints registers / LDS / occupancy of every kernel in zh_device.hip as reported by the compiler (no GPU needed); exits 1 when a kernel
uses scratch memory (spilled registers, arrays indexed at run time, out-of-line calls): none of the product's kernels may."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "zultra_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "-I", CSRC, "-c", os.path.join(CSRC, "zh_device.hip"),
       "-o", "/tmp/zh_device_res.o", "-Rpass-analysis=kernel-resource-usage"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = re.sub(r"^_Z\d+", "", t.split(":", 1)[1].strip())
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
print("%-34s %6s %6s %8s %9s %6s" % ("kernel", "VGPRs", "SGPRs", "scratch", "LDS B/WG", "occ"))
for k, r in rows.items():
    print("%-34s %6s %6s %8s %9s %6s" % (k[:34], r.get("VGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"),
                                       r.get("Occupancy [waves/SIMD]")))
bad = [k for k, r in rows.items() if r.get("ScratchSize [bytes/lane]", "0") != "0"]
if bad:
    print("kernels with scratch memory: %s" % ", ".join(bad))
    sys.exit(1)
