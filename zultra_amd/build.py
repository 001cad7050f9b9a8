"""Build the product library zultra_amd/libzultra_amd.so for gfx950 (MI355X) with hipcc.

    python -m zultra_amd.build          (or __graft_entry__.build())

One shared object holds the HIP kernels, the C-ABI device layer (include/zultra_hip.h) and the drop-in
libzultra API (include/libzultra.h). It is built in-tree so that it travels with the repository snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libzultra_amd.so")
CLI = os.path.join(HERE, "zultra_amd_cli")
SOURCES = ["zh_device.hip", "libzultra.cpp"]
HEADERS = ["zh_platform.h", "zh_common.h", "zh_matchfinder.h", "zh_mf_group_lds.h", "zh_huffman.h", "zh_split.h", "zh_parse.h", "zh_parse_chain.h", "zh_parse_lanes.h", "zh_encode.h", "zh_stitch.h"]


def hipcc_path():
    for p in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if p and os.path.exists(p):
            return p
    raise RuntimeError("hipcc not found: the product library cannot be built without the ROCm toolchain")


def needs_build():
    if not os.path.exists(OUT) or not os.path.exists(CLI):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS + ["zultra_cli.c"]]
    deps += [os.path.join(HERE, "..", "include", f) for f in ("libzultra.h", "zultra_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip",
           "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
           "-I", CSRC, "-o", OUT] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    # the command-line tool (csrc/zultra_cli.c): plain C against include/libzultra.h, linked against the library just built
    cli = [shutil.which("gcc") or "gcc", "-O2", "-Wall", "-o", CLI, os.path.join(CSRC, "zultra_cli.c"), "-L", HERE, "-lzultra_amd", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cli), flush=True)
    subprocess.run(cli, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
