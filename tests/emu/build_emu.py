"""TEST INFRASTRUCTURE: build tests/emu/_build/libzultra_amd_emu.so — the product's own sources (zultra_amd/csrc)
compiled for the CPU against tests/emu/zh_platform.h, a lock-step emulator of the HIP subset they use.
Lets the `-m "not gpu"` suite run the kernel logic against the oracle on a machine without a GPU.
The product package never loads this file; its loader only accepts zultra_amd/libzultra_amd.so."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "zultra_amd", "csrc")
OUT = os.path.join(HERE, "_build", "libzultra_amd_emu.so")


def build(force=False):
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "zh_platform.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = ["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-DZH_TOK_CHUNK=1024u", "-DZH_CUT_LEN=512u", "-DZH_CUT_WARM=288u", "-DZH_MFL_CAP_LIMIT=512u",
           "-Wno-unknown-pragmas", "-I", HERE, "-I", CSRC, "-x", "c++", os.path.join(CSRC, "zh_device.hip"),
           os.path.join(CSRC, "libzultra.cpp"), "-o"]
    cmd[1:1] = os.environ.get("ZH_EMU_DEFINES", "").split()   # extra -D switches (A/B of compile-time variants under the emulator)
    tmp = "%s.%d.tmp" % (OUT, os.getpid())   # several test processes may build at once: each writes its own file, the rename is atomic
    subprocess.run(cmd + [tmp], check=True)
    os.replace(tmp, OUT)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
