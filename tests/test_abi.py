"""CPU: the product library builds for gfx950 and exports every symbol include/*.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(zultra_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def so_path():
    import shutil
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    from zultra_amd import build
    return build.build(verbose=False)


def test_library_exports_every_declared_symbol(so_path):
    L = ctypes.CDLL(so_path)
    names = _declared("libzultra.h") + _declared("zultra_hip.h")
    assert len(names) >= 28
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_bindings_cover_the_headers(so_path):
    from zultra_amd import _ffi
    declared = set(_declared("libzultra.h") + _declared("zultra_hip.h"))
    assert declared == set(_ffi.EXPORTS), declared ^ set(_ffi.EXPORTS)


def test_no_device_means_loud_failure(so_path):
    """In this GPU-less container the library must refuse to compress rather than fall back to a CPU path."""
    L = ctypes.CDLL(so_path)
    L.zultra_hip_device_count.restype = ctypes.c_int
    if L.zultra_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    from zultra_amd._ffi import Lib, ZultraError
    lib = Lib(so_path)
    assert lib.memory_compress(b"hello hello hello hello", 2, 0) is None
    with pytest.raises(ZultraError):
        lib.context(65536, 1)
    with pytest.raises(ZultraError):
        lib.stream(2, 0)


def test_product_never_references_the_oracle():
    pkg = os.path.join(ROOT, "zultra_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "zultra_oracle" not in txt and "libzultra_ref" not in txt, f
                assert "oracle/" not in txt.replace("// oracle/", ""), f
