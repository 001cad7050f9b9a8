import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (`-m "not gpu"`) spends its time in the lock-step emulator, one thread per test: when pytest-xdist is there it is spread
    over four processes (set ZULTRA_TESTS_SERIAL=1, or pass -n yourself, to decide otherwise). Anything else — the GPU run above all — stays
    in one process: one device, and tests that time things."""
    if os.environ.get("PYTEST_XDIST_WORKER") or hasattr(config, "workerinput"):
        return   # (a worker of such a run sees the same options: it must not spread itself again)
    if (config.option.markexpr.strip() == "not gpu" and not getattr(config.option, "numprocesses", None) and config.pluginmanager.hasplugin("xdist")
            and not os.environ.get("ZULTRA_TESTS_SERIAL") and not getattr(config.option, "usepdb", False)):
        config.option.numprocesses = min(4, os.cpu_count() or 1)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import zlibs
    return zlibs.Oracle()


@pytest.fixture(scope="session")
def ref():
    import zlibs
    if not zlibs.have_ref():
        if os.path.isdir("/root/reference/src"):
            zlibs.build_oracle()
        else:
            pytest.skip("compiled reference (oracle/_ref) not available")
    return zlibs.Ref()


@pytest.fixture(scope="session")
def checker(oracle):
    """What the GPU suite compares the device with: the COMPILED REFERENCE itself (oracle/_ref/libzultra_ref.so, built from /root/reference by
    oracle/Makefile; it travels to the GPU box with the snapshot) behind the oracle's interface — stage by stage and on whole streams — and, only
    where that file is absent, the restatement that tests/test_oracle_vs_ref.py pins to it."""
    import zlibs
    if zlibs.have_ref():
        return zlibs.RefStages()
    return oracle
