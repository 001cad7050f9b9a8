import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


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
