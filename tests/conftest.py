import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The test-only CPU oracle (oracle/liboracle.so), built on demand."""
    from oracle import orc as _orc
    _orc.build(ref=os.path.exists("/root/reference/src/distance-cpu.c"))
    return _orc


@pytest.fixture(scope="session")
def ref_cpu(orc):
    if not orc.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference on this box); golden fixtures cover this")
    return orc.RefKernels("cpu")


@pytest.fixture(scope="session")
def ref_avx2(orc):
    if not orc.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference on this box); golden fixtures cover this")
    return orc.RefKernels("avx2")
