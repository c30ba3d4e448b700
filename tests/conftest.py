import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as orc
    orc.build()
    return orc


# PyTorch bundles its own copy of the HIP runtime (same soname as /opt/rocm's).  When a test uses torch tensors on the GPU
# next to libforma_hip.so (the multi-GPU exchange path), torch must be loaded FIRST so that both share one runtime;
# loaded the other way round torch fails with "No HIP GPUs are available".
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    pass
