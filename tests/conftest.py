import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with `pytest -m gpu`)")


@pytest.fixture
def oracle_backend():
    """Run virconv_amd host logic on the CPU oracle operators (tests only)."""
    from oracle.backend import OracleBackend
    from virconv_amd import ops
    with ops.use_backend(OracleBackend()) as be:
        yield be


@pytest.fixture
def hip_backend():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from virconv_amd import ops
    from virconv_amd.backend_hip import HipBackend
    with ops.use_backend(HipBackend()) as be:
        yield be
