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


def require_experiments(backend):
    """Skip unless libvirconv_hip.so was built with -DVC_EXPERIMENTS (the measured-and-rejected kernel variants of
    virconv_amd/csrc/experiments/: VIRCONV_HIPCC_EXTRA=-DVC_EXPERIMENTS python -m virconv_amd.build --force)."""
    import ctypes
    v = ctypes.c_int64(0)
    if backend.lib.vc_debug_get(b"experiments", ctypes.byref(v)) != 0 or v.value == 0:
        pytest.skip("experiment kernels are not in the product build (-DVC_EXPERIMENTS)")
