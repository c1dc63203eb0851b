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
        for key in _EXPERIMENT_SWITCHES:      # (require_experiments moved them; the library defaults come back for the next test)
            be.lib.vc_debug_set(key, 1)
        del _EXPERIMENT_SWITCHES[:]


_EXPERIMENT_SWITCHES = []


def require_experiments(backend):
    """Skip unless libvirconv_hip.so was built with -DVC_EXPERIMENTS (the measured-and-rejected kernel variants of
    virconv_amd/csrc/experiments/: VIRCONV_HIPCC_EXTRA=-DVC_EXPERIMENTS python -m virconv_amd.build --force)."""
    import ctypes
    v = ctypes.c_int64(0)
    if backend.lib.vc_debug_get(b"experiments", ctypes.byref(v)) != 0 or v.value == 0:
        pytest.skip("experiment kernels are not in the product build (-DVC_EXPERIMENTS)")
    # the rejected variants were written against the exact-fp32-MFMA kernels (rounds 1-3) and are compared bit for bit with them: since
    # round 4 the library default is the six-term bf16 split, which they do not have -- select their baseline for this test
    for key in (b"f32_split", b"bw_split"):
        assert backend.lib.vc_debug_set(key, 0) == 0
        _EXPERIMENT_SWITCHES.append(key)
