import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_tier(config):
    """True when the run selects the GPU tier (-m gpu): there a missing checker or a missing GPU is a FAILURE, not a skip -- a box where
    oracle/_ref did not travel must not turn the parity tests into a green run of skips."""
    m = (config.getoption("-m") or "").strip()
    return m == "gpu" or (m.startswith("gpu") and "not gpu" not in m)


@pytest.fixture(scope="session")
def ref(request):
    """The unmodified reference compiled into oracle/_ref (see oracle/Makefile).  CPU tier: skip if it has not been built."""
    from tests import refapi
    try:
        return refapi.Ref()
    except OSError as e:
        if _gpu_tier(request.config):
            pytest.fail(f"-m gpu needs the oracle (oracle/_ref): {e}")
        pytest.skip(f"oracle/_ref not built: {e}")


@pytest.fixture(scope="session")
def engine(request):
    import torch
    if not torch.cuda.is_available():
        if _gpu_tier(request.config):
            pytest.fail("-m gpu needs a GPU")
        pytest.skip("no GPU")
    from secp256k1_zkp_amd import Engine
    return Engine(0)
