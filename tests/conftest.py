import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ref():
    """The unmodified reference compiled into oracle/_ref (see oracle/Makefile); skip if it has not been built."""
    from tests import refapi
    try:
        return refapi.Ref()
    except OSError as e:
        pytest.skip(f"oracle/_ref not built: {e}")


@pytest.fixture(scope="session")
def engine():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from secp256k1_zkp_amd import Engine
    return Engine(0)
