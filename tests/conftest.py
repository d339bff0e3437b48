import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    # Some GPU tests hand torch tensors (device memory) to the engine.  torch bundles its own HIP runtime;
    # it has to be the first one initialised in the process, otherwise torch later reports "No HIP GPUs".
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


@pytest.fixture(scope="session")
def oracle_built():
    import support
    support.oracle_lib()
    return True
