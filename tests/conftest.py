import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.dirname(os.path.abspath(__file__)) not in sys.path:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a machine without a GPU must FAIL loudly rather than pass on nothing; only plain runs skip.
    if _has_gpu():
        return
    markexpr = config.getoption("-m") or ""
    if "gpu" in markexpr and "not gpu" not in markexpr:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


def swap_backend(module, backend):
    """Test seam: rebind a product module's kernel backend (`mixq_amd.linear / fused / eetq._backend`, always the HIP `mixlib` in the product)
    to `backend` - tests/backend_oracle.py on machines without a GPU - and return the previous one.  The product has no API for this."""
    prev = module._backend
    module._backend = backend
    if hasattr(module, "_packed"):
        module._packed.clear()                            # (eetq: packed-weight cache keyed by addresses of the other backend's tensors)
    return prev
