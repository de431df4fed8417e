import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A `-m gpu` run on a box without a HIP device must fail loudly, not pass on skips or on some CPU fallback: if GPU tests
    were selected and there is no device, stop the session with an error before any of them runs."""
    expr = config.getoption("-m") or ""
    wants_gpu = any(it.get_closest_marker("gpu") is not None for it in items)
    if wants_gpu and "gpu" in expr and "not gpu" not in expr:
        import torch
        if not torch.cuda.is_available():
            raise pytest.UsageError("pytest -m gpu needs a HIP device (torch.cuda.is_available() is False): "
                                    "the OFQ MI355X path has no CPU fallback")
