"""pytest wiring: adds the host-side package root (the counterpart of the reference's `src/`,
cf. /root/reference/tests/conftest.py:9-12) to sys.path and registers the `gpu` marker."""
import os
import sys

import pytest

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_ROOT = os.path.join(REPO_ROOT, "flashattention-pytorch_amd")
for p in (PKG_ROOT, REPO_ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def device():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return "cuda"
