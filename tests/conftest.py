import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "musicgeneration_vae-torch_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _seed_per_test(request):
    """every test draws its random inputs from a seed derived from its own id: results do not depend on which tests
    ran before it (a borderline tolerance then fails always or never, not once in a while)"""
    import zlib
    import torch
    torch.manual_seed(zlib.crc32(request.node.nodeid.encode()) & 0x7FFFFFFF)
    yield
