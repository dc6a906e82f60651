import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def hip_lib():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as ge
    ge.build()
    from mtd_gan_amd import _lib
    return _lib.lib()
