import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "vit-unet_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def attn_form():
    """Set the process-level re-attention form for one test (vu_set_attn_form) and restore the default afterwards."""
    from vit_unet.torch import _lib

    def setter(flash=-1, centered=0):
        _lib.set_attn_form(flash, centered)
    yield setter
    _lib.set_attn_form(-1, 0)
