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
    """Set the process-level re-attention form for one test (vu_set_attn_form) and restore the default afterwards.
    key_split: the recompute form's split of the streamed axis over wave pairs (vu_set_flash_key_split).  A test that forces
    the form runs the UNSPLIT sweeps (1: the default) unless it asks for 2; 0 = the library's default."""
    from vit_unet.torch import _lib

    def setter(flash=-1, centered=0, key_split=None, pcache=-1):
        _lib.set_attn_form(flash, centered)
        _lib.set_flash_key_split((1 if flash == 1 else 0) if key_split is None else key_split)
        _lib.set_flash_pcache(pcache)               # -1: the build's default (on); 0: every sweep recomputes (the round 2 - 4 form)
    yield setter
    _lib.set_attn_form(-1, 0)
    _lib.set_flash_key_split(0)
    _lib.set_flash_pcache(-1)
