import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # this directory is the block jig (tests/jigs.lua of the reference): as there, a FIRFilterBlock whose use_fft was not given runs
    # the direct form (firfilter.lua:57 `... and not package.loaded['tests.jigs']`)
    import luaradio_amd.block
    luaradio_amd.block.TESTS_JIGS_LOADED = True


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
