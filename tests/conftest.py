import os
import sys
import warnings

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.dirname(__file__)):
    if p not in sys.path:
        sys.path.insert(0, p)
warnings.filterwarnings("ignore", category=FutureWarning)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs the read-only reference tree at /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    import torch
    have_gpu = torch.cuda.is_available()
    from oracle import refshim
    have_ref = refshim.available()
    for item in items:
        if "gpu" in item.keywords and not have_gpu:
            item.add_marker(pytest.mark.skip(reason="no GPU visible"))
        if "reference" in item.keywords and not have_ref:
            item.add_marker(pytest.mark.skip(reason="reference tree not present"))


@pytest.fixture(scope="session")
def hip_lib():
    """The C-ABI library (compiled on demand; hipcc cross-compiles gfx950 without a GPU)."""
    from geneface_amd.csrc import build
    build.build()
    from geneface_amd import lib
    return lib.lib()


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import build, kernels
    build.build()
    return kernels.lib()
