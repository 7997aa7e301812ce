import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "3d-re-gen_amd")
for p in (ROOT, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "skimage: needs the build container's conda scikit-image")
    config.addinivalue_line("markers", "slow: long-running CPU test")


def pytest_collection_modifyitems(config, items):
    from oracle import mc_skimage
    have_sk = mc_skimage.available()
    have_gpu = False
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        pass
    for item in items:
        if "skimage" in item.keywords and not have_sk:
            item.add_marker(pytest.mark.skip(reason="conda scikit-image not present (GPU box)"))
        if "gpu" in item.keywords and not have_gpu:
            item.add_marker(pytest.mark.skip(reason="no GPU visible"))


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
