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


def pytest_terminal_summary(terminalreporter):
    """measured parity errors of the GPU tests next to their tolerances (also written to gpurun_out/parity_measured.json)"""
    try:
        import parity_support
    except Exception:
        return
    if not parity_support.MEASURED:
        return
    terminalreporter.write_line("parity (measured / tolerance):")
    for name, v, tol in parity_support.MEASURED:
        terminalreporter.write_line("  %-44s %.3e / %.1e" % (name, v, tol))
    try:
        import json
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_measured.json"), "w") as f:
            json.dump([{"name": n, "value": v, "tol": t} for n, v, t in parity_support.MEASURED], f, indent=1)
    except OSError:
        pass
