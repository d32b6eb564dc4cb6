import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "whisper-vits-svc_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "needs_reference: needs /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isdir("/root/reference/vits")
    skip_ref = pytest.mark.skip(reason="/root/reference not present")
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:       # noqa: BLE001
        have_gpu = False
    skip_gpu = pytest.mark.skip(reason="no GPU visible (run on the MI355X box: pytest -m gpu)")
    for item in items:
        if "needs_reference" in item.keywords and not have_ref:
            item.add_marker(skip_ref)
        if "gpu" in item.keywords and not have_gpu:
            item.add_marker(skip_gpu)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
