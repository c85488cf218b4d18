import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture
def hooks():
    """Set test hooks of the loaded library for the duration of one test (tests/common.py set_opt); every hook set is cleared again."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from common import set_opt
    used = []

    def _set(name, value):
        set_opt(name, value)
        used.append(name)
    yield _set
    for n in used:
        set_opt(n, None)
