import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available() -> bool:
    try:
        dvp = importlib.import_module("dv-pari_amd")
        return dvp.lib.dvp_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # -m gpu on a box without a GPU must fail loudly, not skip silently; plain runs skip gpu tests
    if config.getoption("-m") and "gpu" in config.getoption("-m") and "not gpu" not in config.getoption("-m"):
        return
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def dvp():
    """The product package (loads libdvpari_hip.so; raises if it is missing)."""
    return importlib.import_module("dv-pari_amd")


@pytest.fixture(scope="session")
def nat():
    return importlib.import_module("dv-pari_amd._native")
