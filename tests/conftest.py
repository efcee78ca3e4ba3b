import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "gpu_soak: further cases of off-by-default experiment paths; they need a GPU and only run under -m gpu_soak")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` runs one representative case per experiment path; the rest of their parametrisations carry gpu_soak as well and are
    left out unless the mark expression names gpu_soak (`-m gpu_soak`) -- the GPU suite has a time budget."""
    if "gpu_soak" in (config.getoption("-m") or ""):
        return
    keep, drop = [], []
    for it in items:
        (drop if it.get_closest_marker("gpu_soak") else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
