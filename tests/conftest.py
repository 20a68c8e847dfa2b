import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as ge  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def pkg():
    return ge.load_package()


@pytest.fixture(scope="session")
def po():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def desk():
    """The five shipped fr1/desk clouds (tests/golden/desk_pcd_ds.npz)."""
    z = np.load(os.path.join(GOLDEN, "desk_pcd_ds.npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden_json():
    def load(name):
        with open(os.path.join(GOLDEN, name)) as fh:
            return json.load(fh)
    return load
