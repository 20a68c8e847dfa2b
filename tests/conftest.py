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


def low_texture_frame(pkg, seed=6, width=640, height=480):
    """A nearly flat frame with a few low-contrast shapes: the selector keeps fewer than
    num_want / 3 pixels, so the Canny top-up runs (ref src/pcd_generator.cpp:143-175)."""
    bgr, dep = pkg.data.synthetic_rgbd_frame(width=width, height=height, seed=seed, texture=0.0, holes=0.01)
    b = bgr.astype(np.int32)
    b[height // 5:height * 3 // 5, width // 3:width * 2 // 5] += 10
    b[height * 3 // 4:height * 4 // 5, width // 12:width * 11 // 12] -= 10
    b[20:40, width - 140:width - 120] += 60
    return np.clip(b, 0, 255).astype(np.uint8), dep
