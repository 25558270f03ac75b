import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        pkg = importlib.import_module("mola-fe-lidar_amd")
        import ctypes
        n = ctypes.c_int(0)
        pkg._lib.lib().mola_icp_device_count(ctypes.byref(n))
        return n.value > 0
    except Exception:
        return False


HAS_GPU = None


def pytest_collection_modifyitems(config, items):
    global HAS_GPU
    if HAS_GPU is None:
        HAS_GPU = _has_gpu()
    if HAS_GPU:
        for it in items:  # a kernel that never finishes must fail the test, not stall the run
            if "gpu" in it.keywords and not any(m.name == "timeout" for m in it.iter_markers()):
                it.add_marker(pytest.mark.timeout(900))
        return
    skip = pytest.mark.skip(reason="no GPU in this container (GPU tests run on the MI355X box)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def pkg():
    """the product package (directory name has a dash, so import by string)"""
    return importlib.import_module("mola-fe-lidar_amd")


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("mola-fe-lidar_amd.synth")


@pytest.fixture(scope="session")
def O():
    """the CPU oracle (test infrastructure)"""
    from oracle import oracle
    oracle.build()
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "icp_golden.npz"))


@pytest.fixture(scope="session")
def small_scene(synth):
    return synth.Scene(scene_seed=5, half=10.0, wall_y=4.0, wall_h=4.0, n_boxes=6)
