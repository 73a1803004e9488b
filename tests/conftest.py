"""pytest configuration: `gpu` marker, import paths, shared fixtures.

  python -m pytest tests -q -m "not gpu"   # CPU: oracle vs goldens, host logic, ABI surface
  python -m pytest tests -q -m gpu         # MI355X: HIP path vs oracle, through the C ABI
"""
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_pkg():
    """import 3dioumatch_amd (not a valid identifier) and expose its drop-in modules."""
    return importlib.import_module("3dioumatch_amd")


def load_synth():
    load_pkg()
    return importlib.import_module("3dioumatch_amd.synth")


@pytest.fixture(scope="session")
def synth():
    return load_synth()


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle(omp=False)


@pytest.fixture(scope="session")
def oracle_omp():
    from oracle.oracle import Oracle
    return Oracle(omp=True)


@pytest.fixture(scope="session")
def ext():
    """The product's pointnet2._ext on the GPU (fails loudly if the .so is missing)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    load_pkg()
    return importlib.import_module("pointnet2._ext")


@pytest.fixture(scope="session")
def iou_ext():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    load_pkg()
    return importlib.import_module("pcdet.ops.iou3d_nms.iou3d_nms_cuda")


def golden(name):
    import numpy as np
    return np.load(os.path.join(GOLDEN, name))
