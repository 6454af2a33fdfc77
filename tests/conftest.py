import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def shelf_inputs():
    return load_golden("shelf_inputs.npz")


@pytest.fixture(scope="session")
def shelf_spatial():
    return load_golden("shelf_spatial.npz")


@pytest.fixture(scope="session")
def ik_cases():
    return load_golden("ik_cases.npz")


SPATIAL_FRAMES = [1, 50, 100, 131, 150, 200, 220, 295, 300]
