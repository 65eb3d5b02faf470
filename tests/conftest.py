import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def g1():
    return load_golden("g1_phase_delay.npz")


@pytest.fixture(scope="session")
def g2():
    return load_golden("g2_predict_vis.npz")


@pytest.fixture(scope="session")
def g3():
    return load_golden("g3_im_to_vis.npz")


@pytest.fixture(scope="session")
def g4():
    return load_golden("g4_beam.npz")


@pytest.fixture(scope="session")
def g5():
    return load_golden("g5_chain_c1.npz")


@pytest.fixture(scope="session")
def g7():
    return load_golden("g7_wsclean.npz")


@pytest.fixture(scope="session")
def g8():
    return load_golden("g8_producers.npz")


@pytest.fixture(scope="session")
def g9():
    return load_golden("g9_calibration.npz")


@pytest.fixture(scope="session")
def g10():
    return load_golden("g10_degridder.npz")


@pytest.fixture(scope="session")
def g11():
    return load_golden("g11_convert.npz")


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
