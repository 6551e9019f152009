import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def dev():
    import torch
    from visual_odometry_amd import _lib
    _lib.require_device(0)            # raises when the HIP library or the GPU is missing
    torch.cuda.set_device(0)
    return torch.device("cuda:0")


@pytest.fixture()
def det_math(oracle_mod):
    """Oracle in det-math mode (elementary functions from IEEE + - * / sqrt only, the same
    formulas as the device solvers) for bit-exact solver comparisons; restored afterwards."""
    oracle_mod.set_det_math(True)
    yield oracle_mod
    oracle_mod.set_det_math(False)
