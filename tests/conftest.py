import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def raw_arm():
    from mjmpc_amd.models.reacher7dof import reacher7dof_raw
    return reacher7dof_raw()


@pytest.fixture(scope="session")
def ref_arm(raw_arm):
    from oracle.physics_ref import RefArm
    return RefArm(raw_arm.to_flat())
