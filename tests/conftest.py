import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle_built():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def franka():
    from mujoco_ros_pkgs_amd import mjcf
    return mjcf.load_asset("franka_like")


def random_franka_state(model, nenv, seed=0):
    """SURVEY.md §8d synthetic initial state: mid-range pose + U(-0.1,0.1), fingers U(0,0.04), qvel U(-0.1,0.1)."""
    rng = np.random.default_rng(seed)
    nq, nv = model["nq"], model["nv"]
    rngj = np.asarray(model["jnt_range"])
    mid = 0.5 * (rngj[:, 0] + rngj[:, 1])
    qpos = np.tile(np.asarray(model["qpos0"]), (nenv, 1))
    for j in range(model["njnt"]):
        t, qa = model["jnt_type"][j], model["jnt_qposadr"][j]
        if t == 3:
            qpos[:, qa] = mid[j] + rng.uniform(-0.1, 0.1, nenv)
        elif t == 2:
            qpos[:, qa] = rng.uniform(rngj[j, 0], rngj[j, 1], nenv)
    qvel = rng.uniform(-0.1, 0.1, (nenv, nv))
    return qpos, qvel
