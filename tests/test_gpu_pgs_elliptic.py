"""PGS with elliptic cones on the GPU against the oracle (same block algorithm, same arithmetic order): the slab on
a tilted plane (4 corner contacts, sticking and sliding) and the shipped pendulum world switched to PGS."""
import os

import numpy as np
import pytest

from mujoco_ros_pkgs_amd import mjcf
from test_oracle_contact import BOX_ON_PLANE

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _check(oracle_built, m, qpos, qvel, nsteps, tol_force=1e-6):
    from mujoco_ros_pkgs_amd import engine
    cm = engine.CompiledModel(m)
    nenv = qpos.shape[0]
    b = engine.Batch(cm, nenv)
    b.set("qpos", qpos)
    b.set("qvel", qvel)
    b.set_keep_frame(True)  # the last step's solver output stays readable
    b.step(nsteps)
    d = oracle_built.OracleData(m)
    saw_contact = False
    for e in range(nenv):
        d.reset()
        d.qpos[:] = qpos[e]
        d.qvel[:] = qvel[e]
        for s in range(nsteps):
            d.step(1)
        nefc = int(d.field("nefc")[0])
        saw_contact |= nefc > 0
        assert int(b.get("nefc")[e][0]) == nefc
        for f, tol in (("efc_force", tol_force), ("qacc", tol_force), ("qpos", 1e-8), ("qvel", 1e-6)):
            ref = np.asarray(d.field(f))
            k = nefc if f.startswith("efc_") else len(ref)
            if k == 0:
                continue
            np.testing.assert_allclose(b.get(f)[e][:k], ref[:k], rtol=0, atol=tol * (1 + np.abs(ref[:k]).max()), err_msg=f"{f} env {e}")
    b.close()
    return saw_contact


@pytest.mark.gpu
@pytest.mark.parametrize("theta", [0.35, 0.75])
def test_slab_on_tilted_plane(oracle_built, theta):
    g = 9.81
    xml = BOX_ON_PLANE.format(cone="elliptic", gx=g * np.sin(theta), gz=-g * np.cos(theta), mu=0.5).replace(
        'solver="Newton"', 'solver="PGS" iterations="100"')
    m = mjcf.compile_xml_string(xml)
    assert m["solver"] == 0 and m["cone"] == 1
    nenv = 4
    rng = np.random.default_rng(3)
    qpos = np.tile(m["qpos0"], (nenv, 1))
    qpos[:, 2] -= rng.uniform(0, 2e-4, nenv)  # slightly pressed into the plane: contact from the first step
    qvel = np.zeros((nenv, m["nv"]))
    qvel[:, 0:2] = rng.uniform(-0.3, 0.3, (nenv, 2))
    qvel[:, 5] = rng.uniform(-0.5, 0.5, nenv)
    assert _check(oracle_built, m, qpos, qvel, 5)


@pytest.mark.gpu
def test_pendulum_world_with_pgs(oracle_built):
    """The shipped pendulum world (elliptic cones, condim 3) with solver=PGS: ball resting on the plane."""
    xml = open(os.path.join(GOLDEN, "pendulum_world.xml")).read()
    xml = xml.replace("<option", '<option solver="PGS"', 1)
    m = mjcf.compile_xml_string(xml)
    assert m["solver"] == 0 and m["cone"] == 1
    nenv = 3
    rng = np.random.default_rng(4)
    qpos = np.tile(m["qpos0"], (nenv, 1))
    qvel = rng.uniform(-0.2, 0.2, (nenv, m["nv"]))
    assert _check(oracle_built, m, qpos, qvel, 40, tol_force=1e-5)


@pytest.mark.gpu
def test_arm_table_cube_with_pgs_elliptic(oracle_built):
    """BASELINE config 3's model with cone=elliptic under PGS: grasp / rest / push scenarios of the contact tests."""
    from test_gpu_contact import scenario_states
    m = mjcf.compile_xml_file(os.path.join(mjcf.ASSET_DIR, "franka_table.xml"), override={"solver": "PGS", "cone": "elliptic"},
                               nefcmax=57)  # 16 contacts x 3 rows + 9 limits (the elliptic PGS kernel holds <= 64 rows)
    assert m["solver"] == 0 and m["cone"] == 1 and m["nefcmax"] <= 64
    qpos, qvel = scenario_states(m, 6, seed=5)
    assert _check(oracle_built, m, qpos, qvel, 3, tol_force=1e-5)
