"""Oracle vs REAL MuJoCo (SURVEY.md §8c last rows / §8d CPU baseline): runs only on a machine whose $MUJOCO_DIR holds a
MuJoCo release (headers + libmujoco.so); everywhere else it reports the literal "MuJoCo parity: NOT MEASURED (library
absent)" and the oracle stays "parity unpinned" (oracle/mjo.h).  Nothing under /root/reference is read."""
import os

import numpy as np
import pytest

from mujoco_ros_pkgs_amd import mjcf
from oracle import mujoco_ref

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
WORLDS = [("asset", "franka_like"), ("asset", "franka_table"), ("asset", "shadow_hand_like"),
          ("golden", "pendulum_world"), ("golden", "equality_world"), ("golden", "sensors_world"), ("golden", "mocap_world")]


def test_absence_is_reported_not_faked():
    if mujoco_ref.available():
        assert mujoco_ref.version()
    else:
        print("MuJoCo parity: " + mujoco_ref.ABSENT)
        assert mujoco_ref.time_reference("franka_like", 1.0) == "NOT MEASURED (library absent)"
        with pytest.raises(RuntimeError, match="library absent"):
            mujoco_ref.load()


@pytest.mark.skipif(not mujoco_ref.available(), reason="MuJoCo parity: NOT MEASURED (library absent)")
@pytest.mark.parametrize("kind,name", WORLDS)
def test_oracle_matches_real_mujoco(oracle_built, kind, name):
    path = mujoco_ref.asset_path(name) if kind == "asset" else os.path.join(GOLDEN, name + ".xml")
    model = mjcf.compile_xml_file(path)
    sim = mujoco_ref.RefSim(path)
    assert (sim.nq, sim.nv, sim.nu, sim.nsensordata) == (model["nq"], model["nv"], model["nu"], model["nsensordata"])
    # the MJCF-subset compiler against mj_loadXML: the constants the step reads
    for fld in ("qpos0", "body_mass", "body_inertia", "body_subtreemass", "dof_invweight0", "body_invweight0", "geom_rbound"):
        ref = sim.model(fld)
        mine = np.asarray(model[fld], dtype=np.float64).reshape(-1)
        assert ref.shape == mine.shape and np.allclose(ref, mine, rtol=1e-9, atol=1e-12), fld
    rng = np.random.default_rng(5)
    d = oracle_built.OracleData(model)
    qpos = np.asarray(model["qpos0"], dtype=np.float64).copy()
    for j in range(model["njnt"]):
        if model["jnt_type"][j] >= 2:
            qpos[model["jnt_qposadr"][j]] += rng.uniform(-0.05, 0.05)
    qvel = rng.uniform(-0.1, 0.1, model["nv"])
    ctrl = rng.uniform(-1, 1, model["nu"])
    d.qpos[:] = qpos
    d.qvel[:] = qvel
    if model["nu"]:
        d.ctrl[:] = ctrl
    sim.reset()
    sim.set_state(qpos, qvel, ctrl)
    sim.forward()
    d.forward()
    tol = dict(rtol=1e-9, atol=1e-10)
    for fld in ("xpos", "xipos", "subtree_com", "cinert", "cdof", "qM", "qLD", "cvel", "qfrc_bias", "qfrc_passive", "qacc_smooth"):
        assert np.allclose(sim.get(fld), getattr(d, fld), **tol), fld
    s = sim.sizes()
    assert s["ncon"] == d.ncon[0] and s["nefc"] == d.nefc[0]
    n = s["nefc"]
    if n:
        for fld in ("efc_pos", "efc_margin", "efc_R", "efc_D", "efc_aref", "efc_vel"):
            assert np.allclose(sim.get(fld), getattr(d, fld)[:n], **tol), fld
        assert np.allclose(sim.get("efc_J"), d.efc_J[:n * model["nv"]], **tol)
        assert np.allclose(sim.get("efc_force"), d.efc_force[:n], rtol=1e-6, atol=1e-8)
    assert np.allclose(sim.get("qacc"), d.qacc, rtol=1e-7, atol=1e-8)
    # one step, then a short rollout
    sim.step(1)
    d.step(1)
    q, v, a, sd = sim.state()
    assert np.allclose(q, d.qpos, **tol) and np.allclose(v, d.qvel, rtol=1e-8, atol=1e-9)
    sim.step(50)
    d.step(50)
    q, v, a, sd = sim.state()
    assert np.allclose(q, d.qpos, rtol=1e-6, atol=1e-7) and np.allclose(v, d.qvel, rtol=1e-5, atol=1e-6)
    assert np.allclose(sd, d.sensordata, rtol=1e-5, atol=1e-6)
