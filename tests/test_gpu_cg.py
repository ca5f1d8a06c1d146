"""CG solver kernels (the Newton kernel without its Hessian) against the oracle: config 3's scene under both cone types,
limit-row chains beyond 16 dofs, and the hand (4 rows per lane)."""
import os

import numpy as np
import pytest

from mujoco_ros_pkgs_amd import mjcf

pytestmark = pytest.mark.gpu


def _parity(oracle_built, m, qpos, qvel, nsteps, tol, tol_acc=None):
    from mujoco_ros_pkgs_amd import engine
    cm = engine.CompiledModel(m)
    nenv = qpos.shape[0]
    b = engine.Batch(cm, nenv)
    b.set("qpos", qpos)
    b.set("qvel", qvel)
    b.set_keep_frame(True)
    b.step(nsteps)
    d = oracle_built.OracleData(m)
    rows = 0
    for e in range(nenv):
        d.reset()
        d.qpos[:] = qpos[e]
        d.qvel[:] = qvel[e]
        for _ in range(nsteps):
            d.step(1)
        nefc = int(d.field("nefc")[0])
        rows += nefc
        assert int(b.get("nefc")[e][0]) == nefc
        for f in ("efc_force", "qacc", "qpos", "qvel"):
            ref = np.asarray(d.field(f))
            k = nefc if f.startswith("efc_") else len(ref)
            if k:
                t = tol_acc if (tol_acc and f in ("efc_force", "qacc")) else (100 * tol if (tol_acc and f == "qvel") else tol)
                np.testing.assert_allclose(b.get(f)[e][:k], ref[:k], rtol=0, atol=t * (1 + np.abs(ref[:k]).max()), err_msg=f"{f} env {e}")
    b.close()
    assert rows > 0


@pytest.mark.parametrize("cone", ["pyramidal", "elliptic"])
def test_cg_on_arm_table_cube(oracle_built, cone):
    from test_gpu_contact import scenario_states
    m = mjcf.compile_xml_file(os.path.join(mjcf.ASSET_DIR, "franka_table.xml"), override={"solver": "CG", "cone": cone})
    assert m["solver"] == 1
    qpos, qvel = scenario_states(m, 6, seed=5)
    # (with the hand box resting on the cube the problem is ill-conditioned: a first-order method stopped at a cost improvement of
    #  1e-8 leaves qacc ~1e-2 from the optimum, so two correctly rounded runs that stop one iteration apart differ by that much;
    #  the integrated state still agrees to 1e-6 / 1e-4)
    _parity(oracle_built, m, qpos, qvel, 3, 1e-6, tol_acc=5e-2)


def test_cg_on_limit_chain(oracle_built):
    from mujoco_ros_pkgs_amd import engine
    from test_gpu_solver_sizes import limited_chain_xml
    m = mjcf.compile_xml_string(limited_chain_xml(24, "CG"))
    rng = np.random.default_rng(24)
    qpos = rng.uniform(-0.45, 0.45, (4, m["nq"]))
    qvel = rng.uniform(-0.5, 0.5, (4, m["nv"]))
    _parity(oracle_built, m, qpos, qvel, 5, 1e-6)
    # same number of CG iterations as the oracle (readable after a forward pass; a full step reuses the slot)
    b = engine.Batch(engine.CompiledModel(m), 4)
    b.set("qpos", qpos)
    b.set("qvel", qvel)
    b.forward()
    d = oracle_built.OracleData(m)
    for e in range(4):
        d.reset()
        d.qpos[:] = qpos[e]
        d.qvel[:] = qvel[e]
        d.forward()
        assert int(b.get("solver_iter")[e][0]) == int(d.field("solver_iter")[0]) > 2
    b.close()


def test_cg_on_the_hand(oracle_built):
    from mujoco_ros_pkgs_amd import workloads
    m = mjcf.compile_xml_file(os.path.join(mjcf.ASSET_DIR, "shadow_hand_like.xml"), override={"solver": "CG"})
    qpos, qvel = workloads.hand_grasp_states(m, 3, seed=5)
    _parity(oracle_built, m, qpos, qvel, 2, 1e-5)
