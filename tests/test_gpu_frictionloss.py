"""Dry joint friction rows on the GPU against the oracle: hinge chains with frictionloss on every joint (some also
beyond a joint limit), under PGS and Newton, for the nv <= 16 and nv > 16 code paths."""
import numpy as np
import pytest

from mujoco_ros_pkgs_amd import mjcf


def friction_chain_xml(n, solver):
    def body(i):
        ax = ["1 0 0", "0 1 0", "0 0 1"][i % 3]
        return (f'<body name="b{i}" pos="0.05 0.01 -0.04"><joint name="j{i}" type="hinge" axis="{ax}" damping="0.02" '
                f'armature="0.001" limited="true" range="-0.4 0.4" frictionloss="{0.002 * (1 + i % 4)}"/>'
                f'<geom type="capsule" fromto="0 0 0 0.05 0.01 -0.04" size="0.01" mass="0.05" contype="0" conaffinity="0"/>')
    s = "".join(body(i) for i in range(n)) + "</body>" * n
    return f'<mujoco><option timestep="0.001" solver="{solver}" cone="pyramidal"/><worldbody>{s}</worldbody></mujoco>'


@pytest.mark.gpu
@pytest.mark.parametrize("solver,n", [("PGS", 12), ("PGS", 24), ("Newton", 12), ("Newton", 24)])
def test_friction_rows_match_oracle(oracle_built, solver, n):
    from mujoco_ros_pkgs_amd import engine
    m = mjcf.compile_xml_string(friction_chain_xml(n, solver))
    assert m["nefcmax"] == 2 * n
    cm = engine.CompiledModel(m)
    nenv = 4
    rng = np.random.default_rng(n)
    qpos = rng.uniform(-0.5, 0.5, (nenv, m["nq"]))
    qvel = rng.uniform(-0.5, 0.5, (nenv, m["nv"]))
    qvel[1] = 0  # one env starts at rest: some rows stay in the quadratic (sticking) zone
    b = engine.Batch(cm, nenv)
    b.set("qpos", qpos)
    b.set("qvel", qvel)
    b.forward()
    d = oracle_built.OracleData(m)
    for e in range(nenv):
        d.reset()
        d.qpos[:] = qpos[e]
        d.qvel[:] = qvel[e]
        d.forward()
        nefc = int(d.field("nefc")[0])
        assert nefc >= n and int(b.get("nefc")[e][0]) == nefc
        np.testing.assert_array_equal(b.get("efc_type")[e][:nefc], np.asarray(d.field("efc_type"))[:nefc])
        np.testing.assert_array_equal(b.get("efc_frictionloss")[e][:nefc], np.asarray(d.field("efc_frictionloss"))[:nefc])
        ff = np.asarray(d.field("efc_force"))[:n]
        assert np.all(np.abs(ff) <= np.asarray(d.field("efc_frictionloss"))[:n] + 1e-12)
        for f, tol in (("efc_R", 1e-12), ("efc_aref", 1e-9), ("efc_force", 1e-6), ("qacc", 1e-6)):
            ref = np.asarray(d.field(f))
            k = nefc if f.startswith("efc_") else len(ref)
            np.testing.assert_allclose(b.get(f)[e][:k], ref[:k], rtol=0, atol=tol * (1 + np.abs(ref[:k]).max()), err_msg=f"{f} env {e}")
    b.step(30)
    oq, ov, _ = oracle_built.rollout(m, qpos, qvel, 30)
    np.testing.assert_allclose(b.get("qpos"), oq, rtol=0, atol=1e-6)
    np.testing.assert_allclose(b.get("qvel"), ov, rtol=0, atol=1e-5)
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["PGS", "Newton"])
def test_tendon_friction_rows_match_oracle(oracle_built, solver):
    from mujoco_ros_pkgs_amd import engine
    xml = friction_chain_xml(12, solver).replace(
        "</worldbody>", "</worldbody><tendon>"
        '<fixed name="t0" frictionloss="0.004"><joint joint="j1" coef="1"/><joint joint="j2" coef="-0.5"/></fixed>'
        '<fixed name="t1" frictionloss="0.0"><joint joint="j3" coef="1"/></fixed>'
        '<fixed name="t2" frictionloss="0.003" limited="true" range="-0.2 0.2"><joint joint="j4" coef="1"/><joint joint="j7" coef="1"/></fixed>'
        "</tendon>")
    m = mjcf.compile_xml_string(xml)
    assert m["ntendon"] == 3 and m["nefcmax"] == 2 * 12 + 2 + 1
    cm = engine.CompiledModel(m)
    nenv = 3
    rng = np.random.default_rng(5)
    qpos = rng.uniform(-0.5, 0.5, (nenv, m["nq"]))
    qvel = rng.uniform(-0.5, 0.5, (nenv, m["nv"]))
    b = engine.Batch(cm, nenv)
    b.set("qpos", qpos)
    b.set("qvel", qvel)
    b.forward()
    d = oracle_built.OracleData(m)
    for e in range(nenv):
        d.reset()
        d.qpos[:] = qpos[e]
        d.qvel[:] = qvel[e]
        d.forward()
        nefc = int(d.field("nefc")[0])
        assert int(b.get("nefc")[e][0]) == nefc
        types = np.asarray(d.field("efc_type"))[:nefc]
        assert list(types[:14]) == [1] * 12 + [2] * 2  # dof rows, then the two tendons with frictionloss > 0
        np.testing.assert_array_equal(b.get("efc_type")[e][:nefc], types)
        np.testing.assert_array_equal(b.get("efc_id")[e][:nefc], np.asarray(d.field("efc_id"))[:nefc])
        for f, tol in (("efc_J", 1e-12), ("efc_R", 1e-12), ("efc_frictionloss", 0), ("efc_force", 1e-6), ("qacc", 1e-6)):
            ref = np.asarray(d.field(f))
            k = nefc * m["nv"] if f == "efc_J" else (nefc if f.startswith("efc_") else len(ref))
            np.testing.assert_allclose(b.get(f)[e][:k], ref[:k], rtol=0, atol=tol * (1 + np.abs(ref[:k]).max()), err_msg=f"{f} env {e}")
    b.step(30)
    oq, ov, _ = oracle_built.rollout(m, qpos, qvel, 30)
    np.testing.assert_allclose(b.get("qpos"), oq, rtol=0, atol=1e-6)
    b.close()
