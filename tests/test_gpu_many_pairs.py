"""More candidate pairs than two rounds of 64: the collision stage culls every pair first and packs the survivors into narrow-phase
rounds (mjb_constraint.h, collision) -- three cull rounds, survivors from all of them, contact order = pair order.  Two free
clusters of spheres over a plane (12 x 12 cross pairs + 24 plane pairs = 168), GPU against the oracle."""
import numpy as np
import pytest

from mujoco_ros_pkgs_amd import mjcf

pytestmark = pytest.mark.gpu


def cluster_xml(nper=12, solver="PGS", cone="pyramidal"):
    def cluster(name, z):
        g = "".join(f'<geom name="{name}{k}" type="sphere" size="0.03" pos="{0.07 * (k % 4) - 0.1} {0.07 * (k // 4) - 0.07} 0" '
                    f'mass="0.05"/>' for k in range(nper))
        return f'<body name="{name}" pos="0 0 {z}"><freejoint/>{g}</body>'
    return (f'<mujoco><option timestep="0.002" solver="{solver}" cone="{cone}" iterations="60"/><size nconmax="32" njmax="128"/>'
            f'<worldbody><geom name="floor" type="plane" size="2 2 0.1"/>{cluster("a", 0.028)}{cluster("b", 0.084)}</worldbody></mujoco>')


@pytest.mark.parametrize("solver,cone", [("PGS", "pyramidal"), ("Newton", "elliptic")])
def test_three_cull_rounds_match_oracle(oracle_built, solver, cone):
    from mujoco_ros_pkgs_amd import engine
    m = mjcf.compile_xml_string(cluster_xml(solver=solver, cone=cone))
    assert m["ncollpair"] > 128, m["ncollpair"]
    cm = engine.CompiledModel(m)
    nenv = 8
    rng = np.random.default_rng(3)
    qpos = np.tile(np.asarray(m["qpos0"]).ravel(), (nenv, 1))
    qpos[:, 0:2] += rng.uniform(-0.03, 0.03, (nenv, 2))
    qpos[:, 7:9] += rng.uniform(-0.03, 0.03, (nenv, 2))
    qpos[:, 9] += rng.uniform(-0.02, 0.02, nenv)
    qvel = rng.uniform(-0.1, 0.1, (nenv, m["nv"]))
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
        nc = int(np.asarray(d.field("ncon")).ravel()[0])
        assert nc > 0
        assert int(b.get("ncon")[e].ravel()[0]) == nc
        np.testing.assert_array_equal(np.asarray(b.get("contact_geom")[e]).reshape(-1, 2)[:nc], np.asarray(d.field("contact_geom")).reshape(-1, 2)[:nc])
        np.testing.assert_allclose(np.asarray(b.get("contact_dist")[e]).ravel()[:nc], np.asarray(d.field("contact_dist")).ravel()[:nc], rtol=0, atol=1e-12)
        ref = np.asarray(d.field("qacc")).ravel()
        np.testing.assert_allclose(b.get("qacc")[e], ref, rtol=0, atol=1e-6 * (1 + np.abs(ref).max()), err_msg=f"env {e}: qacc")
    b.step(40)
    oq, ov, _ = oracle_built.rollout(m, qpos, qvel, 40)
    # (PGS on up to 32 redundant contacts -- twelve spheres of one body on a plane -- stops at its iteration cap: forty steps amplify
    #  the last-bit differences of the sweeps; the primal solver converges and stays at 1e-7)
    np.testing.assert_allclose(b.get("qpos"), oq, rtol=0, atol=1e-7 if solver == "Newton" else 1e-5)
    b.close()
