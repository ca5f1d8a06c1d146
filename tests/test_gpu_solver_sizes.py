"""Solver paths by system size: nv <= 16 (register-resident factor / J M^-1 rows), 17..32 (generic factor, Cholesky with
one Hessian row per lane in registers) and > 32 (LDS Cholesky), under both PGS and Newton, on hinge chains whose
joint limits are violated at the start (so the rows are active and the forces non-zero)."""
import numpy as np
import pytest

from mujoco_ros_pkgs_amd import mjcf


def limited_chain_xml(n, solver):
    def body(i):
        ax = ["1 0 0", "0 1 0", "0 0 1"][i % 3]
        return (f'<body name="b{i}" pos="0.05 0.01 -0.04"><joint name="j{i}" type="hinge" axis="{ax}" damping="0.02" '
                f'armature="0.001" limited="true" range="-0.3 0.3"/><geom type="capsule" fromto="0 0 0 0.05 0.01 -0.04" '
                f'size="0.01" mass="0.05" contype="0" conaffinity="0"/>')
    s = "".join(body(i) for i in range(n)) + "</body>" * n
    return (f'<mujoco><option timestep="0.001" solver="{solver}" cone="pyramidal"/><worldbody>{s}</worldbody></mujoco>')


@pytest.mark.gpu
# (Newton 17 / 32 / 33: the packed Hessian at its smallest -- chol_schur16's 272-double scratch is larger than the 154 doubles of the triangle --, at its largest, and the first size on the full nv x nv layout)
@pytest.mark.parametrize("solver,n", [("PGS", 12), ("PGS", 24), ("Newton", 12), ("Newton", 17), ("Newton", 24), ("Newton", 32), ("Newton", 33), ("Newton", 40)])
def test_limit_rows_match_oracle(oracle_built, solver, n):
    from mujoco_ros_pkgs_amd import engine
    m = mjcf.compile_xml_string(limited_chain_xml(n, solver))
    assert m["nv"] == n and m["nefcmax"] >= n
    cm = engine.CompiledModel(m)
    nenv = 4
    rng = np.random.default_rng(n)
    qpos = rng.uniform(-0.45, 0.45, (nenv, m["nq"]))  # about a third of the joints start beyond a limit
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
        assert nefc > 0 and int(b.get("nefc")[e][0]) == nefc
        for f, tol in (("qacc_smooth", 1e-9), ("efc_J", 1e-12), ("efc_aref", 1e-9), ("efc_force", 1e-6), ("qacc", 1e-6)):
            ref = np.asarray(d.field(f))
            got = b.get(f)[e]
            k = nefc * n if f == "efc_J" else (nefc if f.startswith("efc_") else len(ref))
            np.testing.assert_allclose(got[:k], ref[:k], rtol=0, atol=tol * (1 + np.abs(ref[:k]).max()), err_msg=f"{f} env {e}")
    b.step(20)
    oq, ov, _ = oracle_built.rollout(m, qpos, qvel, 20)
    np.testing.assert_allclose(b.get("qpos"), oq, rtol=0, atol=1e-7)
    np.testing.assert_allclose(b.get("qvel"), ov, rtol=0, atol=1e-5)
    b.close()
