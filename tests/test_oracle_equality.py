"""Equality constraints (connect / weld / joint) of the CPU oracle -- "parity unpinned" (MuJoCo absent), pinned by
(i) the rows' Jacobians being the exact derivative of their residuals (finite differences along random velocity
directions, which also checks the weld's quaternion correction term), and (ii) physical behaviour: a box hung from
a pendulum by a connect stays attached, a welded free body holds its pose under gravity, a joint equality tracks
its polynomial."""
import numpy as np
import pytest

from mujoco_ros_pkgs_amd import mjcf

XML = """
<mujoco><compiler angle="radian"/>
<option timestep="0.001" cone="{cone}" solver="{solver}" iterations="100" tolerance="1e-10"><flag contact="disable"/></option>
<worldbody>
  <body name="a" pos="0 0 1"><joint name="ja" type="hinge" axis="0 1 0" damping="0.2"/><geom type="capsule" fromto="0 0 0 0.3 0 0" size="0.02" mass="0.5"/>
    <body name="b" pos="0.3 0 0"><joint name="jb" type="hinge" axis="0 1 0" damping="0.2"/><geom type="capsule" fromto="0 0 0 0.3 0 0" size="0.02" mass="0.5"/></body>
  </body>
  <body name="c" pos="0.6 0 0.9"><freejoint name="jc"/><geom type="box" size="0.05 0.05 0.05" mass="0.3"/></body>
  <body name="w" pos="-0.5 0.2 0.7" quat="0.9 0.1 0.3 0.2"><freejoint name="jw"/><geom type="box" size="0.05 0.04 0.03" mass="0.4"/></body>
  <body name="p" pos="0 1 1"><joint name="jp" type="hinge" axis="0 1 0" damping="0.05"/><geom type="capsule" fromto="0 0 0 0.2 0 0" size="0.02" mass="0.2"/></body>
  <body name="q" pos="0 1.5 1"><joint name="jq" type="hinge" axis="0 1 0" damping="0.05"/><geom type="capsule" fromto="0 0 0 0.2 0 0" size="0.02" mass="0.2"/></body>
</worldbody>
<equality>
  <connect name="hang" body1="b" body2="c" anchor="0.3 0 0"/>
  <weld name="hold" body1="w" torquescale="0.8" anchor="-0.5 0.2 0.7"/>
  <weld name="rel" body1="a" body2="w" relpose="0.1 0.2 0.3 0.8 0.2 -0.4 0.4" anchor="0.05 0 0.02" active="{relactive}"/>
  <joint name="gear" joint1="jq" joint2="jp" polycoef="0.1 0.5 0.2 0 0"/>
</equality>
</mujoco>
"""


def model(solver="Newton", cone="elliptic", relactive="false"):
    return mjcf.compile_xml_string(XML.format(solver=solver, cone=cone, relactive=relactive))


def _integrate(m, q, v, eps):
    """qpos (+) eps * v with quaternion handling for free joints (first-order exact for the test)."""
    q = q.copy()
    for j in range(m["njnt"]):
        qa, da, t = m["jnt_qposadr"][j], m["jnt_dofadr"][j], m["jnt_type"][j]
        if t == 0:
            q[qa:qa + 3] += eps * v[da:da + 3]
            w = v[da + 3:da + 6] * eps
            ang = np.linalg.norm(w)
            dq = np.array([1.0, 0, 0, 0]) if ang < 1e-300 else np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * w / ang])
            q[qa + 3:qa + 7] = mjcf.quat_mul(q[qa + 3:qa + 7], dq)   # body-frame angular velocity
        else:
            q[qa] += eps * v[da]
    return q


@pytest.mark.parametrize("relactive", ["false", "true"])
def test_equality_jacobian_is_the_derivative_of_the_residual(oracle_built, relactive):
    m = model(relactive=relactive)
    nrow = 3 + 6 + (6 if relactive == "true" else 0) + 1
    rng = np.random.default_rng(0)
    d = oracle_built.OracleData(m)
    for trial in range(5):
        q0 = np.array(m["qpos0"], dtype=np.float64)
        q0 = _integrate(m, q0, rng.normal(size=m["nv"]), 0.3)
        d.qpos[:] = q0
        d.qvel[:] = 0
        d.forward()
        assert d.nefc[0] >= nrow and np.all(d.efc_type[:nrow] == 0)
        J = np.array(d.efc_J).reshape(-1, m["nv"])[:nrow].copy()
        v = rng.normal(size=m["nv"])
        eps = 1e-6
        res = []
        for sgn in (+1, -1):
            d.qpos[:] = _integrate(m, q0, v, sgn * eps)
            d.forward()
            res.append(np.array(d.efc_pos)[:nrow].copy())
        fd = (res[0] - res[1]) / (2 * eps)
        np.testing.assert_allclose(J @ v, fd, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("solver,cone", [("Newton", "elliptic"), ("PGS", "pyramidal")])
def test_equality_behaviour(oracle_built, solver, cone):
    m = model(solver, cone)
    d = oracle_built.OracleData(m)
    w0 = np.array(d.qpos[m["jnt_qposadr"][m.name2id("joint", "jw")]:][:7]).copy()
    d.step(1500)
    assert np.all(np.isfinite(d.qpos))
    nrow = 3 + 6 + 1
    assert d.nefc[0] >= nrow
    res = np.array(d.efc_pos)[:nrow]
    # connect: the box hangs from the pendulum tip (soft constraint: millimetres under 0.3 kg)
    assert np.linalg.norm(res[:3]) < 5e-3
    qc = m["jnt_qposadr"][m.name2id("joint", "jc")]
    assert d.qpos[qc + 2] > 0.2, "the connected box fell"
    # weld to the world: pose held
    qw = m["jnt_qposadr"][m.name2id("joint", "jw")]
    assert np.linalg.norm(np.array(d.qpos[qw:qw + 3]) - w0[:3]) < 5e-3
    assert abs(abs(np.dot(np.array(d.qpos[qw + 3:qw + 7]), w0[3:])) - 1) < 1e-4
    # joint equality: jq - 0 = 0.1 + 0.5 x + 0.2 x^2 with x = jp
    x = d.qpos[m["jnt_qposadr"][m.name2id("joint", "jp")]]
    y = d.qpos[m["jnt_qposadr"][m.name2id("joint", "jq")]]
    assert abs(y - (0.1 + 0.5 * x + 0.2 * x * x)) < 5e-3
    # bilateral: some equality force is negative somewhere along the run or now (sign not clamped)
    assert np.any(np.array(d.efc_force)[:nrow] < 0) or np.any(np.array(d.efc_force)[:nrow] > 0)
