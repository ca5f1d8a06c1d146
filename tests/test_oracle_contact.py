"""Pins the oracle's contact / constraint path ("parity unpinned" vs MuJoCo, see oracle/mjo.h) with the substitute
oracles of SURVEY.md §8c: (v) collision known answers, (iv) constraint KKT residuals, plus physical facts
(a resting box carries its weight, penetration stays at the soft-constraint scale)."""
import numpy as np
import pytest

from mujoco_ros_pkgs_amd import mjcf

SCENE = """
<mujoco><compiler angle="radian"/>
<option timestep="0.001" cone="pyramidal" solver="PGS" iterations="100" tolerance="1e-10"/>
<worldbody>
  <geom name="floor" type="plane" size="1 1 0.1"/>
  <body name="obj" pos="0 0 {z}">{joint}<geom name="g" type="{gtype}" size="{size}" mass="0.5" {extra}/></body>
  {more}
</worldbody></mujoco>
"""


def scene(gtype, size, z, joint="<freejoint/>", extra="", more=""):
    return mjcf.compile_xml_string(SCENE.format(gtype=gtype, size=size, z=z, joint=joint, extra=extra, more=more))


def contacts(d):
    n = d.ncon[0]
    return (n, d.contact_dist[:n].copy(), d.contact_pos.reshape(-1, 3)[:n].copy(),
            d.contact_frame.reshape(-1, 9)[:n].copy())


def test_plane_sphere_known_answer(oracle_built):
    m = scene("sphere", "0.1", 0.08)
    d = oracle_built.OracleData(m)
    d.forward()
    n, dist, pos, frame = contacts(d)
    assert n == 1 and abs(dist[0] - (-0.02)) < 1e-15
    np.testing.assert_allclose(pos[0], [0, 0, -0.01], atol=1e-15)  # midway between the surfaces
    np.testing.assert_allclose(frame[0][:3], [0, 0, 1], atol=0)
    R = frame[0].reshape(3, 3)
    np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-15)
    # above the margin: no contact
    m2 = scene("sphere", "0.1", 0.11)
    d2 = oracle_built.OracleData(m2)
    d2.forward()
    assert d2.ncon[0] == 0 and d2.nefc[0] == 0


def test_plane_box_four_corners(oracle_built):
    m = scene("box", "0.1 0.2 0.05", 0.049)
    d = oracle_built.OracleData(m)
    d.forward()
    n, dist, pos, frame = contacts(d)
    assert n == 4 and np.allclose(dist, -0.001, atol=1e-15)
    assert sorted(map(tuple, np.round(np.abs(pos[:, :2]), 12))) == [(0.1, 0.2)] * 4
    assert d.nefc[0] == 16 and set(d.efc_type[:16]) == {6}  # 4 pyramidal rows per contact
    # tilted box: only the lowest corner(s)
    d.qpos[3:7] = [np.cos(0.2), np.sin(0.2), 0, 0]
    d.qpos[2] = 0.2
    d.forward()
    assert d.ncon[0] == 0


def test_plane_capsule_and_sphere_sphere(oracle_built):
    m = scene("capsule", "0.05 0.2", 0.04, extra='euler="0 1.5707963267948966 0"')
    d = oracle_built.OracleData(m)
    d.forward()
    n, dist, pos, frame = contacts(d)
    assert n == 2 and np.allclose(dist, -0.01, atol=1e-12)
    np.testing.assert_allclose(sorted(pos[:, 0]), [-0.2, 0.2], atol=1e-12)
    np.testing.assert_allclose(np.abs(frame[:, 3:6] @ np.array([1.0, 0, 0])), 1, atol=1e-12)  # tangent along the axis
    more = '<body name="o2" pos="0.15 0 0.5"><freejoint/><geom type="sphere" size="0.1" mass="1"/></body>'
    m2 = scene("sphere", "0.1", 0.5, more=more)
    d2 = oracle_built.OracleData(m2)
    d2.forward()
    n, dist, pos, frame = contacts(d2)
    assert n == 1 and abs(dist[0] + 0.05) < 1e-15
    np.testing.assert_allclose(frame[0][:3], [1, 0, 0], atol=1e-15)  # from geom1 to geom2
    np.testing.assert_allclose(pos[0], [0.075, 0, 0.5], atol=1e-15)


def test_sphere_box_and_capsule_capsule(oracle_built):
    more = '<body name="o2" pos="0.0 0 0.72"><freejoint/><geom type="sphere" size="0.05" mass="1"/></body>'
    m = scene("box", "0.1 0.1 0.2", 0.5, more=more)
    d = oracle_built.OracleData(m)
    d.forward()
    n, dist, pos, frame = contacts(d)
    # pair order is canonical (sphere, box): normal points from the sphere into the box = -z
    assert n == 1 and abs(dist[0] - (-0.03)) < 1e-15
    np.testing.assert_allclose(frame[0][:3], [0, 0, -1], atol=1e-15)
    more = ('<body name="o2" pos="0 0.08 0.5"><freejoint/><geom type="capsule" size="0.05 0.2" mass="1" '
            'euler="1.5707963267948966 0 0"/></body>')
    m2 = scene("capsule", "0.05 0.2", 0.5, extra='euler="0 1.5707963267948966 0"', more=more)
    d2 = oracle_built.OracleData(m2)
    d2.forward()
    n, dist, pos, frame = contacts(d2)
    # crossed capsules (x-axis through origin, y-axis centred at y=0.08): axes intersect -> full overlap depth
    assert n == 1 and abs(dist[0] + 0.1) < 1e-12


def test_resting_box_carries_its_weight_and_kkt(oracle_built):
    m = scene("box", "0.05 0.05 0.05", 0.05)
    d = oracle_built.OracleData(m)
    for _ in range(600):
        d.step()
    assert abs(d.qvel).max() < 1e-4 and 0.0495 < d.qpos[2] < 0.05  # settled, penetration < 0.5 mm
    nefc = d.nefc[0]
    assert d.ncon[0] == 4 and nefc == 16
    f = d.efc_force[:nefc]
    # every pyramidal row contributes its force to the normal direction: sum = m g
    assert abs(f.sum() - 0.5 * 9.81) < 1e-4
    # dual KKT: f >= 0, residual AR f + b >= 0 where f == 0, == 0 where f > 0
    AR = d.efc_AR.reshape(m["nefcmax"], m["nefcmax"])[:nefc, :nefc]
    res = AR @ f + d.efc_b[:nefc]
    assert f.min() >= 0 and res.min() > -1e-3 and np.abs(res * f).max() < 1e-3
    np.testing.assert_allclose(AR, AR.T, atol=0)
    # B rows are M^-1 J'
    J = d.efc_J.reshape(m["nefcmax"], m["nv"])[:nefc]
    B = d.efc_B.reshape(m["nefcmax"], m["nv"])[:nefc]
    np.testing.assert_allclose(J @ B.T + np.diag(d.efc_R[:nefc]), AR, atol=1e-10)
    np.testing.assert_allclose(d.qfrc_constraint, J.T @ f, atol=1e-12)


def test_joint_limit_row(oracle_built):
    xml = """<mujoco><compiler angle="radian"/><option timestep="0.001" solver="PGS" cone="pyramidal"/>
    <worldbody><body name="l" pos="0 0 1"><joint name="h" type="hinge" axis="0 1 0" limited="true" range="-0.5 0.5"/>
    <inertial pos="0 0 -0.5" mass="1" diaginertia="0.02 0.02 0.01"/></body></worldbody></mujoco>"""
    m = mjcf.compile_xml_string(xml)
    assert m["nefcmax"] == 1
    d = oracle_built.OracleData(m)
    d.qpos[0] = 0.52  # beyond the upper limit
    d.forward()
    assert d.nefc[0] == 1 and d.efc_type[0] == 3 and abs(d.efc_pos[0] + 0.02) < 1e-15 and d.efc_J[0] == -1
    assert d.efc_force[0] > 0 and d.qfrc_constraint[0] < 0  # pushes back towards the range
    d.qpos[0] = -0.6
    d.forward()
    assert d.nefc[0] == 1 and d.efc_J[0] == 1 and d.qfrc_constraint[0] > 0
    d.qpos[0] = 0.1
    d.forward()
    assert d.nefc[0] == 0 and np.array_equal(d.qacc, d.qacc_smooth)
    # released inside the range it swings and is caught by the limits: |q| never exceeds range + a few mrad
    d.reset()
    d.qpos[0] = 0.4
    qmax = 0
    for _ in range(3000):
        d.step()
        qmax = max(qmax, abs(d.qpos[0]))
    assert qmax < 0.52
