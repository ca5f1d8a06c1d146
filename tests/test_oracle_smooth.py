"""Pins the CPU oracle (oracle/libmjo.so) -- "parity unpinned" against real MuJoCo (absent), so it
is pinned by the substitute oracles of SURVEY.md §8c: (i) closed form, (ii) independently derived dynamics
(mujoco_ros_pkgs_amd/refdyn.py: Jacobian-based M, finite-differenced Lagrangian bias), (iii) internal
identities, plus the reference-test facts (time advance, equilibrium at rest, reset state)."""
import os

import numpy as np
import pytest

from conftest import random_franka_state
from mujoco_ros_pkgs_amd import mjcf, refdyn

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

PENDULUM_1DOF = """
<mujoco><compiler angle="radian"/><option timestep="0.001" gravity="0 0 -9.81"><flag contact="disable"/></option>
<worldbody><body name="link" pos="0 0 1"><joint name="hinge" type="hinge" axis="0 1 0" damping="{damping}"/>
<inertial pos="0 0 -0.5" mass="2.0" diaginertia="0.05 0.07 0.01"/></body></worldbody></mujoco>
"""


def dense_M(m, qM):
    nv = m["nv"]
    M = np.zeros((nv, nv))
    for i in range(nv):
        adr, j = m["dof_Madr"][i], i
        while j >= 0:
            M[i, j] = M[j, i] = qM[adr]
            adr += 1
            j = m["dof_parentid"][j]
    return M


def test_closed_form_pendulum_step(oracle_built):
    """(i) 1-dof pendulum: the semi-implicit Euler one-step map is algebraic:
    I = Iyy + m l^2,  a = -m g l sin(q) / I,  v' = v + h a,  q' = q + h v'."""
    m = mjcf.compile_xml_string(PENDULUM_1DOF.format(damping=0))
    d = oracle_built.OracleData(m)
    q, v, h = 0.7, -0.3, 0.001
    d.qpos[0], d.qvel[0] = q, v
    d.step()
    I = 0.07 + 2.0 * 0.25
    a = -2.0 * 9.81 * 0.5 * np.sin(q) / I
    assert abs(d.qacc[0] - a) < 1e-13
    assert abs(d.qvel[0] - (v + h * a)) < 1e-15 and abs(d.qpos[0] - (q + h * (v + h * a))) < 1e-15
    assert d.time[0] == h  # mujoco_env_test.cpp:198-200
    # implicit joint damping: (I + h b) a' = tau - b v
    b = 0.4
    m2 = mjcf.compile_xml_string(PENDULUM_1DOF.format(damping=b))
    d2 = oracle_built.OracleData(m2)
    d2.qpos[0], d2.qvel[0] = q, v
    d2.step()
    a2 = (-2.0 * 9.81 * 0.5 * np.sin(q) - b * v) / (I + h * b)
    assert abs(d2.qvel[0] - (v + h * a2)) < 1e-15


def test_kinematics_and_inertia_match_independent_derivation(oracle_built, franka):
    """(ii) A1/A2: body frames vs direct transform composition; qM (CRB) vs sum_b m Jp'Jp + Jr' I Jr."""
    d = oracle_built.OracleData(franka)
    qpos, qvel = random_franka_state(franka, 5, seed=21)
    for e in range(5):
        d.reset()
        d.qpos[:] = qpos[e]
        d.qvel[:] = qvel[e] * 10
        d.forward()
        kin = refdyn.kinematics(franka, qpos[e])
        np.testing.assert_allclose(d.xpos.reshape(-1, 3), kin["xpos"], atol=1e-14)
        np.testing.assert_allclose(d.xmat.reshape(-1, 3, 3), kin["xmat"], atol=1e-14)
        np.testing.assert_allclose(d.xipos.reshape(-1, 3), kin["xipos"], atol=1e-14)
        np.testing.assert_allclose(d.ximat.reshape(-1, 3, 3), kin["ximat"], atol=1e-14)
        M = refdyn.mass_matrix(franka, qpos[e])
        np.testing.assert_allclose(dense_M(franka, d.qM), M, atol=1e-13)
        # A9: RNE bias vs finite-differenced Lagrangian (FD error ~1e-8)
        c = refdyn.bias_lagrange(franka, qpos[e], qvel[e] * 10)
        np.testing.assert_allclose(d.qfrc_bias, c, atol=2e-6, rtol=1e-6)


def test_internal_identities(oracle_built, franka):
    """(iii) L'DL reconstructs M; M qacc_smooth = qfrc_smooth; solve_m inverts M; sensors read what they should."""
    d = oracle_built.OracleData(franka)
    qpos, qvel = random_franka_state(franka, 3, seed=22)
    rng = np.random.default_rng(0)
    for e in range(3):
        d.reset()
        d.qpos[:] = qpos[e]
        d.qvel[:] = qvel[e] * 5
        d.ctrl[:] = rng.uniform(-5, 5, franka["nu"])
        d.forward()
        M = dense_M(franka, d.qM)
        nv = franka["nv"]
        L = np.eye(nv)
        for i in range(nv):
            adr, j = franka["dof_Madr"][i] + 1, franka["dof_parentid"][i]
            while j >= 0:
                L[i, j] = d.qLD[adr]
                adr += 1
                j = franka["dof_parentid"][j]
        D = np.diag([d.qLD[franka["dof_Madr"][i]] for i in range(nv)])
        np.testing.assert_allclose(L.T @ D @ L, M, atol=1e-12)
        np.testing.assert_allclose(M @ d.qacc_smooth, d.qfrc_smooth, atol=1e-11)
        x = rng.normal(size=nv)
        np.testing.assert_allclose(M @ d.solve_m(x), x, atol=1e-11)
        np.testing.assert_allclose(d.qfrc_smooth, d.qfrc_passive - d.qfrc_bias + d.qfrc_actuator, atol=1e-13)
        spring = np.zeros(nv)  # (the finger slides carry a centring spring: -k (q - springref))
        for j in range(franka["njnt"]):
            qa, da = franka["jnt_qposadr"][j], franka["jnt_dofadr"][j]
            spring[da] = -franka["jnt_stiffness"][j] * (d.qpos[qa] - franka["qpos_spring"][qa])
        np.testing.assert_allclose(d.qfrc_passive, spring - np.asarray(franka["dof_damping"]) * d.qvel, atol=1e-12)
        s = d.sensordata
        np.testing.assert_array_equal(s[:9], d.qpos)
        np.testing.assert_array_equal(s[9:18], d.qvel)
        np.testing.assert_allclose(s[18:21], d.site_xpos[:3], atol=0)
        assert abs(np.linalg.norm(s[21:25]) - 1) < 1e-12


def test_energy_and_momentum_free_body(oracle_built):
    """(iii) a torque-free free body: linear momentum conserved exactly under zero gravity, angular momentum
    about the com conserved to O(h) drift, time advances by exactly h per step."""
    xml = """<mujoco><option timestep="0.001" gravity="0 0 0"><flag contact="disable"/></option><worldbody>
    <body name="b" pos="0 0 1"><freejoint name="f"/><inertial pos="0 0 0" mass="1.5" diaginertia="0.02 0.03 0.05"/></body>
    </worldbody></mujoco>"""
    m = mjcf.compile_xml_string(xml)
    d = oracle_built.OracleData(m)
    d.qvel[:] = [0.3, -0.2, 0.1, 1.0, 2.0, -1.5]
    d.forward()
    I = np.diag([0.02, 0.03, 0.05])

    def ang_mom():
        R = d.xmat.reshape(-1, 3, 3)[1]
        return R @ I @ d.qvel[3:6]  # free-joint angular velocity is expressed in the body frame

    L0 = ang_mom()
    for k in range(200):
        d.step()
    np.testing.assert_allclose(d.qvel[:3], [0.3, -0.2, 0.1], atol=1e-15)
    np.testing.assert_allclose(d.qpos[:3], np.array([0, 0, 1]) + 0.2 * np.array([0.3, -0.2, 0.1]), atol=1e-13)
    d.forward()
    assert np.linalg.norm(ang_mom() - L0) < 2e-3 * np.linalg.norm(L0)
    assert abs(np.linalg.norm(d.qpos[3:7]) - 1) < 1e-14
    assert abs(d.time[0] - 0.2) < 1e-12


def test_pendulum_world_equilibrium_stays_at_rest(oracle_built):
    """Reference fact (7): a pendulum released at its stable equilibrium stays EXACTLY at rest
    (mujoco_sensors_test.cpp:389-391: ground-truth variance == 0 over 1001 steps)."""
    m = mjcf.compile_xml_file(os.path.join(GOLDEN, "pendulum_world.xml"))
    assert (m["nq"], m["nv"], m["nbody"]) == (13, 11, 6)  # SURVEY.md §8 model table, config 1
    assert m["solver"] == 2 and len(m.get("skipped_collision_pairs", [])) == 0  # Newton as shipped, every geom pair handled
    d = oracle_built.OracleData(m)
    q0 = d.qpos.copy()
    for _ in range(1001):
        d.step()
    # the articulated pendulum hangs straight down: exactly at rest; the ball settles on the ground plane
    assert np.all(d.qvel[:5] == 0) and np.array_equal(d.qpos[:6], q0[:6])
    assert 0.049 < d.qpos[8] < 0.0501 and np.abs(d.qvel[5:]).max() < 1e-6
    d.reset()
    assert np.array_equal(d.qpos, m["qpos0"]) and d.time[0] == 0 and np.all(d.qvel == 0)


def test_philox_normal_statistics(oracle_built):
    """The counter-based N(0,1) stream behind the OU ctrl-noise injector (mujoco_env.cpp:469-481)."""
    L = oracle_built.lib()
    z = np.array([L.mjo_normal(12345, e, s, 0) for e in range(200) for s in range(100)])
    assert abs(z.mean()) < 0.03 and abs(z.std() - 1) < 0.03
    assert L.mjo_normal(12345, 3, 7, 1) == L.mjo_normal(12345, 3, 7, 1)
    assert L.mjo_normal(12345, 3, 7, 1) != L.mjo_normal(12345, 3, 7, 2)
    # Philox-4x32-10 known-answer test (Random123 kat_vectors: zero counter, zero key)
    import ctypes as C
    ctr = (C.c_uint32 * 4)(0, 0, 0, 0)
    key = (C.c_uint32 * 2)(0, 0)
    L.mjo_philox4x32(ctr, key)
    assert [hex(x) for x in ctr] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]


def test_mjcf_compiler_basics():
    """Model compiler facts used everywhere else: sizes of the shipped models, capsule fromto, inertia from geoms."""
    m = mjcf.load_asset("franka_like")
    assert (m["nq"], m["nv"], m["nu"], m["nbody"], m["nsensordata"], m["nM"]) == (9, 9, 9, 12, 25, 44)
    p = mjcf.compile_xml_file(os.path.join(GOLDEN, "pendulum_world.xml"), disable=("contact",))
    g = p["names"]["geom"].index("EE")
    np.testing.assert_allclose(p["geom_size"][g][:2], [0.02, 0.1])
    np.testing.assert_allclose(p["geom_pos"][g], [0, 0, 0.2])
    ball = p["names"]["body"].index("body_ball")
    assert abs(p["body_mass"][ball] - 0.1) < 1e-15
    np.testing.assert_allclose(p["body_inertia"][ball], 0.4 * 0.1 * 0.05 ** 2)
    assert p["cone"] == 1 and p["solver"] == 2 and p["timestep"][0] == 0.001
    with pytest.raises(mjcf.MjcfError):
        mjcf.compile_xml_string("<mujoco><worldbody><body><geom type='mesh'/></body></worldbody></mujoco>")


def test_operation_counting_build_is_the_same_oracle(oracle_built, franka):
    """oracle/libmjo_count.so (the algorithmic flop count) is the unchanged oracle source with a counting `double`:
    its rollout is bit-identical to libmjo.so's and its count is in the expected range for the 9-dof arm."""
    import ctypes as C
    import os
    from mujoco_ros_pkgs_amd import binding
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    L = C.CDLL(os.path.join(here, "oracle", "libmjo_count.so"))
    L.mjo_flops_get.restype = C.c_ulonglong
    L.mjo_rollout.argtypes = [C.POINTER(binding.ModelDesc), C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                              C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_double, C.c_uint64, C.c_int64, C.c_int]
    from conftest import random_franka_state
    qpos, qvel = random_franka_state(franka, 3, seed=2)
    ref_q, ref_v, _ = oracle_built.rollout(franka, qpos, qvel, 25, noise_std=5.0, noise_rate=0.1, seed=7)
    desc, keep = binding.make_desc(franka)
    q, v = np.ascontiguousarray(qpos).copy(), np.ascontiguousarray(qvel).copy()
    pd = C.POINTER(C.c_double)
    L.mjo_flops_reset()
    L.mjo_rollout(C.byref(desc), 3, 25, q.ctypes.data_as(pd), v.ctypes.data_as(pd), None, None, 5.0, 0.1, 7, 0, 1)
    assert np.array_equal(q, ref_q) and np.array_equal(v, ref_v)
    per = L.mjo_flops_get() / 75.0
    assert 4000 < per < 16000, per   # SURVEY.md 8d planning estimate: ~8 kflop for the no-contact arm
