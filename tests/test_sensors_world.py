"""The reference's sensors world (mujoco_ros_sensors/test/sensors_world.xml, kept as a data fixture) replayed on
the oracle and on the GPU engine: the ground-truth facts mujoco_sensors_test.cpp pins --
(7) a pendulum released at its stable equilibrium stays EXACTLY at rest: the ground-truth readings of
    vel_EE, vel_joint2, immovable_pos, immovable_quat have variance == 0 over 1001 single steps (:389-391, :584);
(8) a published reading equals sensordata[adr] / cutoff (:325-328, :550) -- here: sensordata layout and values."""
import os

import numpy as np
import pytest

from mujoco_ros_pkgs_amd import mjcf

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def world():
    return mjcf.compile_xml_file(os.path.join(GOLDEN, "sensors_world.xml"))


def expected_static(m):
    imm = m["names"]["body"].index("immovable")
    return np.concatenate([m["body_pos"][imm], m["body_quat"][imm]])


def test_sensor_layout_matches_reference_world():
    m = world()
    names = m["names"]["sensor"]
    assert names == ["immovable_pos", "immovable_quat", "vel_EE", "vel_joint2"]
    assert list(m["sensor_adr"]) == [0, 3, 7, 10] and m["nsensordata"] == 11
    assert list(m["sensor_type"]) == [23, 24, 2, 9]  # mjSENS_FRAMEPOS, FRAMEQUAT, VELOCIMETER, JOINTVEL
    assert np.all(m["sensor_cutoff"] == 0)  # cutoff <= 0 -> the plugin divides by 1 (mujoco_sensors_test.cpp:285)


def test_equilibrium_ground_truth_variance_is_zero_oracle(oracle_built):
    m = world()
    d = oracle_built.OracleData(m)
    rows = []
    for _ in range(1001):
        d.step()
        rows.append(d.sensordata.copy())
    rows = np.array(rows)
    assert np.all(np.ptp(rows, axis=0) == 0.0)  # every reading identical over 1001 steps => variance exactly 0
    np.testing.assert_array_equal(rows[0][:7], expected_static(m))
    assert np.all(rows[:, 7:] == 0.0)


@pytest.mark.gpu
def test_equilibrium_ground_truth_variance_is_zero_gpu(oracle_built):
    from mujoco_ros_pkgs_amd import engine
    m = world()
    cm = engine.CompiledModel(m)
    b = engine.Batch(cm, 5)
    rows = []
    for _ in range(1001):
        b.step(1)  # single steps, as the reference test does
        rows.append(b.get("sensordata"))
    rows = np.array(rows)  # [step, env, S]
    assert np.all(np.ptp(rows, axis=0) == 0.0)  # every reading identical over 1001 steps => variance exactly 0
    for e in range(5):
        np.testing.assert_allclose(rows[0, e, :7], expected_static(m), rtol=0, atol=1e-15)
    assert np.all(rows[:, :, 7:] == 0.0)
    # a moving pendulum: velocimeter / jointvel match the oracle
    q = np.tile(m["qpos0"], (5, 1))
    v = np.zeros((5, m["nv"]))
    v[:, 3] = np.linspace(0.5, 2.5, 5)  # joint1
    v[:, 4] = -1.0  # joint2
    b.reset()
    b.set("qpos", q)
    b.set("qvel", v)
    b.step(20)
    oq, ov, os_ = oracle_built.rollout(m, q, v, 20)
    assert np.abs(b.get("sensordata") - os_).max() < 1e-9
    assert np.abs(os_[:, 7:10]).max() > 0.1
    b.close()
