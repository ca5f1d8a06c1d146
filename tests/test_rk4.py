"""RK4 (<option integrator="RK4">, mj_RungeKutta(m, d, 4): viewer-selectable in the reference, mujoco_ros/src/viewer.cpp:579-603).
CPU: the oracle's restatement converges with order 4 (Euler: order 1) on a double pendulum and conserves energy three orders of
magnitude better; GPU: the step loop's RK4 path against the oracle on the three model families (no constraints / PGS / Newton),
split steps == fused steps, sensordata = the step's own evaluation (sub-stage evaluations skip the sensors)."""
import os

import numpy as np
import pytest

PENDULUM = """<mujoco><option timestep="{dt}" integrator="{integ}" gravity="0 0 -9.81"><flag energy="enable"/></option><worldbody>
<body name="b" pos="0 0 1"><joint name="h" type="hinge" axis="0 1 0"/><geom type="capsule" fromto="0 0 0 0.5 0 0" size="0.02" mass="1"/>
<body name="c" pos="0.5 0 0"><joint name="h2" type="hinge" axis="0 1 0"/><geom type="capsule" fromto="0 0 0 0.4 0 0" size="0.02" mass="0.5"/></body></body>
</worldbody></mujoco>"""


def _run(po, dt, integ, T=0.4):
    from mujoco_ros_pkgs_amd import mjcf
    m = mjcf.compile_xml_string(PENDULUM.format(dt=dt, integ=integ))
    d = po.OracleData(m)
    d.reset()
    d.qpos[:] = [0.3, -0.2]
    d.qvel[:] = [0.5, 0.1]
    d.forward()
    e0 = float(d.energy[0] + d.energy[1])
    d.step(int(round(T / dt)))
    d.forward()
    return np.array(d.qpos).copy(), float(d.time[0]), abs(float(d.energy[0] + d.energy[1]) - e0)


def test_oracle_rk4_is_fourth_order(oracle_built):
    ref, _, _ = _run(oracle_built, 0.4 / 4096, "RK4")
    err = {}
    for integ in ("Euler", "RK4"):
        err[integ] = [np.abs(_run(oracle_built, 0.4 / n, integ)[0] - ref).max() for n in (64, 128, 256)]
    assert 1.8 < err["Euler"][0] / err["Euler"][1] < 2.2 and 1.8 < err["Euler"][1] / err["Euler"][2] < 2.2
    assert 14 < err["RK4"][0] / err["RK4"][1] < 18 and 14 < err["RK4"][1] / err["RK4"][2] < 18
    assert err["RK4"][0] < 1e-4 * err["Euler"][0]
    _, t, de_rk = _run(oracle_built, 0.4 / 128, "RK4")
    _, _, de_eu = _run(oracle_built, 0.4 / 128, "Euler")
    assert abs(t - 0.4) < 1e-12 and de_rk < 1e-3 * de_eu


def test_mjcf_refuses_the_implicit_integrator():
    """mjINT_IMPLICIT (Coriolis derivatives, LU factor) is refused; implicitfast is implemented since round 6 (tests/test_activation_states.py)."""
    from mujoco_ros_pkgs_amd import mjcf
    with pytest.raises(mjcf.MjcfError):
        mjcf.compile_xml_string(PENDULUM.format(dt=0.002, integ="implicit"))
    assert mjcf.compile_xml_string(PENDULUM.format(dt=0.002, integ="implicitfast"))["integrator"] == 3


def _model(kind):
    from mujoco_ros_pkgs_amd import mjcf
    if kind == "franka_like":
        return mjcf.load_asset("franka_like", override={"integrator": "RK4"})
    over = {"integrator": "RK4"}
    if kind == "Newton":
        over["solver"] = "Newton"
    return mjcf.compile_xml_file(os.path.join(mjcf.ASSET_DIR, "franka_table.xml"), override=over)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["franka_like", "PGS", "Newton"])
def test_gpu_rk4_matches_oracle(kind, oracle_built):
    from mujoco_ros_pkgs_amd import engine, mjcf
    from conftest import random_franka_state
    from test_gpu_contact import scenario_states
    model = mjcf.Model(dict(_model(kind)))
    model["enableflags"] = int(model["enableflags"]) | 2   # mjENBL_ENERGY: mjData.energy must follow the evaluations on every path
    assert model["integrator"] == 1
    nenv, K = 12, 20
    if kind == "franka_like":
        qpos, qvel = random_franka_state(model, nenv, seed=3)
    else:
        qpos, qvel = scenario_states(model, nenv, seed=5)
    ctrl = np.random.default_rng(4).uniform(-3, 3, (nenv, model["nu"]))
    cm = engine.CompiledModel(model)
    outs = []
    for split in (False, True):
        b = engine.Batch(cm, nenv)
        b.set("qpos", qpos)
        b.set("qvel", qvel)
        b.set("ctrl", ctrl)
        if split:
            for _ in range(K):
                b.step1()
                b.step2()
        else:
            b.step(K)
        outs.append((b.get("qpos"), b.get("qvel"), b.get("sensordata"), b.get("time"), b.get("energy")))
        b.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])   # split == fused, bit for bit
    assert np.array_equal(outs[0][4], outs[1][4]) and np.all(outs[0][4][:, 1] > 0)              # ... mjData.energy included
    # ... and so are chained split steps (second half of one step + first half of the next in one launch, mjb_step21_prefix)
    b = engine.Batch(cm, nenv)
    b.set("qpos", qpos)
    b.set("qvel", qvel)
    b.set("ctrl", ctrl)
    assert b.lib.mjb_step1_prefix(b.ptr, nenv) == 0
    for k in range(K):
        assert (b.lib.mjb_step21_prefix if k + 1 < K else b.lib.mjb_step2_prefix)(b.ptr, nenv) == 0
    assert np.array_equal(b.get("qpos"), outs[0][0]) and np.array_equal(b.get("qvel"), outs[0][1])
    b.close()
    # ... and so is the step cut at the callback points of its four evaluations (mjb_step2_rk_prefix): between the cuts the callback
    # envs show the view of the NEXT evaluation, at mj_RungeKutta's time t0 + c h
    b = engine.Batch(cm, nenv)
    b.set("qpos", qpos)
    b.set("qvel", qvel)
    b.set("ctrl", ctrl)
    h = model["timestep"][0]
    for k in range(K):
        assert b.lib.mjb_step1_prefix(b.ptr, nenv) == 0
        for rk in range(4):
            assert b.lib.mjb_step2_rk_prefix(b.ptr, nenv, rk) == 0
            if k == 0:
                assert np.allclose(b.get("time")[:, 0], (0.5, 0.5, 1.0, 1.0)[rk] * h, rtol=0, atol=1e-15)
    assert np.array_equal(b.get("qpos"), outs[0][0]) and np.array_equal(b.get("qvel"), outs[0][1])
    assert np.array_equal(b.get("sensordata"), outs[0][2])
    assert np.array_equal(b.get("energy"), outs[0][4])      # the energy of the LAST evaluation, as the fused step leaves it
    assert b.lib.mjb_step2_rk_prefix(b.ptr, nenv, 0) != 0   # (no split step open)
    b.close()
    q, v, sd, t, _ = outs[0]
    oq, ov, osd = oracle_built.rollout(model, qpos, qvel, K, ctrl=ctrl)
    tol = 1e-9 if kind == "franka_like" else 1e-7
    assert np.abs(q - oq).max() < tol and np.abs(v - ov).max() < 100 * tol, (np.abs(q - oq).max(), np.abs(v - ov).max())
    assert np.abs(sd - osd).max() < 1e-6
    assert np.allclose(t[:, 0], K * model["timestep"][0], rtol=0, atol=1e-12)
    # ... and RK4 is not Euler: the same rollout under Euler lands somewhere else
    from mujoco_ros_pkgs_amd import mjcf
    euler = mjcf.Model(dict(model))
    euler["integrator"] = 0
    eq, _, _ = oracle_built.rollout(euler, qpos, qvel, K, ctrl=ctrl)
    assert np.abs(eq - oq).max() > 1e-6
