"""GPU parity: the HIP engine (through the C-ABI) against the CPU oracle on identical inputs.

Tolerances (fp64, stated per SURVEY.md §8c(vi)): one forward pass / one step <= 1e-11 relative+absolute on
every quantity (the GPU build contracts a*b+c into fma, the oracle does not, so bit equality is not
expected); 100-step rollouts <= 1e-8 on qpos/qvel (error growth of a chaotic 9-dof arm).
"""
import numpy as np
import pytest

from conftest import random_franka_state

pytestmark = pytest.mark.gpu

FWD_FIELDS = ["xpos", "xquat", "xmat", "xipos", "ximat", "xanchor", "xaxis", "site_xpos", "site_xmat", "subtree_com",
              "cinert", "crb", "cdof", "qM", "qLD", "qLDiagInv", "actuator_length", "cvel", "cdof_dot",
              "actuator_velocity", "qfrc_passive", "qfrc_bias", "cacc", "actuator_force",
              "qfrc_actuator", "qfrc_smooth", "qacc_smooth", "qfrc_constraint", "qacc", "sensordata"]


# (cfrc_body is engine-side scratch: the HIP rne keeps each body's own inertial force and sums over masks, the oracle
#  keeps mj_rne's in-place subtree sums; both feed the same qfrc_bias, which is compared)


def _close(a, b, tol, what):
    a, b = np.asarray(a), np.asarray(b)
    err = np.abs(a - b)
    bound = tol * (1.0 + np.abs(b))
    assert np.all(err <= bound), f"{what}: max err {err.max():.3e} (ref scale {np.abs(b).max():.3e})"


@pytest.fixture(scope="module")
def setup(franka, oracle_built):
    from mujoco_ros_pkgs_amd import engine
    cm = engine.CompiledModel(franka)
    return franka, cm, engine, oracle_built


@pytest.mark.parametrize("lanes", [8, 16, 32, 64])
def test_forward_fields_match_oracle(setup, lanes):
    model, cm, engine, po = setup
    nenv = 37  # ragged vs every group size
    qpos, qvel = random_franka_state(model, nenv, seed=1)
    ctrl = np.random.default_rng(2).uniform(-20, 20, (nenv, model["nu"]))
    b = engine.Batch(cm, nenv)
    b.set_launch(lanes, 0)
    b.set("qpos", qpos)
    b.set("qvel", qvel)
    b.set("ctrl", ctrl)
    b.forward()
    got = {f: b.get(f) for f in FWD_FIELDS}
    d = po.OracleData(model)
    for e in range(nenv):
        d.reset()
        d.qpos[:] = qpos[e]
        d.qvel[:] = qvel[e]
        d.ctrl[:] = ctrl[e]
        d.forward()
        for f in FWD_FIELDS:
            _close(got[f][e], d.field(f), 1e-11, f"{f} env {e} lanes {lanes}")
    b.close()


@pytest.mark.parametrize("lanes", [16, 64])
def test_step_matches_oracle(setup, lanes):
    model, cm, engine, po = setup
    nenv = 64
    qpos, qvel = random_franka_state(model, nenv, seed=3)
    ctrl = np.random.default_rng(4).uniform(-10, 10, (nenv, model["nu"]))
    b = engine.Batch(cm, nenv)
    b.set_launch(lanes, 0)
    b.set("qpos", qpos)
    b.set("qvel", qvel)
    b.set("ctrl", ctrl)
    b.step(1)
    q1, v1, t1 = b.get("qpos"), b.get("qvel"), b.get("time")
    oq, ov, _ = po.rollout(model, qpos, qvel, 1, ctrl=ctrl)
    _close(q1, oq, 1e-12, "qpos after 1 step")
    _close(v1, ov, 1e-11, "qvel after 1 step")
    assert np.all(t1 == model["timestep"][0])  # reference: mujoco_env_test.cpp:198-200 (exact)
    b.step(99)
    q100, v100 = b.get("qpos"), b.get("qvel")
    oq, ov, os_ = po.rollout(model, qpos, qvel, 100, ctrl=ctrl)
    _close(q100, oq, 1e-8, "qpos after 100 steps")
    _close(v100, ov, 1e-8, "qvel after 100 steps")
    _close(b.get("sensordata"), os_, 1e-8, "sensordata after 100 steps")
    assert np.allclose(b.get("time")[:, 0], 100 * model["timestep"][0], atol=1e-6)  # mujoco_env_test.cpp:219-221
    b.close()


def test_fused_equals_single_steps_and_split(setup):
    """K fused steps == K single-step launches == K (step1 + step2) pairs, bit for bit."""
    model, cm, engine, po = setup
    nenv = 16
    qpos, qvel = random_franka_state(model, nenv, seed=5)
    ctrl = np.random.default_rng(6).uniform(-10, 10, (nenv, model["nu"]))
    outs = []
    for mode in range(3):
        b = engine.Batch(cm, nenv)
        b.set("qpos", qpos)
        b.set("qvel", qvel)
        b.set("ctrl", ctrl)
        if mode == 0:
            b.step(7)
        elif mode == 1:
            for _ in range(7):
                b.step(1)
        else:
            for _ in range(7):
                b.step1()
                b.step2()
        outs.append((b.get("qpos"), b.get("qvel"), b.get("time"), b.get("sensordata")))
        b.close()
    for k in range(4):
        assert np.array_equal(outs[0][k], outs[1][k])
        assert np.array_equal(outs[0][k], outs[2][k])


def test_ctrl_noise_matches_oracle(setup):
    model, cm, engine, po = setup
    nenv = 32
    qpos, qvel = random_franka_state(model, nenv, seed=7)
    std, rate, seed, off = 5.0, 0.1, 12345, 1000
    b = engine.Batch(cm, nenv)
    b.set("qpos", qpos)
    b.set("qvel", qvel)
    b.set_ctrl_noise(std, rate, seed, off)
    b.step(20)
    b.step(5)  # step counter continues across launches
    oq, ov, _ = po.rollout(model, qpos, qvel, 25, noise_std=std, noise_rate=rate, seed=seed, env_offset=off)
    _close(b.get("qpos"), oq, 1e-9, "qpos with OU ctrl noise")
    _close(b.get("qvel"), ov, 1e-9, "qvel with OU ctrl noise")
    ctrl = b.get("ctrl")
    assert np.abs(ctrl).max() > 0 and np.std(ctrl) > 0.1 * std * np.sqrt(1 - np.exp(-2 * 25 * model["timestep"][0] / rate))
    b.close()


def test_reset_mask_and_callback_fields(setup):
    model, cm, engine, po = setup
    nenv = 8
    qpos, qvel = random_franka_state(model, nenv, seed=8)
    b = engine.Batch(cm, nenv)
    b.set("qpos", qpos)
    b.set("qvel", qvel)
    b.step(3)
    mask = np.zeros(nenv, np.uint8)
    mask[[1, 5]] = 1
    before = b.get("qpos")
    b.reset(mask)
    after, t = b.get("qpos"), b.get("time")
    for e in range(nenv):
        if mask[e]:
            assert np.array_equal(after[e], model["qpos0"]) and t[e, 0] == 0  # mujoco_env_test.cpp:483-529
            assert np.all(b.get("qvel")[e] == 0)
        else:
            assert np.array_equal(after[e], before[e]) and t[e, 0] > 0
    # qfrc_applied and xfrc_applied (the plugin-writable force fields) act as in the oracle
    b.reset()
    b.set("qpos", qpos)
    b.set("qvel", qvel)
    qa = np.random.default_rng(9).uniform(-3, 3, (nenv, model["nv"]))
    xf = np.zeros((nenv, model["nbody"] * 6))
    xf[:, 6 * 9:6 * 9 + 6] = np.random.default_rng(10).uniform(-5, 5, (nenv, 6))  # wrench on the hand
    b.set("qfrc_applied", qa)
    b.set("xfrc_applied", xf)
    b.forward()
    d = po.OracleData(model)
    for e in range(nenv):
        d.reset()
        d.qpos[:] = qpos[e]
        d.qvel[:] = qvel[e]
        d.qfrc_applied[:] = qa[e]
        d.xfrc_applied[:] = xf[e]
        d.forward()
        _close(b.get("qfrc_smooth")[e], d.qfrc_smooth, 1e-11, "qfrc_smooth with applied forces")
        _close(b.get("qacc")[e], d.qacc, 1e-10, "qacc with applied forces")
    b.close()


def test_large_batch_properties(setup):
    """BASELINE size (4096 envs): size-independent properties instead of a full oracle replay --
    identical envs stay bit-identical, every env's time is K*dt, state finite, sampled envs match the oracle."""
    model, cm, engine, po = setup
    nenv, K = 4096, 50
    qpos, qvel = random_franka_state(model, nenv, seed=11)
    qpos[1::2] = qpos[0::2]  # pairs of identical envs
    qvel[1::2] = qvel[0::2]
    b = engine.Batch(cm, nenv)
    b.set("qpos", qpos)
    b.set("qvel", qvel)
    b.step(K)
    q, v, t = b.get("qpos"), b.get("qvel"), b.get("time")
    assert np.all(np.isfinite(q)) and np.all(np.isfinite(v))
    assert np.array_equal(q[0::2], q[1::2]) and np.array_equal(v[0::2], v[1::2])
    assert np.allclose(t, K * model["timestep"][0], atol=1e-9)
    idx = np.array([0, 17, 1023, 2048, 4095])
    oq, ov, _ = po.rollout(model, qpos[idx], qvel[idx], K)
    _close(q[idx], oq, 1e-8, "sampled envs qpos")
    _close(v[idx], ov, 1e-8, "sampled envs qvel")
    b.close()


# ------------------------------------------------------------------ ball / free joints, springs, geoms (config 1 world)
@pytest.fixture(scope="module")
def pendulum_setup(oracle_built):
    import os
    from mujoco_ros_pkgs_amd import engine, mjcf
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    xml = open(os.path.join(golden, "pendulum_world.xml")).read()
    # same world plus a spring on the ball joint and on joint1, so every joint-type branch of
    # kinematics / comVel / passive / Euler is exercised (ball, hinge, free)
    xml = xml.replace('<joint name="balljoint" type="ball" pos="0 0 1"/>',
                      '<joint name="balljoint" type="ball" pos="0 0 1" stiffness="3.0" damping="0.2"/>')
    xml = xml.replace('<joint name="joint1" type="hinge" pos="0 0 0.6" axis="0 1 0"/>',
                      '<joint name="joint1" type="hinge" pos="0 0 0.6" axis="0 1 0" stiffness="1.5" springref="0.3"/>')
    model = mjcf.compile_xml_string(xml)
    return model, engine.CompiledModel(model), engine, oracle_built


def _pendulum_states(model, nenv, seed):
    rng = np.random.default_rng(seed)
    qpos = np.tile(np.asarray(model["qpos0"]), (nenv, 1))
    q = rng.normal(size=(nenv, 4))
    qpos[:, 0:4] = q / np.linalg.norm(q, axis=1, keepdims=True) * rng.uniform(0.8, 1.2, (nenv, 1))  # un-normalised on purpose
    qpos[:, 4:6] = rng.uniform(-1, 1, (nenv, 2))
    qpos[:, 6:9] = np.array([1.0, 0.0, 0.3]) + rng.uniform(-0.2, 0.2, (nenv, 3))
    q = rng.normal(size=(nenv, 4))
    qpos[:, 9:13] = q / np.linalg.norm(q, axis=1, keepdims=True)
    qvel = rng.uniform(-1, 1, (nenv, model["nv"]))
    return qpos, qvel


def test_ball_free_joint_fields_match_oracle(pendulum_setup):
    model, cm, engine, po = pendulum_setup
    nenv = 24
    qpos, qvel = _pendulum_states(model, nenv, seed=31)
    b = engine.Batch(cm, nenv)
    b.set("qpos", qpos)
    b.set("qvel", qvel)
    b.forward()
    fields = ["qpos", "xpos", "xquat", "xmat", "xipos", "ximat", "xanchor", "xaxis", "geom_xpos", "geom_xmat",
              "subtree_com", "cinert", "crb", "cdof", "qM", "qLD", "cvel", "cdof_dot", "qfrc_passive", "qfrc_bias",
              "qacc_smooth", "qacc"]
    got = {f: b.get(f) for f in fields}
    d = po.OracleData(model)
    for e in range(nenv):
        d.reset()
        d.qpos[:] = qpos[e]
        d.qvel[:] = qvel[e]
        d.forward()
        for f in fields:
            _close(got[f][e], d.field(f), 1e-10 if f in ("qacc", "qacc_smooth") else 1e-11, f"{f} env {e}")
    # quaternions in qpos come back normalised (ros_interface_test.cpp:342-351)
    assert np.allclose(np.linalg.norm(got["qpos"][:, 0:4], axis=1), 1, atol=1e-15)
    b.step(50)
    oq, ov, _ = po.rollout(model, qpos, qvel, 50)
    _close(b.get("qpos"), oq, 1e-7, "pendulum world qpos after 50 steps")
    _close(b.get("qvel"), ov, 1e-6, "pendulum world qvel after 50 steps")
    b.close()


def test_pregenerated_ctrl_noise_equals_in_kernel_generation(setup):
    """Long fused launches read their OU normals from a buffer a throughput kernel filled ahead of them (mjb_noise_kernel) -- the
    launch expected next is even generated on a side stream while the current one runs -- short ones draw them inside the step
    kernel: same Philox key, same function, so every split of a rollout into launches gives the same bits.  Covers: a speculation
    hit (two equal launches), a miss (a different length), a key change in between (mjb_set_ctrl_noise voids what was generated)."""
    model, cm, engine, po = setup
    nenv = 48
    qpos, qvel = random_franka_state(model, nenv, seed=11)

    def run(plan):
        b = engine.Batch(cm, nenv)
        b.set("qpos", qpos)
        b.set("qvel", qvel)
        b.set_ctrl_noise(5.0, 0.1, 12345, 7)
        for item in plan:
            if item == "rekey":
                b.set_ctrl_noise(3.0, 0.2, 999, 7)
            else:
                b.step(item)
        out = b.get("qpos"), b.get("qvel"), b.get("ctrl")
        b.close()
        return out

    fused = run([32, 32, 16, 40, "rekey", 32, 32, 5, 32])
    single = run([1] * 120 + ["rekey"] + [1] * 101)
    for a, c in zip(fused, single):
        assert np.array_equal(a, c)
    assert np.abs(fused[2]).max() > 0
