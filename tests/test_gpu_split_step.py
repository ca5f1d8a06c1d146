"""The SPLIT step of plain-PGS models (include/mjb.h: mjb_set_split_step; csrc/mjb_smooth_kernel.h + mjb_cstep_kernel): per step the smooth stages of
mj_step (mujoco_env.cpp:498,552,593) run one env per LANE -- free / ball / hinge / slide joints, geom and
site frames, both L'DL factors -- and hand their results to the constraint stages (A4 - A7, A13, A16), which run one env per wavefront.  Same step as
the fused kernel's and the oracle's: every test compares all three."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def eng(oracle_built):
    from mujoco_ros_pkgs_amd import engine, mjcf
    return engine, mjcf, oracle_built


@pytest.mark.parametrize("name,noise", [("franka_table", 2.0), ("split_step_tree", 0.3)])
@pytest.mark.parametrize("K", [1, 10, 60])
def test_split_step_matches_fused_kernel_and_oracle(eng, name, noise, K):
    """BASELINE configs[2]'s arm + table + cube, and a tree with the joint kinds of the reference's pendulum world (ball joint with an off-centre anchor,
    hinges, a free body; mujoco_ros/test/pendulum_world.xml:18-38) + a slide, springs on the ball and the free joint, limits, contacts."""
    import split_step_check
    d, wo, warn = split_step_check.run(name, 256, K, noise=noise, verbose=False)
    assert warn == (0, 0)
    # the two forms run the same arithmetic up to the order of a few sums (spatial quantities about the root body's origin instead of the subtree com)
    assert d["qpos"] <= 1e-12 and d["qvel"] <= 1e-11 and d["sensordata"] <= 1e-11 and d["time"] == 0 and d["ctrl"] == 0, d
    assert d["qacc"] <= 1e-9 and d["energy"] <= 1e-11, d
    assert wo["qpos"] <= 1e-11 and wo["qvel"] <= 1e-10 and wo["sensordata"] <= 1e-10, wo


def test_split_step_takes_slices_and_odd_batch_sizes(eng):
    """Env ranges that are not multiples of 64 (tail lanes of the smooth kernel), more slices than wavefronts, and a launch cut in two."""
    engine, mjcf, po = eng
    import split_step_check
    model = mjcf.load_asset("franka_table")
    cm = engine.CompiledModel(model)
    for n in (1, 63, 130):
        qpos, qvel = split_step_check.states("franka_table", model, n, seed=5)
        out = []
        for mode, cuts in ((0, (12,)), (1, (12,)), (1, (5, 7))):
            b = engine.Batch(cm, n)
            b.set_split_step(mode)
            b.set("qpos", qpos)
            b.set("qvel", qvel)
            b.set_ctrl_noise(2.0, 0.1, 777, 0)
            for c in cuts:
                b.step(c)
            assert b.split_step_info()[1] == bool(mode)
            out.append((b.get("qpos"), b.get("qvel"), b.get("sensordata")))
            b.close()
        assert np.abs(out[0][0] - out[1][0]).max() <= 1e-12 and np.abs(out[0][1] - out[1][1]).max() <= 1e-11
        # a launch cut in two is the same rollout, bit for bit
        assert np.array_equal(out[1][0], out[2][0]) and np.array_equal(out[1][1], out[2][1]) and np.array_equal(out[1][2], out[2][2])


def test_split_step_resets_like_mj_step(eng):
    """mj_checkPos / mj_checkVel inside the smooth kernel, mj_checkAcc inside the constraint kernel (the state one step after mj_resetData comes from
    the host: DevState::reset_step): warning counters and states as the fused kernel's and the oracle's."""
    engine, mjcf, po = eng
    import split_step_check
    model = mjcf.load_asset("franka_table")
    cm = engine.CompiledModel(model)
    n = 128
    qpos, qvel = split_step_check.states("franka_table", model, n, seed=9)
    qpos[5, 9] = np.nan
    qvel[17, 7] = 1e12
    qvel[17, 8] = np.nan
    qpos[40, 0] = np.inf
    qvel[90, 6] = 5e11 / model["dof_damping"][6] if model["dof_damping"][6] > 0 else 9e9  # finite for mj_checkVel? no: beyond mjMAXVAL -> reset by mj_checkVel
    qvel[100, 10] = 9e9  # fine for mj_checkVel; the damping force makes qacc huge -> mj_checkAcc resets when it passes 1e10
    ctrl = np.random.default_rng(1).uniform(-5, 5, (n, model["nu"]))
    got = {}
    for mode in (1, 0):
        b = engine.Batch(cm, n)
        b.set_split_step(mode)
        b.set("qpos", qpos)
        b.set("qvel", qvel)
        b.set("ctrl", ctrl)
        b.step(3)
        got[mode] = (b.get("qpos"), b.get("qvel"), b.get("ctrl"), b.get("time"), b.get("qacc_warmstart"), [b.warning(w) for w in range(8)])
        b.close()
    assert got[1][5] == got[0][5], f"warning counters differ: {got[1][5]} vs {got[0][5]}"
    assert got[1][5][4] >= 2 and got[1][5][5] >= 1
    for a, c in zip(got[1][:5], got[0][:5]):
        assert np.all(np.isfinite(a)) and np.abs(a - c).max() <= 1e-9
    d = po.OracleData(model)
    for e in (5, 17, 40, 100):
        d.reset()
        d.qpos[:] = qpos[e]
        d.qvel[:] = qvel[e]
        d.ctrl[:] = ctrl[e]
        d.step(3)
        assert np.abs(got[1][0][e] - d.field("qpos")).max() <= 1e-9 and np.abs(got[1][1][e] - d.field("qvel")).max() <= 1e-9, e


def test_split_step_statistics_and_mixing_with_the_fused_kernel(eng):
    """mjb_set_stats counts the constraint kernel's env-steps; fused and split launches of one batch interleave (state crosses HBM either way)."""
    engine, mjcf, po = eng
    import split_step_check
    model = mjcf.load_asset("franka_table")
    cm = engine.CompiledModel(model)
    n = 192
    qpos, qvel = split_step_check.states("franka_table", model, n, seed=3)
    ref = engine.Batch(cm, n)
    ref.set_split_step(0)
    mix = engine.Batch(cm, n)
    for b in (ref, mix):
        b.set("qpos", qpos)
        b.set("qvel", qvel)
        b.set_ctrl_noise(2.0, 0.1, 99, 0)
    mix.set_stats(True)
    for k, mode in enumerate((1, 0, 1, 1, 0)):
        mix.set_split_step(mode)
        mix.step(4 + k)
        ref.step(4 + k)
    assert np.abs(mix.get("qpos") - ref.get("qpos")).max() <= 1e-12 and np.abs(mix.get("qvel") - ref.get("qvel")).max() <= 1e-11
    st = mix.stats()
    assert st["evaluations"] == n * sum(4 + k for k in range(5)), st["evaluations"]
    ref.close()
    mix.close()


def test_automatic_mode_keeps_small_batches_on_the_fused_kernel(eng):
    engine, mjcf, po = eng
    model = mjcf.load_asset("franka_table")
    b = engine.Batch(engine.CompiledModel(model), 4096)
    b.step(2)
    assert b.split_step_info()[0] == 0 and not b.split_step_info()[1]  # eligible, but 4096 envs run faster fused (profiles/r06_split_step.txt)
    b.close()
