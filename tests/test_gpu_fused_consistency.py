"""Oracle-free invariant of the fused launch: K steps inside one kernel must equal K one-step launches BIT FOR BIT
(same code, same inputs, state crossing HBM instead of staying in LDS), in every kernel variant (lanes per env
x solver) and in both frame layouts (compact / full via keep_frame).

Why it exists: anything the compiler keeps in registers (or spills) across the in-kernel step loop is invisible to
one-step parity tests.  During r01 a register-allocation hazard of ROCm 7.2's LLVM (a VGPR spill emitted before the
exec mask was restored, see tools/check_spill_exec.py) corrupted frame addresses only from the second fused step
on; this test is the guard for that class of failure.  The derived fields left by keep_frame are also checked
against the split step (mjb_step1), which dumps the same quantities from a one-step kernel."""
import os

import numpy as np
import pytest

from conftest import random_franka_state

pytestmark = pytest.mark.gpu

STATE = ["qpos", "qvel", "qacc", "qacc_warmstart", "sensordata", "ctrl", "time"]


def _variants():
    out = [("franka_like", None, lanes) for lanes in (8, 16, 32, 64)]
    out += [("franka_table", s, 64) for s in ("PGS", "PGS-elliptic", "Newton", "Newton-elliptic", "CG")]
    out += [("shadow_hand_like", "hand-128", 64), ("shadow_hand_like", "hand-160", 64)]
    return out


def _model(name, solver):
    from mujoco_ros_pkgs_amd import mjcf
    if solver in (None, "PGS"):
        return mjcf.load_asset(name)
    if solver.startswith("hand-"):
        return mjcf.load_asset(name, nefcmax=int(solver[5:]))
    over = {"solver": solver.split("-")[0]}
    if solver.endswith("-elliptic"):
        over["cone"] = "elliptic"
    # (PGS with elliptic cones keeps a contact's rows in consecutive lanes of one wavefront: <= 64 rows = 16 contacts x 3 + 9 limits)
    kw = {"nefcmax": 57} if solver == "PGS-elliptic" else {}
    return mjcf.compile_xml_file(os.path.join(mjcf.ASSET_DIR, name + ".xml"), override=over, **kw)


def _initial(model, name, nenv):
    if name == "franka_like":
        return random_franka_state(model, nenv, seed=11)
    if name == "shadow_hand_like":
        from mujoco_ros_pkgs_amd import workloads
        return workloads.hand_grasp_states(model, nenv, seed=5)
    from test_gpu_contact import scenario_states
    return scenario_states(model, nenv, seed=5)


@pytest.mark.parametrize("name,solver,lanes", _variants())
@pytest.mark.parametrize("keep", [False, True])
def test_fused_steps_equal_single_steps(name, solver, lanes, keep):
    from mujoco_ros_pkgs_amd import engine
    model = _model(name, solver)
    cm = engine.CompiledModel(model)
    nenv, K = 67, 9
    qpos, qvel = _initial(model, name, nenv)
    res = []
    for plan in ([K], [1] * K, [4, 5]):
        b = engine.Batch(cm, nenv)
        if name == "franka_like":
            b.set_launch(lanes, 0)
        b.set_keep_frame(keep)
        b.set("qpos", qpos)
        b.set("qvel", qvel)
        b.set_ctrl_noise(0.1 if name == "shadow_hand_like" else 2.0, 0.1, 77, 0)
        for k in plan:
            b.step(k)
        res.append({f: b.get(f).copy() for f in STATE})
        b.close()
    for other, what in ((res[1], "1 x K"), (res[2], "4 + 5")):
        for f in STATE:
            assert np.array_equal(res[0][f], other[f]), (
                f"{name}/{solver}/lanes {lanes}: fused K differs from {what} in {f}: "
                f"max |d| {np.abs(res[0][f] - other[f]).max():.3e}")
    assert np.all(np.isfinite(res[0]["qpos"]))


@pytest.mark.parametrize("name,solver", [("franka_table", "PGS"), ("franka_table", "Newton"), ("shadow_hand_like", "Newton")])
def test_chunked_launch_equals_short_launches(name, solver):
    """Fused launches of >= 100 steps of the constrained kernels hand out (chunk of steps, env) work items from a device-side
    queue, an item waiting for its env's previous chunk (mjb_step.hip, `dyn`): bit-equal to the same steps as short launches,
    which run one item per env -- with more envs than the chip holds at once, so that items do wait."""
    from mujoco_ros_pkgs_amd import engine
    model = _model(name, solver)
    cm = engine.CompiledModel(model)
    nenv, K = (1100 if name == "shadow_hand_like" else 4500), 120
    qpos, qvel = _initial(model, name, nenv)
    res = []
    for plan in ([K], [40, 40, 40]):
        b = engine.Batch(cm, nenv)
        b.set("qpos", qpos)
        b.set("qvel", qvel)
        b.set_ctrl_noise(0.1 if name == "shadow_hand_like" else 2.0, 0.1, 77, 0)
        for k in plan:
            b.step(k)
        res.append({f: b.get(f).copy() for f in STATE})
        b.close()
    for f in STATE:
        assert np.array_equal(res[0][f], res[1][f]), f"{name}/{solver}: chunked launch differs in {f}: max |d| {np.abs(res[0][f] - res[1][f]).max():.3e}"
    assert np.all(np.isfinite(res[0]["qpos"]))


def test_chunked_launch_with_hwsim_and_kept_frame():
    """The same with the device-side hwsim stage (its PID state lives in HBM across the chunks) and with keep_frame (full frame
    layout, constraint rows built in MuJoCo's place): chunked == short launches, state and derived fields."""
    from mujoco_ros_pkgs_amd import engine
    from test_hwsim import _cfg, _commands
    model = _model("franka_table", "PGS")
    cm = engine.CompiledModel(model)
    spec = _cfg(model)
    nenv, K = 300, 120
    cp, cv, ce = _commands(len(spec), nenv, 1)
    qpos, qvel = _initial(model, "franka_table", nenv)
    res = []
    for plan in ([K], [40, 40, 40]):
        b = engine.Batch(cm, nenv)
        b.set_keep_frame(True)
        b.hwsim_configure(spec)
        b.hwsim_set_command("position", cp)
        b.hwsim_set_command("velocity", cv)
        b.hwsim_set_command("effort", ce)
        b.set("qpos", qpos)
        b.set("qvel", qvel)
        for k in plan:
            b.step(k)
        res.append({f: b.get(f).copy() for f in STATE + ["qfrc_applied", "efc_force", "xpos", "nefc"]})
        b.close()
    for f in res[0]:
        if f == "efc_force":  # (rows beyond nefc are whatever the frame held)
            for e in range(nenv):
                k = int(res[0]["nefc"][e, 0])
                assert np.array_equal(res[0][f][e, :k], res[1][f][e, :k]), f"chunked launch differs in {f}, env {e}"
        else:
            assert np.array_equal(res[0][f], res[1][f]), f"chunked launch differs in {f}"


@pytest.mark.parametrize("name,solver,lanes", [("franka_like", None, 16), ("franka_table", "PGS", 64),
                                               ("franka_table", "Newton", 64)])
def test_keep_frame_matches_split_step(name, solver, lanes):
    """Derived fields after a fused step(K) with keep_frame == those mjb_step1 dumps for the K-th step."""
    from mujoco_ros_pkgs_amd import engine
    model = _model(name, solver)
    cm = engine.CompiledModel(model)
    nenv, K = 33, 5
    qpos, qvel = _initial(model, name, nenv)
    fields = ["xpos", "xquat", "xipos", "cinert", "cdof", "qM", "qLD", "cvel", "qfrc_bias", "qfrc_smooth", "qacc_smooth"]
    if model["nefcmax"] > 0:
        fields += ["contact_dist", "contact_pos", "efc_J", "efc_aref", "efc_force", "qfrc_constraint"]
    a = engine.Batch(cm, nenv)
    a.set_keep_frame(True)
    a.set("qpos", qpos)
    a.set("qvel", qvel)
    a.step(K)
    fused = {f: a.get(f).copy() for f in fields}
    b = engine.Batch(cm, nenv)
    b.set("qpos", qpos)
    b.set("qvel", qvel)
    b.step(K - 1)
    b.step1()
    b.step2()
    split = {f: b.get(f).copy() for f in fields}
    for f in ("qpos", "qvel"):
        assert np.array_equal(a.get(f), b.get(f)), f
    for f in fields:
        x, y = fused[f], split[f]
        if f.startswith(("efc_", "contact_")):
            continue  # rows past nefc / ncon are stale scratch; compared through the solver outputs below
        assert np.array_equal(x, y), f"{f}: max |d| {np.abs(x - y).max():.3e}"
    if model["nefcmax"] > 0:
        assert np.array_equal(fused["qfrc_constraint"], split["qfrc_constraint"])
    a.close()
    b.close()


@pytest.mark.parametrize("asset,ncb", [("franka_like", 0), ("franka_like", 5), ("franka_like", 37), ("franka_table", 3), ("shadow_hand_like", 2)])
def test_prefix_split_step_equals_whole_batch_steps(asset, ncb):
    """mjb_step1_prefix / mjb_step_rest / mjb_step2_prefix (the host runtime's split step: only the callback envs [0, ncb) are split,
    the rest take the same step fused): with nothing written between the halves the batch must end up bit-identical to whole-batch
    fused steps -- under the OU ctrl noise, i.e. the Philox step counter advances once per step for both groups."""
    from mujoco_ros_pkgs_amd import engine, mjcf
    from bench import WORKLOADS, initial_state
    model = mjcf.load_asset(asset)
    cm = engine.CompiledModel(model)
    nenv, K = 37, 6
    qpos, qvel = initial_state(asset, model, nenv, seed=5)
    a, b = engine.Batch(cm, nenv), engine.Batch(cm, nenv)
    for x in (a, b):
        x.set("qpos", qpos)
        x.set("qvel", qvel)
        x.set_ctrl_noise(WORKLOADS[asset][1], 0.1, 99, 0)
    a.step(K)
    lib = b.lib
    for _ in range(K):
        assert lib.mjb_step1_prefix(b.ptr, ncb) == 0
        if ncb:
            assert np.all(np.isfinite(b.get("xpos", 0, ncb)))          # derived fields of the callback envs are readable here
            with pytest.raises(engine.EngineError):
                b.get("xpos", 0, nenv) if ncb < nenv else (_ for _ in ()).throw(engine.EngineError("n/a"))
        assert lib.mjb_step_rest(b.ptr, ncb) == 0
        assert lib.mjb_step2_prefix(b.ptr, ncb) == 0
    # (kernel variant 4 -- the hand: Newton, capacity > 128 rows -- runs env-steps of 61 .. 64 rows through different template
    #  instantiations of the solver on the full and on the fused frame (J in LDS / in HBM): same arithmetic, fma contraction may
    #  differ in the last bit, so its split envs are compared to 1e-9 instead of bit for bit; the fused rest stays bit-equal)
    exact = asset != "shadow_hand_like"
    for f in ("qpos", "qvel", "qacc", "sensordata", "time", "ctrl"):
        x, y = a.get(f), b.get(f)
        assert np.array_equal(x[ncb:], y[ncb:]), f
        assert np.array_equal(x[:ncb], y[:ncb]) if exact else np.allclose(x[:ncb], y[:ncb], rtol=0, atol=1e-9 * (1 + np.abs(x[:ncb]).max())), f
    # mjb_step2_prefix issues the rest itself when the caller skipped it
    assert lib.mjb_step1_prefix(b.ptr, ncb) == 0 and lib.mjb_step2_prefix(b.ptr, ncb) == 0
    a.step(1)
    assert np.array_equal(a.get("qpos")[ncb:], b.get("qpos")[ncb:])
    # chained split steps (mjb_step21_prefix: the second half of a step and the first half of the next in one launch, what the host
    # runtime issues between two callback rounds): bit-identical to the unchained sequence, derived fields readable in between
    c = engine.Batch(cm, nenv)
    c.set("qpos", qpos)
    c.set("qvel", qvel)
    c.set_ctrl_noise(WORKLOADS[asset][1], 0.1, 99, 0)
    assert lib.mjb_step1_prefix(c.ptr, ncb) == 0
    for k in range(K + 1):
        assert lib.mjb_step_rest(c.ptr, ncb) == 0
        if k < K:
            assert lib.mjb_step21_prefix(c.ptr, ncb) == 0
            if ncb:
                assert np.all(np.isfinite(c.get("xpos", 0, ncb)))
        else:
            assert lib.mjb_step2_prefix(c.ptr, ncb) == 0
    # (franka_like: the chained launch runs the generic kernel, the unchained halves the dense one -- same stages, DPP instead of
    #  loop reductions, so its callback envs agree to rounding; the other models run one kernel both ways: bit for bit)
    for f in ("qpos", "qvel", "qacc", "time", "ctrl"):
        x, y = b.get(f), c.get(f)
        assert np.array_equal(x[ncb:], y[ncb:]), f"chained split steps, fused rest: {f}"
        if asset == "franka_like" or not exact:
            assert np.allclose(x[:ncb], y[:ncb], rtol=0, atol=1e-9 * (1 + np.abs(x).max())), f"chained split steps: {f}"
        else:
            assert np.array_equal(x[:ncb], y[:ncb]), f"chained split steps: {f}"
    # abandoning a split step: before mjb_step_rest a new mjb_step1_prefix restarts it; after it the other envs have taken
    # the step, so mjb_step1_prefix is refused until the split is finished or the whole batch is reset / stepped
    assert lib.mjb_step1_prefix(c.ptr, ncb) == 0 and lib.mjb_step1_prefix(c.ptr, ncb) == 0
    assert lib.mjb_step_rest(c.ptr, ncb) == 0
    if ncb < nenv:
        assert lib.mjb_step1_prefix(c.ptr, ncb) != 0
    assert lib.mjb_step2_prefix(c.ptr, ncb) == 0
    b.step(1)
    assert np.array_equal(b.get("qpos")[ncb:], c.get("qpos")[ncb:])
    assert lib.mjb_step1_prefix(c.ptr, ncb) == 0 and lib.mjb_step_rest(c.ptr, ncb) == 0
    c.reset()                                                   # abandons the open split, waits for the rest launch
    assert lib.mjb_step1_prefix(c.ptr, ncb) == 0 and lib.mjb_step2_prefix(c.ptr, ncb) == 0
    assert np.all(np.isfinite(c.get("qpos")))
    a.close()
    b.close()
    c.close()
