"""The lane = env form of the unconstrained fused step (csrc/mjb_lane_env.hip) against the CPU oracle and against the generic
16-lanes-per-env kernel, through the C-ABI.

Tolerances (fp64): the lane = env kernel sums over bodies in ascending order and refines hardware reciprocal seeds, the oracle walks
leaf to root and divides, so a step agrees to rounding (<= 1e-11 relative + absolute on qacc / sensordata after one step); rollouts
of K steps <= 1e-9 on qpos / qvel (the same bound the generic kernels are held to: tests/test_gpu_parity.py).
"""
import numpy as np
import pytest

from conftest import random_franka_state

pytestmark = pytest.mark.gpu


def _close(a, b, tol, what):
    a, b = np.asarray(a), np.asarray(b)
    err = np.abs(a - b)
    bound = tol * (1.0 + np.abs(b))
    assert np.all(err <= bound), f"{what}: max err {err.max():.3e} (ref scale {np.abs(b).max():.3e})"


@pytest.fixture(scope="module")
def eng(oracle_built):
    from mujoco_ros_pkgs_amd import engine, mjcf
    return engine, mjcf, oracle_built


@pytest.fixture(autouse=True, params=[0, 1, 2, 3], ids=["one_wave", "two_halves", "pipelined", "trio"])
def form(request, eng):
    """Every test runs on the kernel's four forms (mjb_lane_env_set_form): one wavefront per 64 envs, two wavefronts splitting the step
    into its position and velocity halves, two wavefronts pipelined body by body, three (the pose chain alone on the first).  (A forced form
    falls back where it does not fit.)"""
    engine = eng[0]
    lib = engine.binding.load_library()
    lib.mjb_lane_env_set_form(request.param)
    yield request.param
    lib.mjb_lane_env_set_form(-1)


def tree_state(model, nenv, seed):
    rng = np.random.default_rng(seed)
    qpos = np.tile(np.asarray(model["qpos0"], dtype=np.float64), (nenv, 1))
    for j in range(model["njnt"]):
        qpos[:, j] += rng.uniform(-0.6, 0.6, nenv) if model["jnt_type"][j] == 3 else rng.uniform(-0.03, 0.05, nenv)
    qvel = rng.uniform(-0.5, 0.5, (nenv, model["nv"]))
    return qpos, qvel


def make(engine, cm, nenv, qpos, qvel, mode, ctrl=None):
    b = engine.Batch(cm, nenv)
    b.set_lane_env(mode)
    b.set("qpos", qpos)
    b.set("qvel", qvel)
    if ctrl is not None:
        b.set("ctrl", ctrl)
    return b


@pytest.mark.parametrize("asset", ["franka_like", "lane_env_tree"])
def test_topology_is_compiled_in(eng, asset):
    engine, mjcf, _ = eng
    cm = engine.CompiledModel(mjcf.load_asset(asset))
    b = engine.Batch(cm, 4)
    topo, used = b.lane_env_info()
    assert topo >= 0 and not used
    b.close()
    # a model with constraint rows never matches
    cm3 = engine.CompiledModel(mjcf.load_asset("franka_table"))
    b3 = engine.Batch(cm3, 4)
    assert b3.lane_env_info()[0] == -1
    b3.close()


@pytest.mark.parametrize("asset,nenv", [("franka_like", 200), ("lane_env_tree", 77)])
def test_one_step_matches_oracle_and_generic_kernel(eng, asset, nenv):
    """Constant ctrl, one step: qpos, qvel, qacc, sensordata, time against the oracle's mjo_step and the 16-lane kernel."""
    engine, mjcf, po = eng
    model = mjcf.load_asset(asset)
    model["enableflags"] = int(model["enableflags"]) | 2
    cm = engine.CompiledModel(model)
    qpos, qvel = (random_franka_state if asset == "franka_like" else tree_state)(model, nenv, 3)
    qvel = qvel * 3
    ctrl = np.random.default_rng(4).uniform(-3, 3, (nenv, model["nu"]))
    out = {}
    for mode in (1, 0):
        b = make(engine, cm, nenv, qpos, qvel, mode, ctrl)
        b.step(1)
        assert b.lane_env_info()[1] == (mode == 1)
        out[mode] = {f: b.get(f) for f in ("qpos", "qvel", "qacc", "qacc_warmstart", "sensordata", "time", "energy", "ctrl")}
        b.close()
    d = po.OracleData(model)
    for e in range(nenv):
        d.reset()
        d.qpos[:] = qpos[e]
        d.qvel[:] = qvel[e]
        d.ctrl[:] = ctrl[e]
        d.step(1)
        for f in ("qpos", "qvel", "qacc", "sensordata", "energy"):
            _close(out[1][f][e], d.field(f), 1e-11, f"{asset} {f} env {e} vs oracle")
    for f in out[1]:
        _close(out[1][f], out[0][f], 1e-11, f"{asset} {f} vs the generic kernel")


@pytest.mark.parametrize("asset,nenv,K,std", [("franka_like", 130, 10, 5.0), ("franka_like", 64, 200, 43.5), ("lane_env_tree", 96, 10, 1.0),
                                              ("lane_env_tree", 33, 300, 1.5)])
def test_noise_rollout_matches_oracle(eng, asset, nenv, K, std):
    """OU ctrl noise on: K < 16 generates the normals inside the kernel, longer launches read the pre-generated buffer."""
    engine, mjcf, po = eng
    model = mjcf.load_asset(asset)
    cm = engine.CompiledModel(model)
    qpos, qvel = (random_franka_state if asset == "franka_like" else tree_state)(model, nenv, 5)
    b = make(engine, cm, nenv, qpos, qvel, 1)
    b.set_ctrl_noise(std, 0.1, 777, 1000)
    b.step(K)
    assert b.lane_env_info()[1]
    q, v, sd = b.get("qpos"), b.get("qvel"), b.get("sensordata")
    t = b.get("time")
    b.close()
    oq, ov, osd = po.rollout(model, qpos, qvel, K, noise_std=std, noise_rate=0.1, seed=777, env_offset=1000, nthreads=8)
    _close(q, oq, 1e-9, f"{asset} qpos after {K} steps")
    _close(v, ov, 1e-9, f"{asset} qvel after {K} steps")
    _close(sd, osd, 1e-9, f"{asset} sensordata after {K} steps")
    assert np.allclose(t, K * float(np.ravel(model["timestep"])[0]), atol=1e-12)


def test_launch_splits_agree(eng):
    """3 x 40 steps, 120 steps at once and lane = env / generic launches interleaved end in the same state (the Philox counters,
    the OU state and time travel through HBM exactly as in the generic kernels)."""
    engine, mjcf, _ = eng
    model = mjcf.load_asset("franka_like")
    cm = engine.CompiledModel(model)
    nenv = 70
    qpos, qvel = random_franka_state(model, nenv, 9)
    res = []
    for plan in ([(1, 120)], [(1, 40), (1, 40), (1, 40)], [(1, 40), (0, 40), (1, 40)], [(0, 120)]):
        b = make(engine, cm, nenv, qpos, qvel, 1)
        b.set_ctrl_noise(20.0, 0.1, 5, 0)
        for mode, k in plan:
            b.set_lane_env(mode)
            b.step(k)
        res.append((b.get("qpos"), b.get("qvel"), b.get("ctrl"), b.get("time")))
        b.close()
    for i in (1, 2):
        for a, c in zip(res[0], res[i]):
            if i == 1:
                assert np.array_equal(a, c), "splitting a lane = env launch changed the result"
            else:
                _close(a, c, 1e-9, "lane = env / generic interleaved")
    for a, c in zip(res[0], res[3]):
        _close(a, c, 1e-9, "lane = env vs generic, 120 steps")


def test_bad_state_resets_like_mj_step(eng):
    """mj_checkPos / mj_checkVel / mj_checkAcc: NaN qpos, a huge qvel and a qacc beyond mjMAXVAL reset the env and count the warning,
    in the lane = env kernel as in the generic one and in the oracle."""
    engine, mjcf, po = eng
    model = mjcf.load_asset("franka_like")
    cm = engine.CompiledModel(model)
    nenv = 128
    qpos, qvel = random_franka_state(model, nenv, 11)
    qpos[5, 2] = np.nan
    qvel[17, 0] = 1e12
    qvel[17, 1] = np.nan
    qpos[40, 0] = np.inf
    qvel[90, 3] = 9e9  # fine for mj_checkVel; the damping force makes qacc huge but finite -> no reset unless beyond 1e10
    ctrl = np.random.default_rng(1).uniform(-5, 5, (nenv, model["nu"]))
    got = {}
    for mode in (1, 0):
        b = make(engine, cm, nenv, qpos, qvel, mode, ctrl)
        b.step(3)
        got[mode] = (b.get("qpos"), b.get("qvel"), b.get("ctrl"), b.get("time"), [b.warning(w) for w in range(8)])
        b.close()
    assert got[1][4] == got[0][4], f"warning counters differ: {got[1][4]} vs {got[0][4]}"
    assert got[1][4][4] == 2 and got[1][4][5] == 1 and got[1][4][6] >= 1
    for a, c in zip(got[1][:4], got[0][:4]):
        _close(a, c, 1e-9, "state after resets, lane = env vs generic")
    assert np.all(np.isfinite(got[1][0])) and np.all(got[1][2][5] == 0) and np.all(got[1][2][17] == 0)
    d = po.OracleData(model)
    for e in (5, 17, 40, 90):
        d.reset()
        d.qpos[:] = qpos[e]
        d.qvel[:] = qvel[e]
        d.ctrl[:] = ctrl[e]
        d.step(3)
        _close(got[1][0][e], d.field("qpos"), 1e-9, f"env {e} qpos vs oracle")
        _close(got[1][1][e], d.field("qvel"), 1e-9, f"env {e} qvel vs oracle")


def test_automatic_mode_threshold(eng):
    """mode -1: whole-batch launches of >= MJB_LANE_ENV_MIN_ENVS envs (default 4096) take the lane = env kernel, smaller ones do not."""
    engine, mjcf, po = eng
    model = mjcf.load_asset("franka_like")
    cm = engine.CompiledModel(model)
    for nenv, expect in ((1024, False), (4096, True), (16384 + 37, True)):
        qpos, qvel = random_franka_state(model, nenv, 2)
        b = make(engine, cm, nenv, qpos, qvel, -1)
        b.set_ctrl_noise(10.0, 0.1, 3, 0)
        b.step(20)
        assert b.lane_env_info()[1] == expect
        if expect:
            idx = np.arange(0, nenv, nenv // 6 + 1)
            q = b.get("qpos")
            assert np.all(np.isfinite(q))
            # sampled envs against the oracle (their global env ids key the Philox stream: one rollout call per sampled env)
            for e in idx[:6]:
                oq, ov, _ = po.rollout(model, qpos[e:e + 1], qvel[e:e + 1], 20, noise_std=10.0, noise_rate=0.1, seed=3, env_offset=int(e))
                _close(q[e], oq[0], 1e-9, f"env {e} of {nenv}")
        b.close()


def test_lean_lds_variant_at_full_occupancy(eng):
    """> 32768 envs = more than two wavefronts per CU: the 40 KB-per-wavefront instantiation (cinert in registers instead of LDS).  Sampled envs
    against the oracle, the whole batch finite, and equal to rounding to the same envs run in a small batch (the 160 KB instantiation: the LDS
    budget moves data, not arithmetic -- but the compiler contracts a few multiply-adds differently around the moved loads, so the two agree to
    ~1e-15, not bit for bit; one batch always runs one instantiation, whatever env range a launch covers)."""
    engine, mjcf, po = eng
    model = mjcf.load_asset("franka_like")
    cm = engine.CompiledModel(model)
    nenv, K = 40000, 25
    qpos, qvel = random_franka_state(model, nenv, 21)
    b = make(engine, cm, nenv, qpos, qvel, 1)
    b.set_ctrl_noise(30.0, 0.1, 99, 0)
    b.step(K)
    q, v, sd = b.get("qpos"), b.get("qvel"), b.get("sensordata")
    b.close()
    assert np.all(np.isfinite(q)) and np.all(np.isfinite(v))
    for e in (0, 63, 64, 12345, 39999):
        oq, ov, osd = po.rollout(model, qpos[e:e + 1], qvel[e:e + 1], K, noise_std=30.0, noise_rate=0.1, seed=99, env_offset=int(e))
        _close(q[e], oq[0], 1e-9, f"env {e} qpos")
        _close(v[e], ov[0], 1e-9, f"env {e} qvel")
        _close(sd[e], osd[0], 1e-9, f"env {e} sensordata")
    lo = 12288  # a 64-aligned slice: same lanes, same noise keys (env_offset), the other instantiation
    small = make(engine, cm, 256, qpos[lo:lo + 256], qvel[lo:lo + 256], 1)
    small.set_ctrl_noise(30.0, 0.1, 99, lo)
    small.step(K)
    _close(small.get("qpos"), q[lo:lo + 256], 1e-12, "160 KB vs 40 KB instantiation, qpos")
    _close(small.get("qvel"), v[lo:lo + 256], 1e-12, "160 KB vs 40 KB instantiation, qvel")
    small.close()


JIT_ARM = """
<mujoco model="jit_arm">
  <compiler angle="radian"/>
  <option timestep="0.002" gravity="0 0 -9.81" integrator="Euler"><flag contact="disable"/></option>
  <default><joint armature="0.05" damping="1.5"/></default>
  <worldbody>
    <body name="b1" pos="0 0 0.3">
      <inertial pos="0 0 0.1" mass="2.0" diaginertia="0.02 0.02 0.01"/>
      <joint name="j1" type="hinge" axis="0 0 1"/>
      <body name="b2" pos="0 0 0.25" quat="0.9 0.1 0.4 0.1">
        <inertial pos="0.1 0 0" quat="0.8 0.2 0.5 0.1" mass="1.2" diaginertia="0.01 0.012 0.006"/>
        <joint name="j2" type="hinge" axis="0 1 0" pos="0 0.01 -0.02" stiffness="5" springref="0.3"/>
        <body name="b3" pos="0.3 0 0">
          <inertial pos="0.1 0 0" mass="0.8" diaginertia="0.004 0.006 0.006"/>
          <joint name="j3" type="hinge" axis="0 1 0"/>
          <body name="b4" pos="0.25 0 0">
            <inertial pos="0.05 0 0" mass="0.3" diaginertia="0.001 0.001 0.001"/>
            <joint name="j4" type="slide" axis="1 0 0" stiffness="80" damping="6"/>
            <site name="tool" pos="0.1 0 0" quat="0.7071067811865476 0 0.7071067811865476 0"/>
          </body>
          <body name="b5" pos="0.1 0.1 0">
            <inertial pos="0 0.05 0" mass="0.2" diaginertia="0.0006 0.0004 0.0006"/>
            <joint name="j5" type="hinge" axis="1 0 0"/>
          </body>
        </body>
      </body>
    </body>
  </worldbody>
  <actuator>
    <motor joint="j1" ctrllimited="true" ctrlrange="-5 5"/>
    <position joint="j2" kp="60"/>
    <motor joint="j3" gear="2"/>
    <motor joint="j4" forcelimited="true" forcerange="-3 3"/>
  </actuator>
  <sensor>
    <jointpos joint="j5"/>
    <framepos objtype="site" objname="tool"/>
    <framequat objtype="site" objname="tool"/>
    <actuatorfrc actuator="3"/>
    <clock/>
  </sensor>
</mujoco>
"""


def test_topology_built_by_hiprtc(eng):
    """A model whose structure is NOT among the compiled-in topologies: its first eligible launch builds the kernel for it through hiprtc
    (mjb_lane_env_info: -2), and the result meets the same bars -- one step vs the oracle, a noise rollout, the reset paths vs the generic kernel."""
    engine, mjcf, po = eng
    xml = JIT_ARM.replace('actuator="3"', 'actuator="' + "act3" + '"').replace('<motor joint="j4" forcelimited', '<motor name="act3" joint="j4" forcelimited')
    model = mjcf.compile_xml_string(xml)
    model["enableflags"] = int(model["enableflags"]) | 2
    cm = engine.CompiledModel(model)
    nenv = 100
    rng = np.random.default_rng(8)
    qpos = np.tile(np.asarray(model["qpos0"], dtype=np.float64), (nenv, 1)) + rng.uniform(-0.7, 0.7, (nenv, model["nq"])) * np.where(np.asarray(model["jnt_type"]) == 3, 1.0, 0.05)
    qvel = rng.uniform(-1, 1, (nenv, model["nv"]))
    ctrl = rng.uniform(-2, 2, (nenv, model["nu"]))
    b = make(engine, cm, nenv, qpos, qvel, 1, ctrl)
    assert b.lane_env_info()[0] == -2
    b.step(1)
    topo, used = b.lane_env_info()
    if topo == -3:
        pytest.fail("hiprtc build of the lane = env kernel not available on this box: " + b.lane_env_error())
    assert used
    out = {f: b.get(f) for f in ("qpos", "qvel", "qacc", "sensordata", "energy")}
    b.close()
    d = po.OracleData(model)
    for e in range(nenv):
        d.reset()
        d.qpos[:] = qpos[e]
        d.qvel[:] = qvel[e]
        d.ctrl[:] = ctrl[e]
        d.step(1)
        for f in out:
            _close(out[f][e], d.field(f), 1e-11, f"jit arm {f} env {e}")
    # noise rollout (pre-generated normals) and resets
    b = make(engine, cm, nenv, qpos, qvel, 1)
    b.set_ctrl_noise(1.5, 0.1, 31, 500)
    b.step(150)
    oq, ov, osd = po.rollout(model, qpos, qvel, 150, noise_std=1.5, noise_rate=0.1, seed=31, env_offset=500, nthreads=8)
    _close(b.get("qpos"), oq, 1e-9, "jit arm qpos after 150 steps")
    _close(b.get("sensordata"), osd, 1e-9, "jit arm sensordata after 150 steps")
    b.close()
    qb, vb = qpos.copy(), qvel.copy()
    qb[3, 1] = np.nan
    vb[70, 2] = 2e11
    res = {}
    for mode in (1, 0):
        b = make(engine, cm, nenv, qb, vb, mode, ctrl)
        b.step(4)
        res[mode] = (b.get("qpos"), b.get("qvel"), [b.warning(w) for w in range(8)])
        b.close()
    assert res[1][2] == res[0][2] and res[1][2][4] == 1 and res[1][2][5] == 1
    _close(res[1][0], res[0][0], 1e-9, "jit arm, state after resets vs the generic kernel")


def two_arm_xml(n=7):
    """Two n-link hinge chains on one base: 2 n dofs, 2 n + 1 bodies -- more state than 40 KB of LDS per wavefront holds."""
    def chain(tag, y):
        out, close = "", ""
        for i in range(n):
            ax = ("0 0 1", "0 1 0", "1 0 0")[i % 3]
            out += f'<body name="{tag}{i}" pos="{0.0 if i else 0.1} {y if i == 0 else 0} {0.12 if i else 0.3}" quat="0.96 0.1 0.2 {0.05 * (i % 2)}">'
            out += f'<inertial pos="0.02 0 0.06" mass="{1.5 - 0.15 * i}" diaginertia="0.004 0.005 0.003"/><joint name="{tag}j{i}" type="hinge" axis="{ax}" damping="0.7" armature="0.02"/>'
            close += "</body>"
        return out + close
    act = "".join(f'<motor joint="{t}j{i}" ctrllimited="true" ctrlrange="-3 3"/>' for t in "LR" for i in range(n))
    return ('<mujoco model="two_arm"><compiler angle="radian"/><option timestep="0.002" integrator="Euler"><flag contact="disable"/></option><worldbody>'
            '<body name="base" pos="0 0 0.2"><inertial pos="0 0 0" mass="5" diaginertia="0.1 0.1 0.1"/>' + chain("L", 0.2) + chain("R", -0.2) +
            '</body></worldbody><actuator>' + act + f'</actuator><sensor><jointpos joint="Lj{n // 2}"/><jointvel joint="Rj{n - 1}"/></sensor></mujoco>')


def test_hiprtc_topology_beyond_the_lean_lds_budget(eng):
    """A 14-dof two-arm model built by hiprtc: its (qpos, qvel) pairs and body forces need 56 KB of LDS per wavefront, so even a batch that would get 40 KB
    per wavefront (> 32768 envs) runs the 80 KB instantiation.  Sampled envs against the oracle at both ends of the batch-size range."""
    engine, mjcf, po = eng
    model = mjcf.compile_xml_string(two_arm_xml())
    assert model["nv"] == 14 and model["nbody"] == 16
    cm = engine.CompiledModel(model)
    for nenv, K in ((200, 60), (33000, 20)):
        rng = np.random.default_rng(3)
        qpos = rng.uniform(-0.8, 0.8, (nenv, model["nq"]))
        qvel = rng.uniform(-1, 1, (nenv, model["nv"]))
        b = make(engine, cm, nenv, qpos, qvel, 1)
        assert b.lane_env_info()[0] == -2
        b.set_ctrl_noise(1.0, 0.1, 17, 0)
        b.step(K)
        topo, used = b.lane_env_info()
        assert used, f"lane = env kernel not used (info {topo}): {b.lane_env_error()}"
        q, v = b.get("qpos"), b.get("qvel")
        b.close()
        assert np.all(np.isfinite(q))
        for e in (0, nenv // 2, nenv - 1):
            oq, ov, _ = po.rollout(model, qpos[e:e + 1], qvel[e:e + 1], K, noise_std=1.0, noise_rate=0.1, seed=17, env_offset=int(e))
            _close(q[e], oq[0], 1e-9, f"two-arm env {e} of {nenv} qpos")
            _close(v[e], ov[0], 1e-9, f"two-arm env {e} of {nenv} qvel")


def test_form_follows_batch_size(eng):
    """The automatic rule (mjb_lane_env_set_form(-1)): three wavefronts while a block has a CU's LDS to itself (<= 64 x CUs envs), two halves up to
    twice that, one wavefront beyond; the three agree with the oracle on sampled envs."""
    engine, mjcf, po = eng
    import torch
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    lib = engine.binding.load_library()
    lib.mjb_lane_env_set_form(-1)
    model = mjcf.load_asset("franka_like")
    cm = engine.CompiledModel(model)
    for nenv, want in ((4096, 3), (64 * ncu, 3), (64 * ncu + 64, 1), (128 * ncu, 1), (128 * ncu + 64, 0)):
        qpos, qvel = random_franka_state(model, nenv, 5)
        b = make(engine, cm, nenv, qpos, qvel, 1)
        b.set_ctrl_noise(10.0, 0.1, 7, 0)
        b.step(12)
        assert b.lane_env_info()[1] and lib.mjb_lane_env_last_form() == want, (nenv, lib.mjb_lane_env_last_form(), want)
        q = b.get("qpos")
        for e in (0, nenv // 2 + 1, nenv - 1):
            oq, _, _ = po.rollout(model, qpos[e:e + 1], qvel[e:e + 1], 12, noise_std=10.0, noise_rate=0.1, seed=7, env_offset=int(e))
            _close(q[e], oq[0], 1e-9, f"env {e} of {nenv}")
        b.close()


def test_hiprtc_build_is_cached_on_disk(tmp_path):
    """the hiprtc build of a topology (~3 s per LDS budget and form) is kept under $MJB_JIT_CACHE, keyed on the gfx arch, the kernel
    headers' text, the LDS budget, the form and the topology: a SECOND PROCESS loads the code object instead of compiling, and computes the same step."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r'''
import ctypes as C, json, sys, time
sys.path.insert(0, %r)
sys.path.insert(0, %r)
import numpy as np
from mujoco_ros_pkgs_amd import engine, mjcf
from test_gpu_lane_env import JIT_ARM
xml = JIT_ARM.replace('actuator="3"', 'actuator="act3"').replace('<motor joint="j4" forcelimited', '<motor name="act3" joint="j4" forcelimited')
model = mjcf.compile_xml_string(xml)
cm = engine.CompiledModel(model)
b = engine.Batch(cm, 64)
b.set_lane_env(1)
rng = np.random.default_rng(2)
b.set("qvel", rng.uniform(-1, 1, (64, model["nv"])))
t0 = time.perf_counter()
b.step(3)
q = b.get("qpos")
dt = time.perf_counter() - t0
comp, hits = C.c_int(0), C.c_int(0)
b.lib.mjb_lane_env_jit_counts(C.byref(comp), C.byref(hits))
print(json.dumps(dict(used=b.lane_env_info()[1], compiled=comp.value, hits=hits.value, seconds=dt, qsum=float(np.abs(q).sum()), err=b.lane_env_error())))
''' % (root, os.path.join(root, "tests"))
    env = dict(os.environ, MJB_JIT_CACHE=str(tmp_path / "jit"))
    runs = []
    for _ in range(2):
        out = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        runs.append(json.loads(out.stdout.strip().splitlines()[-1]))
    first, second = runs
    if not first["used"]:
        pytest.skip("hiprtc build not available on this box: " + first["err"])
    assert first["compiled"] >= 1 and first["hits"] == 0, first
    assert second["used"] and second["compiled"] == 0 and second["hits"] >= 1, second
    assert second["qsum"] == first["qsum"]
    files = list((tmp_path / "jit").glob("le_*.hsaco"))
    assert len(files) == first["compiled"], files
    # (the point of the cache; with a margin, so that a cold page cache on a fresh box cannot fail the functional checks above)
    assert second["seconds"] < first["seconds"] + 1.0, (first["seconds"], second["seconds"])
    # MJB_JIT_CACHE=0: nothing is read or written
    off = subprocess.run([sys.executable, "-c", script], env=dict(os.environ, MJB_JIT_CACHE="0"), capture_output=True, text=True, timeout=300)
    r = json.loads(off.stdout.strip().splitlines()[-1])
    assert r["compiled"] >= 1 and r["hits"] == 0


def test_sensors_every_step_is_the_same_launch(eng):
    """mjb_set_sensors_every_step: the lane = env kernel evaluates A15 at every step instead of the launch's last -- nothing observable changes."""
    engine, mjcf, po = eng
    model = mjcf.load_asset("franka_like")
    cm = engine.CompiledModel(model)
    nenv = 192
    qpos, qvel = random_franka_state(model, nenv, 21)
    out = []
    for every in (False, True):
        b = make(engine, cm, nenv, qpos, qvel, 1)
        b.set_sensors_every_step(every)
        b.set_ctrl_noise(3.0, 0.1, 5, 0)
        b.step(40)
        assert b.lane_env_info()[1]
        out.append((b.get("qpos"), b.get("sensordata")))
        b.close()
    assert np.array_equal(out[0][0], out[1][0])
    assert np.abs(out[0][1] - out[1][1]).max() <= 1e-13
