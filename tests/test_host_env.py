"""Host runtime (libmjr_host.so): the reference's env-lifecycle, ROS-interface and plugin tests replayed on the
batched, ROS-free MujocoEnv.  Each test names the reference test it restates
(/root/reference mujoco_ros/test/{mujoco_env_test,ros_interface_test,mujoco_ros_plugin_test}.cpp).

Backends: "oracle" = tests/host_harness/oracle_backend.c (CPU; exercises host logic only, `-m "not gpu"`),
"hip" = the product backend libmjb.so (`-m gpu`)."""
import ctypes as C
import os
import subprocess
import time

import numpy as np
import pytest

from mujoco_ros_pkgs_amd import binding, mjcf

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")


@pytest.fixture(scope="session")
def host(oracle_built):
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(g.ROOT, "mujoco_ros_pkgs_amd", "host", "libmjr_host.so")):
        g.build()
    from mujoco_ros_pkgs_amd import host_binding
    host_binding.load_library()
    return host_binding


@pytest.fixture(scope="session")
def oracle_factory(oracle_built):
    d = os.path.join(HERE, "host_harness")
    if os.environ.get("MJB_PREBUILT") != "1":
        subprocess.check_call(["make", "-s", "-C", d])
    lib = C.CDLL(os.path.join(d, "liboracle_backend.so"))
    return lib.oracle_backend_factory  # address used as mjr_backend_factory


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def factory(request, oracle_factory):
    return oracle_factory if request.param == "oracle" else None


def pendulum():
    # the reference's world as shipped (contacts on, elliptic cones, Newton solver); one documented deviation:
    # the capsule-box pairs (pendulum links vs the static box, never in contact in these tests) are skipped
    # because that narrow phase is not implemented
    return mjcf.compile_xml_file(os.path.join(GOLDEN, "pendulum_world.xml"))


def empty():
    return mjcf.compile_xml_file(os.path.join(GOLDEN, "empty_world.xml"), disable=("contact",))


def wait(cond, timeout=2.0, dt=0.001):
    t0 = time.time()
    while not cond():
        if time.time() - t0 > timeout:
            return False
        time.sleep(dt)
    return True


def start(host, factory, model, params=None, nenv=1, admin_hash=""):
    env = host.HostEnv(params or {}, admin_hash)
    env.queue_model(model, nenv=nenv, backend_factory=factory)
    env.start()
    assert wait(lambda: env.operational_status == 0 and env.model_valid), "model was not loaded: " + env.load_error
    return env


# ------------------------------------------------------------------ mujoco_env_test.cpp
def test_eval_mode_requires_hash(host):
    """mujoco_env_test.cpp:59-75 (EvalModeWithoutHashThrow) / :77-93 (RunEvalNoHashValid)."""
    with pytest.raises(RuntimeError):
        host.HostEnv({"eval_mode": True})
    env = host.HostEnv({"eval_mode": False})
    env.close()
    env = host.HostEnv({"eval_mode": True}, admin_hash="some_hash")
    env.close()


def test_eval_mode_pause_needs_hash(host, factory):
    """mujoco_env_test.cpp:95-153: in eval mode pausing needs the admin hash, unpausing never does."""
    env = start(host, factory, empty(), {"eval_mode": True, "unpause": True}, admin_hash="right_hash")
    assert env.setting("run") == 1
    assert not env.toggle_paused(True, "wrong_hash") and env.setting("run") == 1
    assert not env.set_pause(True, "") and env.setting("run") == 1
    assert env.toggle_paused(True, "right_hash") and env.setting("run") == 0
    assert env.toggle_paused(False, "") and env.setting("run") == 1
    env.shutdown()


def test_step_refusals(host, factory):
    """mujoco_env_test.cpp:155-183 (StepWithoutModel... StepWhileRunning) and :255-275 (negative steps)."""
    env = host.HostEnv({"unpause": False})
    assert not env.step(1)  # no model loaded
    env.close()
    env = start(host, factory, empty(), {"unpause": True})
    assert not env.step(1)  # running
    env.shutdown()
    env = start(host, factory, empty(), {"unpause": False})
    assert not env.step(0) and not env.step(-10)
    env.shutdown()


def test_step_single_and_multi_while_paused(host, factory):
    """mujoco_env_test.cpp:185-204: one step => time == timestep EXACTLY; :206-225: 100 steps within 1e-6."""
    m = empty()
    env = start(host, factory, m, {"unpause": False})
    assert env.get_field("time")[0] == 0.0
    assert env.step(1)
    assert env.get_field("time")[0] == m["timestep"][0]
    assert env.sim_time == m["timestep"][0]  # /clock equivalent, ros_interface_test.cpp:78-98
    assert env.step(100)
    assert abs(env.get_field("time")[0] - 101 * m["timestep"][0]) < 1e-6
    env.shutdown()


def test_step_unblocked(host, factory):
    """mujoco_env_test.cpp:227-253: a non-blocking step request returns immediately and completes later."""
    m = empty()
    env = start(host, factory, m, {"unpause": False})
    assert env.step(50, blocking=False)
    assert wait(lambda: env.setting("env_steps_request") == 0)
    assert abs(env.get_field("time")[0] - 50 * m["timestep"][0]) < 1e-6
    env.shutdown()


def test_threads_start_and_stop(host, factory):
    """mujoco_env_test.cpp:277-310."""
    env = start(host, factory, empty(), {"unpause": False})
    assert env.physics_running == 1 and env.event_running == 1
    env.shutdown()
    assert env.physics_running == 0 and env.event_running == 0


def test_steps_terminate(host, factory):
    """mujoco_env_test.cpp:390-426: num_steps=100 => the physics loop exits after exactly 100 steps."""
    m = pendulum()
    env = start(host, factory, m, {"num_steps": 100, "unpause": True})
    assert wait(lambda: env.pending_steps == 0, timeout=20)
    assert wait(lambda: env.physics_running == 0, timeout=5)
    assert abs(env.get_field("time")[0] - 100 * m["timestep"][0]) <= 0.1 * m["timestep"][0]
    env.shutdown()


def test_realtime_pacing_without_sim_time(host, factory):
    """use_sim_time only gates the /clock publication (mujoco_env.cpp:701-703); the loop keeps pacing on data_->time
    (:466-467, :536, :560).  With realtime_index 1 (100 % real time) the simulation must track the wall clock -- not
    free-run -- and the /clock value must stay untouched."""
    m = empty()
    env = start(host, factory, m, {"unpause": True, "use_sim_time": False, "realtime_index": 1})
    t0 = time.time()
    time.sleep(0.5)
    sim, wall = env.data_time, time.time() - t0
    assert env.sim_time == 0.0
    assert 0.2 * wall < sim < 1.5 * wall + 0.05, f"sim {sim:.3f} s vs wall {wall:.3f} s: the loop is not paced"
    env.shutdown()
    # the same run with the clock on publishes the time it paces on
    env = start(host, factory, m, {"unpause": True, "use_sim_time": True, "realtime_index": 1})
    time.sleep(0.2)
    assert env.sim_time > 0 and abs(env.sim_time - env.data_time) < 0.1
    env.shutdown()


def test_manual_steps(host, factory):
    """mujoco_env_test.cpp:428-481 incl. "pending manual steps should not change in unpaused mode"."""
    m = pendulum()
    env = start(host, factory, m, {"unpause": False})
    assert env.setting("env_steps_request") == 0 and env.setting("run") == 0
    assert env.get_field("time")[0] == 0
    env.set_setting("env_steps_request", 1)
    assert wait(lambda: env.setting("env_steps_request") == 0, 1.0)
    assert env.get_field("time")[0] == m["timestep"][0]
    env.set_setting("run", 1)
    env.set_setting("env_steps_request", 100)
    time.sleep(0.01)
    assert env.setting("env_steps_request") == 100
    env.set_setting("env_steps_request", 0)
    env.set_setting("run", 0)
    time.sleep(0.05)
    t = env.get_field("time")[0]
    env.set_setting("env_steps_request", 100)
    assert wait(lambda: env.setting("env_steps_request") == 0)
    assert abs(env.get_field("time")[0] - (t + 100 * m["timestep"][0])) <= 0.1 * m["timestep"][0]
    env.shutdown()


def test_reset(host, factory):
    """mujoco_env_test.cpp:483-529: reset zeroes time, keeps the pause state, restores qpos/qvel."""
    m = pendulum()
    env = start(host, factory, m, {"unpause": False})
    assert env.step(100)
    assert abs(env.get_field("time")[0] - 100 * m["timestep"][0]) < 1e-6
    env.set_setting("reset_request", 1)
    assert wait(lambda: env.setting("reset_request") == 0)
    assert env.setting("run") == 0 and abs(env.get_field("time")[0]) < 1e-6
    env.set_setting("run", 1)
    env.set_setting("reset_request", 1)
    assert wait(lambda: env.setting("reset_request") == 0)
    assert env.setting("run") == 1
    env.set_setting("run", 0)
    time.sleep(0.05)
    j2 = env.name2id(host.OBJ_JOINT, "joint2")
    assert j2 != -1
    q, v = env.get_field("qpos"), env.get_field("qvel")
    q[m["jnt_qposadr"][j2]] = 0.5
    v[m["jnt_dofadr"][j2]] = 0.1
    env.set_field("qpos", q)
    env.set_field("qvel", v)
    env.reset_request()
    assert wait(lambda: env.setting("reset_request") == 0)
    assert env.get_field("qpos")[m["jnt_qposadr"][j2]] != 0.5
    assert env.get_field("qvel")[m["jnt_dofadr"][j2]] != 0.1
    env.shutdown()


# ------------------------------------------------------------------ ros_interface_test.cpp
def test_default_initial_joint_states(host, factory):
    """ros_interface_test.cpp:263-299: exact default state of pendulum_world.xml."""
    m = pendulum()
    env = start(host, factory, m, {"unpause": False})
    assert env.setting("run") == 0 and env.pending_steps == -1 and abs(env.get_field("time")[0]) < 1e-6
    ids = {n: env.name2id(host.OBJ_JOINT, n) for n in ("balljoint", "joint1", "joint2", "ball_freejoint")}
    assert all(i != -1 for i in ids.values())
    q, v = env.get_field("qpos"), env.get_field("qvel")
    qa = lambda n: m["jnt_qposadr"][ids[n]]
    assert list(q[qa("balljoint"):qa("balljoint") + 4]) == [1.0, 0.0, 0.0, 0.0]
    assert q[qa("joint1")] == 0.0 and q[qa("joint2")] == 0.0
    assert list(q[qa("ball_freejoint"):qa("ball_freejoint") + 7]) == [1.0, 0.0, 0.06, 1.0, 0.0, 0.0, 0.0]
    assert np.all(v == 0)
    # with contacts on the ball drops 1 cm onto the ground plane and stays there; the pendulum never moves
    assert env.step(400)
    q, v = env.get_field("qpos"), env.get_field("qvel")
    assert 0.049 < q[qa("ball_freejoint") + 2] < 0.0501 and np.abs(v).max() < 1e-5
    assert list(q[qa("balljoint"):qa("balljoint") + 4]) == [1.0, 0.0, 0.0, 0.0] and q[qa("joint1")] == 0.0
    env.shutdown()


def test_custom_initial_joint_states(host, factory):
    """ros_interface_test.cpp:301-424: string-valued joint_map entries; quaternions come back normalised
    (within 9e-4 of the configured values); ill-sized entries are ignored."""
    m = pendulum()
    params = {
        "unpause": False,
        "initial_joint_positions/joint_map": {"joint1": "-1.57", "joint2": "-0.66", "balljoint": "1.0 0.0 0.0 0.0",
                                              "ball_freejoint": "2.0 1.0 1.06 0.0 0.707 0.0 0.707", "nonexistent": "1.0",
                                              "joint1_bad": "1 2"},
        "initial_joint_velocities/joint_map": {"joint2": "1.05", "ball_freejoint": "1.0 2.0 3.0 10 20 30",
                                               "balljoint": "5 5"},
    }
    env = start(host, factory, m, params)
    ids = {n: env.name2id(host.OBJ_JOINT, n) for n in ("balljoint", "joint1", "joint2", "ball_freejoint")}
    q, v = env.get_field("qpos"), env.get_field("qvel")
    qa = lambda n: m["jnt_qposadr"][ids[n]]
    da = lambda n: m["jnt_dofadr"][ids[n]]
    assert q[qa("joint1")] == -1.57 and q[qa("joint2")] == -0.66
    np.testing.assert_allclose(q[qa("ball_freejoint"):qa("ball_freejoint") + 3], [2.0, 1.0, 1.06], atol=0)
    np.testing.assert_allclose(q[qa("ball_freejoint") + 3:qa("ball_freejoint") + 7], [0.0, 0.707, 0.0, 0.707], atol=9e-4)
    assert abs(np.linalg.norm(q[qa("ball_freejoint") + 3:qa("ball_freejoint") + 7]) - 1) < 1e-12
    assert v[da("joint2")] == 1.05
    np.testing.assert_allclose(v[da("ball_freejoint"):da("ball_freejoint") + 6], [1, 2, 3, 10, 20, 30])
    assert np.all(v[da("balljoint"):da("balljoint") + 3] == 0)  # wrong count -> ignored
    # reset re-applies the configured initial state (mujoco_env.cpp:252-253)
    assert env.step(10)
    env.reset_request()
    assert wait(lambda: env.setting("reset_request") == 0)
    assert env.get_field("qpos")[qa("joint1")] == -1.57 and abs(env.get_field("time")[0]) < 1e-12
    env.shutdown()


def test_step_action(host, factory):
    """ros_interface_test.cpp:209-261: Step goal of 1 and 100 steps succeeds while paused; is preempted
    (success=false) when the simulation is unpaused."""
    m = pendulum()
    env = start(host, factory, m, {"unpause": False})
    ok, pre = env.step_goal(1)
    assert ok and not pre and abs(env.get_field("time")[0] - m["timestep"][0]) < 1e-12
    ok, pre = env.step_goal(100)
    assert ok and not pre and abs(env.get_field("time")[0] - 101 * m["timestep"][0]) < 1e-6
    env.set_pause(False)
    ok, pre = env.step_goal(100)
    assert not ok and pre
    env.shutdown()


def test_pause_shutdown_reset_services(host, factory):
    """ros_interface_test.cpp:100-207."""
    m = pendulum()
    env = start(host, factory, m, {"unpause": False})
    assert env.set_pause(False) and env.setting("run") == 1
    assert wait(lambda: env.get_field("time")[0] > 0)
    assert env.set_pause(True) and env.setting("run") == 0
    time.sleep(0.05)
    env.reset_request()
    assert wait(lambda: env.setting("reset_request") == 0)
    assert abs(env.get_field("time")[0]) < 1e-12 and env.setting("run") == 0
    env.set_setting("exit_request", 1)
    assert wait(lambda: env.physics_running == 0 and env.event_running == 0)
    env.shutdown()


# ------------------------------------------------------------------ mujoco_ros_plugin_test.cpp
PLUGINS = [{"type": "mujoco_ros/TestPlugin", "example_param": "example_string",
            "nested_array_param_1": [{"nested_array_param_2": 1}], "nested_struct_param_1": {"nested_struct_param_2": 2}}]


def test_plugin_callbacks_fire(host, factory):
    """mujoco_ros_plugin_test.cpp:97-121 (control / passive / lastStage on one step), :123-128 (onGeomChanged),
    :165-180 (nested config reaches the plugin)."""
    m = pendulum()
    env = start(host, factory, m, {"unpause": False, "MujocoPlugins": PLUGINS})
    assert env.num_plugins == 1 and env.num_cb_ready_plugins == 1
    for f in ("got_config_param", "got_lvl1_nested_array", "got_lvl2_nested_array", "got_lvl1_nested_struct",
              "got_lvl2_nested_struct"):
        assert env.plugin_flag(0, f) == 1
    for f in ("ran_control_cb", "ran_passive_cb", "ran_last_cb"):
        assert env.plugin_flag(0, f) == 0
    assert env.step(1)
    for f in ("ran_control_cb", "ran_passive_cb", "ran_last_cb"):
        assert env.plugin_flag(0, f) == 1
    assert env.plugin_flag(0, "ran_render_cb") == 0  # no offscreen camera -> no render hand-off
    env.notify_geom_changed(0)
    assert env.plugin_flag(0, "ran_on_geom_changed_cb") == 1
    env.reset_request()
    assert wait(lambda: env.setting("reset_request") == 0)
    assert env.plugin_flag(0, "ran_reset") == 1
    env.shutdown()


def test_failed_plugin_load_is_skipped(host, factory):
    """mujoco_ros_plugin_test.cpp:182-319: a plugin whose load() fails stays registered but is never called
    back nor reset; it recovers on the next reload."""
    m = pendulum()
    env = start(host, factory, m, {"unpause": False, "MujocoPlugins": PLUGINS, "should_fail": True})
    assert env.num_plugins == 1 and env.num_cb_ready_plugins == 0
    assert env.plugin_flag(0, "should_fail") == 1
    assert env.step(1)
    for f in ("ran_control_cb", "ran_passive_cb", "ran_last_cb"):
        assert env.plugin_flag(0, f) == 0
    env.reset_request()
    assert wait(lambda: env.setting("reset_request") == 0)
    assert env.plugin_flag(0, "ran_reset") == 0
    # reload with should_fail cleared -> plugin works again
    env.set_param("should_fail", False)
    env.queue_model(m, nenv=1, backend_factory=factory)
    assert wait(lambda: env.operational_status == 0)
    assert env.num_plugins == 1 and env.num_cb_ready_plugins == 1
    assert env.step(1)
    assert env.plugin_flag(0, "ran_control_cb") == 1
    env.shutdown()


def test_control_only_plugin_chains_split_steps(host, factory):
    """A plugin that declares control / passive callbacks only ("callbacks": "control": ros_control's shape) lets the runtime chain
    consecutive split steps (one launch per step between two callback rounds, mjb_step21_prefix; backends without it split as
    before): every step still delivers one control callback per callback env, the step requests are counted as before, and the
    trajectory equals the all-callbacks plugin's, whose steps are never chained."""
    m = mjcf.load_asset("franka_like")
    nenv, K = 5, 23
    cfg = {"type": "mujoco_ros/TestPlugin", "ctrl_bias": 1.5, "passive_bias": -0.125}
    chained = start(host, factory, m, {"unpause": False, "MujocoPlugins": [dict(cfg, callbacks="control")]}, nenv=nenv)
    plain = start(host, factory, m, {"unpause": False, "MujocoPlugins": [cfg]}, nenv=nenv)
    for env in (chained, plain):
        env.set_callback_envs(3)
        assert env.step(K)
        assert env.plugin_flag(0, "control_calls") == 3 * K
    assert not chained.plugin_flag(0, "ran_last_cb") and plain.plugin_flag(0, "ran_last_cb")
    for e in range(nenv):
        np.testing.assert_allclose(chained.get_field("qpos", e), plain.get_field("qpos", e), rtol=0, atol=1e-10)
        np.testing.assert_allclose(chained.get_field("qvel", e), plain.get_field("qvel", e), rtol=0, atol=1e-9)
        assert abs(chained.get_field("time", e)[0] - K * m["timestep"][0]) < 1e-12
    # a one-step request after the burst (nothing to chain with) and a reset leave the runtime in step
    assert chained.step(1) and plain.step(1)
    np.testing.assert_allclose(chained.get_field("qpos", 0), plain.get_field("qpos", 0), rtol=0, atol=1e-10)
    for x in (chained, plain):
        x.shutdown()


def test_xfrc_written_mid_run_with_a_callback_prefix(host, factory):
    """A control callback that starts writing xfrc_applied in mid-run flips the engine's launch parameters (use_xfrc: the fused
    launches move from the compact to the full frame) while -- with only a PREFIX of the envs carrying callbacks -- the fused launch
    of the other envs of the same step may still be in flight on its own stream: the upload has to wait for it (ADVICE r03:
    mjb_api.hip sync_params vs mjb_step_rest).  The trajectory must equal the run in which every env is a callback env (no rest
    launch exists there), env by env for the callback envs, and the untouched run for the others."""
    m = mjcf.load_asset("franka_like")
    nenv, ncb, K = 6, 2, 12
    t_push = 5 * m["timestep"][0]
    plug = [{"type": "mujoco_ros/TestPlugin", "xfrc_time": t_push, "xfrc_z": 30.0, "xfrc_body": 4}]
    pre = start(host, factory, m, {"unpause": False, "MujocoPlugins": plug}, nenv=nenv)
    full = start(host, factory, m, {"unpause": False, "MujocoPlugins": plug}, nenv=nenv)
    bare = start(host, factory, m, {"unpause": False}, nenv=nenv)
    pre.set_callback_envs(ncb)
    for env in (pre, full, bare):
        for e in range(nenv):
            env.set_field("qvel", 0.1 * (e + 1) * np.ones(m["nv"]), env=e)
        for _ in range(K):   # (one-step requests: every step is a split step of its own, the rest launch runs beside the callbacks)
            assert env.step(1)
    for e in range(nenv):
        want = full if e < ncb else bare
        np.testing.assert_allclose(pre.get_field("qpos", e), want.get_field("qpos", e), rtol=0, atol=1e-12)
        np.testing.assert_allclose(pre.get_field("qvel", e), want.get_field("qvel", e), rtol=0, atol=1e-11)
    assert np.abs(full.get_field("qvel", 0) - bare.get_field("qvel", 0)).max() > 1e-4   # the push does act
    for x in (pre, full, bare):
        x.shutdown()


def test_rk4_fires_control_and_passive_callbacks_at_every_evaluation(host, factory):
    """mj_RungeKutta runs mj_forwardSkip -- and mjcb_passive / mjcb_control inside it -- once per evaluation, four times a step, and
    that is the stated reason lastStageCallback exists (mujoco_ros/include/mujoco_ros/plugin_utils.h:119-125: "called at the end of a
    full step, never inside integrator sub-steps").  With <option integrator="RK4"> the runtime cuts the step's second half at the
    evaluations: 4 control + 4 passive + 1 lastStage call per env and step; what the callbacks write is used by every evaluation
    (the same forces applied by hand give the same trajectory); the callback envs' step equals the other envs' fused RK4 step when
    the plugin writes nothing."""
    m = mjcf.load_asset("franka_like", override={"integrator": "RK4"})
    nenv, ncb, K = 4, 3, 5
    plug = [{"type": "mujoco_ros/TestPlugin", "ctrl_bias": 0.75, "passive_bias": -0.125}]
    env = start(host, factory, m, {"unpause": False, "MujocoPlugins": plug}, nenv=nenv)
    hand = start(host, factory, m, {"unpause": False}, nenv=nenv)
    idle = start(host, factory, m, {"unpause": False, "MujocoPlugins": [{"type": "mujoco_ros/TestPlugin"}]}, nenv=nenv)
    env.set_callback_envs(ncb)
    idle.set_callback_envs(ncb)
    for x in (env, hand, idle):
        for e in range(nenv):
            x.set_field("qvel", 0.2 * (e + 1) * np.ones(m["nv"]), env=e)
    for e in range(nenv):   # ctrl += bias at each of the four evaluations of step k: 4 k + 1 .. 4 k + 4 times the bias -- by hand below
        hand.set_field("qfrc_applied", np.full(m["nv"], -0.125), env=e)
    assert env.step(K) and idle.step(K)
    assert env.plugin_flag(0, "control_calls") == 4 * ncb * K
    assert env.plugin_flag(0, "passive_calls") == 4 * ncb * K
    assert env.plugin_flag(0, "last_calls") == ncb * K
    assert abs(env.get_field("time", 0)[0] - K * m["timestep"][0]) < 1e-12
    # an idle plugin: the callback envs' cut step == the fused step the other envs (and a plugin-free run) take
    bare = start(host, factory, m, {"unpause": False}, nenv=nenv)
    for e in range(nenv):
        bare.set_field("qvel", 0.2 * (e + 1) * np.ones(m["nv"]), env=e)
    assert bare.step(K)
    for e in range(nenv):
        np.testing.assert_allclose(idle.get_field("qpos", e), bare.get_field("qpos", e), rtol=0, atol=1e-13)
    # the biased plugin: ctrl accumulates one bias per evaluation (the view's ctrl is the state field the next evaluation starts from)
    ctrl_end = env.get_field("ctrl", 0)
    np.testing.assert_allclose(ctrl_end, 0.75 * 4 * K)
    assert np.abs(env.get_field("qvel", 0) - bare.get_field("qvel", 0)).max() > 1e-4
    np.testing.assert_allclose(env.get_field("qvel", nenv - 1), bare.get_field("qvel", nenv - 1), rtol=0, atol=1e-13)   # not a callback env
    for x in (env, hand, idle, bare):
        x.shutdown()


def test_plugin_data_contract_batched(host, factory):
    """Callbacks run once per env instance per step, see that instance's view, and what they write is used:
    ctrl written in controlCallback drives the actuators, qfrc_passive is ADDED to (plugin_utils.h:91-107)."""
    m = mjcf.load_asset("franka_like")
    nenv = 3
    plug = [{"type": "mujoco_ros/TestPlugin", "ctrl_bias": 2.5, "passive_bias": -0.25}]
    env = start(host, factory, m, {"unpause": False, "MujocoPlugins": plug}, nenv=nenv)
    ref = start(host, factory, m, {"unpause": False}, nenv=nenv)
    assert env.step(1) and ref.step(1)
    assert env.plugin_flag(0, "control_calls") == nenv and env.plugin_flag(0, "last_env") == nenv - 1
    # reference run: same forces applied by hand
    ref2 = start(host, factory, m, {"unpause": False}, nenv=nenv)
    for e in range(nenv):
        ref2.set_field("ctrl", np.full(m["nu"], 2.5), env=e)
        ref2.set_field("qfrc_applied", np.full(m["nv"], -0.25), env=e)  # same generalized force as the passive add
    assert ref2.step(1)
    for e in range(nenv):
        np.testing.assert_allclose(env.get_field("qvel", e), ref2.get_field("qvel", e), rtol=0, atol=1e-12)
        assert np.abs(env.get_field("qvel", e) - ref.get_field("qvel", e)).max() > 1e-6
        np.testing.assert_allclose(env.get_field("ctrl", e), 2.5)
    for x in (env, ref, ref2):
        x.shutdown()
