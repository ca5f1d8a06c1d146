"""The host runtime sharding one batch over several devices (include/mjr_host.h mjr_env_queue_model_devices): env block
i lives behind its own backend, the blocks are launched together, and NOTHING observable changes -- trajectories (ctrl-noise stream
keyed by the global env index), per-env field access, reset masks, plugin callbacks over all envs.  CPU: the oracle harness backend
(two blocks); GPU: two HIP batches (both on device 0 of the 1-GPU box -- the code path is the multi-device one)."""
import numpy as np
import pytest

from mujoco_ros_pkgs_amd import mjcf
from test_host_env import factory, host, oracle_factory, start, wait  # noqa: F401 (fixtures)
from test_host_env import PLUGINS


def _run(host, factory, model, nenv, devices, steps, plugins=None):
    params = {"unpause": False, "ctrl_noise_std": 3.0, "ctrl_noise_rate": 0.1}
    if plugins:
        params["MujocoPlugins"] = plugins
    env = host.HostEnv(params)
    env.queue_model(model, nenv=nenv, backend_factory=factory, devices=devices)
    env.start()
    assert wait(lambda: env.operational_status == 0 and env.model_valid), env.load_error
    assert env.nenv == nenv
    rng = np.random.default_rng(3)
    q0 = np.asarray(model["qpos0"], dtype=np.float64)
    for e in range(nenv):
        env.set_field("qpos", q0 + 0.1 * rng.standard_normal(model["nq"]), env=e)
    assert env.step(steps)
    out = np.array([env.get_field("qpos", env=e) for e in range(nenv)]), np.array([env.get_field("qvel", env=e) for e in range(nenv)])
    return env, out


def test_sharded_equals_unsharded(host, factory, franka):
    nenv, steps = 7, 40
    env1, (q1, v1) = _run(host, factory, franka, nenv, None, steps)
    env1.shutdown()
    env3, (q3, v3) = _run(host, factory, franka, nenv, [0, 0, 0], steps)   # blocks of 3 + 2 + 2 envs
    assert np.array_equal(q1, q3) and np.array_equal(v1, v3)
    assert len({tuple(r) for r in q3.round(12)}) == nenv       # the envs really differ (own noise streams, own states)
    # reset request: every block goes back to the initial state
    env3.reset_request()
    assert wait(lambda: env3.setting("reset_request") == 0)
    for e in (0, 3, 6):
        assert np.allclose(env3.get_field("qpos", env=e), franka["qpos0"]) and abs(env3.get_field("time", env=e)[0]) < 1e-15
    env3.shutdown()


def test_sharded_plugin_callbacks_reach_every_block(host, factory, franka):
    nenv = 5
    env, _ = _run(host, factory, franka, nenv, [0, 0], 3, plugins=PLUGINS)
    assert env.num_cb_ready_plugins == 1
    assert env.plugin_flag(0, "ran_control_cb") == 1 and env.plugin_flag(0, "ran_last_cb") == 1
    assert env.plugin_flag(0, "last_env") == nenv - 1            # the callback round ends with the last env of the last block
    env.shutdown()
