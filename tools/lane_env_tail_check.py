#!/usr/bin/env python3
"""lane_env_tail_check.py -- batches whose last block of 64 envs is partly empty (4097, 4100, 8191 envs) on the multi-wavefront forms of the lane = env kernel:
the tail lanes run on the last env's data, every wavefront of the block takes the same barriers; envs on both sides of the tail against the oracle after 50 noise steps."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mujoco_ros_pkgs_amd import engine, mjcf
from oracle import pyoracle
from conftest import random_franka_state
pyoracle.build()
m = mjcf.load_asset("franka_like"); cm = engine.CompiledModel(m)
lib = engine.binding.load_library()
for n in (4100, 4097, 8191):
    for form in (3, 2, 1):
        lib.mjb_lane_env_set_form(form)
        qpos, qvel = random_franka_state(m, n, 3)
        b = engine.Batch(cm, n); b.set_lane_env(1)
        b.set("qpos", qpos); b.set("qvel", qvel); b.set_ctrl_noise(10.0, 0.1, 7, 0)
        b.step(50)
        q = b.get("qpos"); worst = 0
        for e in (0, 4095, 4096, n - 1):
            oq, _, _ = pyoracle.rollout(m, qpos[e:e+1], qvel[e:e+1], 50, noise_std=10.0, noise_rate=0.1, seed=7, env_offset=int(e))
            worst = max(worst, float(np.abs(q[e] - oq[0]).max()))
        print(n, "form requested", form, "ran", lib.mjb_lane_env_last_form(), "worst |dqpos| vs oracle", worst, "finite", bool(np.isfinite(q).all()))
        b.close()
lib.mjb_lane_env_set_form(-1)
