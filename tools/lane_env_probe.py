#!/usr/bin/env python3
"""lane_env_probe.py [envs] -- where a step of the lane = env kernel's two-wavefront forms spends its cycles: run on the measurement build
(tools/lane_env_probe_build.sh -> csrc/libmjb_xprobe.so, -DMJB_LE_PROBE: s_memtime stamps around the step's phases, summed over the launch and
written over env 0's sensordata).  Per step and wavefront: [sweep root->leaf (own work), rendezvous F, waiting at the per-body barriers (+ V: the
force block), sweep leaf->root, energy / factors, rendezvous A, solves + Euler (V: waiting for them), rendezvous B].
    MJB_LANE_ENV_DUO=2 MJB_LIBRARY=$PWD/mujoco_ros_pkgs_amd/csrc/libmjb_xprobe.so python tools/lane_env_probe.py 4096"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mujoco_ros_pkgs_amd import engine, mjcf
from tests.conftest import random_franka_state
model = mjcf.load_asset("franka_like")
cm = engine.CompiledModel(model)
nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
qpos, qvel = random_franka_state(model, nenv, 0)
b = engine.Batch(cm, nenv)
b.set_lane_env(1)
b.set("qpos", qpos); b.set("qvel", qvel)
b.set_ctrl_noise(5.0, 0.1, 12345, 0)
b.step(1000)
b.step(1000)
sd = b.get("sensordata")[0]
print("P:", np.round(sd[0:8]).astype(int), int(sd[0:8].sum()))
print("V:", np.round(sd[8:16]).astype(int), int(sd[8:16].sum()))
