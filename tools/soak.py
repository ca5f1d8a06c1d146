#!/usr/bin/env python3
"""Long rollouts of the BASELINE workloads on one GPU: env-steps, auto-reset (mj_check*) count, finiteness of the final
state and spread of a few observables.  usage: tools/soak.py [--launches N]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mujoco_ros_pkgs_amd import engine, mjcf  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--launches", type=int, default=50)
args = ap.parse_args()
CASES = [("franka_like", {}, 4096, 1000, 43.5), ("franka_like", {}, 65536, 250, 43.5), ("franka_table", {}, 4096, 200, 43.5),
         ("franka_table", {"solver": "Newton"}, 4096, 200, 43.5), ("shadow_hand_like", {}, 1024, 100, 1.0), ("shadow_hand_grasp", {}, 1024, 100, 1.0)]
for name, over, nenv, K, std in CASES:
    model = mjcf.load_asset(name) if not over else mjcf.compile_xml_file(os.path.join(mjcf.ASSET_DIR, name + ".xml"), override=over)
    noise = bench.WORKLOADS.get(name, (name, std, nenv))[1]
    cm = engine.CompiledModel(model)
    b = engine.Batch(cm, nenv)
    qpos, qvel = bench.initial_state(name, model, nenv, seed=77)
    b.set("qpos", qpos)
    b.set("qvel", qvel)
    b.set_ctrl_noise(noise, 0.1, 12345, 0)
    t0 = time.perf_counter()
    for _ in range(args.launches):
        b.step(K)
    b.synchronize()
    dt = time.perf_counter() - t0
    q, v = b.get("qpos"), b.get("qvel")
    out = {"model": name, "override": over, "envs": nenv, "steps_per_env": K * args.launches,
           "env_steps": nenv * K * args.launches, "env_steps_per_s": nenv * K * args.launches / dt,
           "auto_resets": b.warning_count(), "finite": bool(np.isfinite(q).all() and np.isfinite(v).all()),
           "max_abs_qvel": float(np.abs(v).max()), "sim_time_s": float(b.get("time")[0, 0]), "lane_env_kernel": b.lane_env_info()[1]}
    print(json.dumps(out))
    b.close()
