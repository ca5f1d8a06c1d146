#!/usr/bin/env python
"""Per-launch kernel time of consecutive fused launches of one BASELINE workload (does the cost drift with the state?).
usage: python tools/launch_series.py [--config 3] [--launches 10]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import CONFIG_MODEL, WORKLOADS, initial_state  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=3)
ap.add_argument("--launches", type=int, default=10)
ap.add_argument("--substeps", type=int, default=0)
args = ap.parse_args()
from mujoco_ros_pkgs_amd import engine, mjcf  # noqa: E402
name = CONFIG_MODEL[args.config]
label, noise_std, E, S, _ = WORKLOADS[name]
S = args.substeps or S
model = mjcf.Model(dict(mjcf.load_asset(name)))
cm = engine.CompiledModel(model)
b = engine.Batch(cm, E)
qpos, qvel = initial_state(name, model, E, seed=1000)
b.set("qpos", qpos)
b.set("qvel", qvel)
b.set_ctrl_noise(noise_std, 0.1, 12345, 0)
for i in range(args.launches):
    t0 = time.perf_counter()
    ms = b.time_steps(S, 1)
    wall = 1e3 * (time.perf_counter() - t0)
    nefc = b.get("nefc") if False else None
    print(f"launch {i}: kernel {ms:8.3f} ms  wall {wall:8.3f} ms  -> {E * S / ms / 1e3:7.2f} M env-steps/s  (steps {i * S} .. {(i + 1) * S})")
