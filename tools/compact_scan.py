#!/usr/bin/env python3
"""compact_scan.py [model] [envs] [steps] -- the fused (compact) frame against the full frame (keep_frame), step by step: batch B runs on the full frame; before
every step its state is copied into batch A, which takes the same step on the fused frame.  The two run the same code on different LDS layouts and must
agree to rounding; prints, by the step's row count, how many env-steps did not."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from bench import WORKLOADS, initial_state
from mujoco_ros_pkgs_amd import engine, mjcf

name = sys.argv[1] if len(sys.argv) > 1 else "shadow_hand_grasp"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
m = mjcf.load_asset(name)
cm = engine.CompiledModel(m)
A, B = engine.Batch(cm, n), engine.Batch(cm, n)
B.set_keep_frame(True)
qp, qv = initial_state(name, m, n, 1000)
for b in (A, B):
    b.set("qpos", qp); b.set("qvel", qv)
    b.set_ctrl_noise(WORKLOADS[name][1], 0.1, 12345, 0)
bad = {}
tot = {}
worst = (0.0, None)
for s in range(steps):
    for k in ("qpos", "qvel", "qacc_warmstart", "ctrlnoise", "time"):
        A.set(k, B.get(k))
    A.step(1); B.step(1)
    nefc = B.get("nefc")[:, 0].astype(int)
    dv = np.abs(A.get("qvel") - B.get("qvel")).max(axis=1)
    for r in np.unique(nefc):
        sel = nefc == r
        tot[r] = tot.get(r, 0) + int(sel.sum())
        nb = int((dv[sel] > 1e-9).sum())
        if nb: bad[r] = bad.get(r, 0) + nb
    if dv.max() > worst[0]: worst = (float(dv.max()), (s, int(dv.argmax()), int(nefc[dv.argmax()])))
print("worst |dqvel| fused vs full:", worst)
print("rows : env-steps with |dqvel| > 1e-9 / env-steps")
for r in sorted(tot):
    if bad.get(r, 0): print(f"  {r:4d} : {bad[r]} / {tot[r]}")
print("env-steps scanned", sum(tot.values()), " off:", sum(bad.values()), " resets A / B:", A.warning_count(), B.warning_count())
