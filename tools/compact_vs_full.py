#!/usr/bin/env python3
"""compact_vs_full.py <state.npz> [model] -- one step from a saved env state (tools/replay_reset.py writes them) on the fused (compact) frame and
on the full frame (keep_frame), both against the oracle: the two frames run the same code on different LDS layouts and must agree to rounding."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from mujoco_ros_pkgs_amd import engine, mjcf
from oracle import pyoracle

st = np.load(sys.argv[1])
name = sys.argv[2] if len(sys.argv) > 2 else "shadow_hand_grasp"
m = mjcf.load_asset(name)
cm = engine.CompiledModel(m)
pyoracle.build()
d = pyoracle.OracleData(m, fast=False)
d.reset()
d.qpos[:] = st["qpos"]; d.qvel[:] = st["qvel"]; d.qacc_warmstart[:] = st["qacc_warmstart"]; d.ctrl[:] = st["ctrl_step"]; d.time[:] = st["time"]
d.step()
print(f"oracle: ncon {int(d.ncon[0])} nefc {int(d.nefc[0])} iters {int(d.solver_iter[0])} max|qacc| {np.abs(d.qacc).max():.3e}")
n = 64
for keep in (False, True):
    b = engine.Batch(cm, n)
    b.set_keep_frame(keep)
    for k, src in (("qpos", "qpos"), ("qvel", "qvel"), ("qacc_warmstart", "qacc_warmstart"), ("ctrl", "ctrl_step"), ("time", "time")):
        b.set(k, np.tile(st[src][None, :], (n, 1)))
    b.step(1)
    qv, qp = b.get("qvel"), b.get("qpos")
    print(f"{'full frame (keep_frame)' if keep else 'fused (compact) frame  '}: |qvel - oracle| {np.abs(qv - np.array(d.qvel)[None, :]).max():.3e}  |qpos - oracle| {np.abs(qp - np.array(d.qpos)[None, :]).max():.3e}  "
          f"finite {bool(np.isfinite(qv).all())}  spread over the 64 copies {np.abs(qv - qv[0:1]).max():.1e}  resets {b.warning_count()}")
    b.close()
