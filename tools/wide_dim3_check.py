#!/usr/bin/env python3
"""wide_dim3_check.py -- a Newton / elliptic model whose cones are ALL of dimension 3, with a row capacity beyond 128 (the kernel variant with fused frames that lay the
cone blocks out by row), stepped on the fused frame against the full frame.  Run under MJB_WIDE_FRAME=1 for the wide frame (two rows per lane: the line search parks ten
constants per contact in the contact's block -- nine doubles apart at three rows of stride three)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from mujoco_ros_pkgs_amd import engine, mjcf
from test_gpu_contact import scenario_states
m = mjcf.compile_xml_file(os.path.join(mjcf.ASSET_DIR, "franka_table.xml"), override={"solver": "Newton", "cone": "elliptic"}, nefcmax=160, nconmax=32)
cm = engine.CompiledModel(m)
n = 128
qpos, qvel = scenario_states(m, n, seed=5)
A, B = engine.Batch(cm, n), engine.Batch(cm, n)
B.set_keep_frame(True)
for b in (A, B):
    b.set("qpos", qpos); b.set("qvel", qvel)
worst = 0.0
for s in range(40):
    for k in ("qpos", "qvel", "qacc_warmstart", "time"):
        A.set(k, B.get(k))
    A.step(1); B.step(1)
    worst = max(worst, float(np.abs(A.get("qvel") - B.get("qvel")).max()))
nefc = B.get("nefc")[:, 0]
print(f"fused frame id {A.fused_frame()[0]} ({A.fused_frame()[1]} B): worst |dqvel| fused vs full over 40 steps x {n} envs: {worst:.3e}; rows at the end: mean {nefc.mean():.1f} max {nefc.max()}; resets {A.warning_count()} / {B.warning_count()}")
sys.exit(0 if worst <= 1e-11 else 1)
