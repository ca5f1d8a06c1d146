"""Measured GPU-vs-oracle error of the 50-step contact rollout of tests/test_gpu_contact.py, per solver / cone (what its bounds are
set against)."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from mujoco_ros_pkgs_amd import engine, mjcf
from oracle import pyoracle as po
from test_gpu_contact import scenario_states
for solver, cone in (("PGS", "pyramidal"), ("Newton", "pyramidal"), ("Newton", "elliptic")):
    model = mjcf.Model(dict(mjcf.load_asset("franka_table")))
    model["solver"] = {"PGS": 0, "Newton": 2}[solver]; model["cone"] = {"pyramidal": 0, "elliptic": 1}[cone]
    cm = engine.CompiledModel(model)
    nenv = 32
    qpos, qvel = scenario_states(model, nenv, seed=3)
    b = engine.Batch(cm, nenv); b.set("qpos", qpos); b.set("qvel", qvel); b.set_ctrl_noise(3.0, 0.1, 12345, 0)
    b.step(50)
    oq, ov, _ = po.rollout(model, qpos, qvel, 50, noise_std=3.0, noise_rate=0.1, seed=12345)
    eq = (np.abs(b.get("qpos") - oq) / (1 + np.abs(oq))).max(); ev = (np.abs(b.get("qvel") - ov) / (1 + np.abs(ov))).max()
    print(solver, cone, "50 steps: qpos %.2e qvel %.2e" % (eq, ev))
