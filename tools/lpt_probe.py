#!/usr/bin/env python3
"""Does launching the costliest envs first shorten a fused launch?  (an env's K steps are sequential and the heaviest env
of config 3 takes about twice the mean.)  Cost proxy per env = PGS sweeps x rows of its current state; the batch is timed as it is and
after permuting the envs into descending cost order on the host."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mujoco_ros_pkgs_amd import mjcf, engine
from bench import WORKLOADS, initial_state
name = "franka_table"
m = mjcf.load_asset(name); cm = engine.CompiledModel(m); n = 4096
def make(qp, qv, ws=None):
    b = engine.Batch(cm, n); b.set("qpos", qp); b.set("qvel", qv)
    if ws is not None: b.set("qacc_warmstart", ws)
    b.set_ctrl_noise(WORKLOADS[name][1], 0.1, 12345, 0); return b
qp, qv = initial_state(name, m, n, 1000)
b = make(qp, qv); b.step(1000); b.synchronize()
qp, qv, ws = b.get("qpos"), b.get("qvel"), b.get("qacc_warmstart")
b.forward(); cost = b.get("solver_iter")[:, 0].astype(float) * b.get("nefc")[:, 0]
print("cost mean %.0f p90 %.0f p99 %.0f max %.0f" % (cost.mean(), np.percentile(cost, 90), np.percentile(cost, 99), cost.max()))
for label, perm in (("as is", np.arange(n)), ("costliest first", np.argsort(-cost)), ("costliest last", np.argsort(cost))):
    bb = make(qp[perm], qv[perm], ws[perm]); bb.step(1); bb.synchronize()
    ms = min(bb.time_steps(200, 1) for _ in range(1))
    print(f"{label:16s}: {ms:.2f} ms / 200 steps -> {n*200/ms/1e3:.2f} M env-steps/s"); bb.close()
