#!/usr/bin/env python3
"""late_state_parity.py [model] [envs] [warm-up steps] [rounds] [samples] -- one-step parity on the states a workload actually REACHES: the batch runs the bench
workload (OU ctrl noise, fused launches) for the warm-up, then, `rounds` times: the state of `samples` random envs (qpos, qvel, qacc_warmstart, ctrl-noise state,
time) is copied into the oracle, both take ONE step (the batch on its production fused frame, same Philox draw), the results are compared, and the batch runs
another 200 steps.  Complements the tests, which start from synthetic states: rare constraint sets (cones of mixed dimension, many limit rows) only appear late."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from bench import WORKLOADS, initial_state
from mujoco_ros_pkgs_amd import engine, mjcf
from oracle import pyoracle

name = sys.argv[1] if len(sys.argv) > 1 else "shadow_hand_grasp"
n = int(sys.argv[2]) if len(sys.argv) > 2 else WORKLOADS[name][2]
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 10
ns = int(sys.argv[5]) if len(sys.argv) > 5 else 64
m = mjcf.load_asset(name)
cm = engine.CompiledModel(m)
pyoracle.build()
b = engine.Batch(cm, n)
qp, qv = initial_state(name, m, n, 1000)
b.set("qpos", qp); b.set("qvel", qv)
noise = WORKLOADS[name][1]
b.set_ctrl_noise(noise, 0.1, 12345, 0)
done = 0
while done < warm:
    k = min(200, warm - done); b.step(k); done += k
rng = np.random.default_rng(1)
d = pyoracle.OracleData(m, fast=False)
worst = {"qpos": 0.0, "qvel": 0.0}
hist = {}
for r in range(rounds):
    envs = rng.choice(n, size=min(ns, n), replace=False)
    st = {k: b.get(k) for k in ("qpos", "qvel", "qacc_warmstart", "ctrlnoise", "time")}
    b.step(1)
    gq, gv = b.get("qpos"), b.get("qvel")
    for e in envs:
        d.reset()
        d.qpos[:] = st["qpos"][e]; d.qvel[:] = st["qvel"][e]; d.qacc_warmstart[:] = st["qacc_warmstart"][e]; d.ctrlnoise[:] = st["ctrlnoise"][e]; d.time[:] = st["time"][e]
        d.ctrl_noise(noise, 0.1, 12345, int(e), done)
        d.step()
        eq, ev = float(np.abs(gq[e] - d.qpos).max()), float(np.abs(gv[e] - d.qvel).max())
        nefc = int(d.nefc[0]); nc = int(d.ncon[0])
        dims = tuple(sorted(set(np.array(d.contact_dim[:nc]).astype(int).tolist())))
        key = (nefc // 16 * 16, dims)
        h = hist.setdefault(key, [0, 0.0]); h[0] += 1; h[1] = max(h[1], ev)
        worst["qpos"] = max(worst["qpos"], eq); worst["qvel"] = max(worst["qvel"], ev)
        if ev > 1e-8:
            print(f"  round {r} env {e} step {done}: ncon {nc} nefc {nefc} dims {dims} |dqpos| {eq:.2e} |dqvel| {ev:.2e}", flush=True)
    done += 1
    b.step(199); done += 199
print(f"{name}: {rounds} x {len(envs)} env-steps between step {warm} and {done}: worst |dqpos| {worst['qpos']:.2e} |dqvel| {worst['qvel']:.2e}; resets {b.warning_count()}")
for key in sorted(hist):
    print(f"  rows {key[0]:3d}-{key[0] + 15:3d} contact dims {key[1]}: {hist[key][0]} env-steps, worst |dqvel| {hist[key][1]:.2e}")
