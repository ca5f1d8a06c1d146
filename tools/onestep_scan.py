#!/usr/bin/env python3
"""onestep_scan.py <model> <solver> <cone> [envs] [steps] -- where a rollout's error against the oracle comes from: the parity sweep's states (tests/test_gpu_contact.py
scenario_states), stepped one step at a time; before every step EVERY env's state (qpos, qvel, qacc_warmstart) is copied into the oracle, which takes the same step.
Prints the env-steps whose one-step error is out of line, with the oracle's view of the constraint set and the solver's iteration counts on both sides."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from mujoco_ros_pkgs_amd import engine, mjcf
from oracle import pyoracle
from test_gpu_contact import scenario_states

name, solver, cone = sys.argv[1], sys.argv[2], sys.argv[3]
n = int(sys.argv[4]) if len(sys.argv) > 4 else 256
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 30
m = mjcf.compile_xml_file(os.path.join(mjcf.ASSET_DIR, name + ".xml"), override={"solver": solver, "cone": cone})
pyoracle.build()
if name == "franka_table":
    qpos, qvel = scenario_states(m, n, seed=123)
else:
    from mujoco_ros_pkgs_amd import workloads
    qpos, qvel = (workloads.hand_power_grasp_states if name == "shadow_hand_grasp" else workloads.hand_grasp_states)(m, n, seed=123)
b = engine.Batch(engine.CompiledModel(m), n)
b.set_keep_frame(bool(os.environ.get("KEEP_FRAME")))
b.set("qpos", qpos); b.set("qvel", qvel)
d = pyoracle.OracleData(m, fast=False)
worst = []
for s in range(steps):
    st = {k: b.get(k) for k in ("qpos", "qvel", "qacc_warmstart")}
    b.step(1)
    gq, gv = b.get("qpos"), b.get("qvel")
    git = b.get("solver_iter")[:, 0].astype(int) if os.environ.get("KEEP_FRAME") else None
    for e in range(n):
        d.reset()
        d.qpos[:] = st["qpos"][e]; d.qvel[:] = st["qvel"][e]; d.qacc_warmstart[:] = st["qacc_warmstart"][e]
        d.step()
        ev = float(np.abs(gv[e] - d.qvel).max())
        if ev > float(os.environ.get("SCAN_TOL", "1e-10")):
            nc = int(d.ncon[0])
            print(f"step {s} env {e}: |dqvel| {ev:.2e} |dqpos| {np.abs(gq[e] - d.qpos).max():.2e}  oracle: ncon {nc} nefc {int(d.nefc[0])} iters {int(d.solver_iter[0])}"
                  + (f" (GPU iters {git[e]})" if git is not None else "") + f" dims {np.array(d.contact_dim[:nc]).astype(int).tolist()} max|qacc| {np.abs(d.qacc).max():.2e}", flush=True)
        worst.append(ev)
print(f"{name} {solver} {cone}: {n} envs x {steps} steps, one-step |dqvel|: max {max(worst):.2e} p99 {np.percentile(worst, 99):.2e} median {np.median(worst):.2e}")
