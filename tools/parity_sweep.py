#!/usr/bin/env python3
"""Wide randomised parity sweep (not part of the test suite: minutes of CPU oracle time): many random states of the contact
scenes under every solver / cone combination, N steps on the GPU against the oracle, worst absolute state error per case.
usage: tools/parity_sweep.py [--envs 256] [--steps 30]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mujoco_ros_pkgs_amd import engine, mjcf, workloads  # noqa: E402
from oracle import pyoracle  # noqa: E402
from test_gpu_contact import scenario_states  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=256)
ap.add_argument("--steps", type=int, default=30)
args = ap.parse_args()
pyoracle.build()
CASES = [("franka_table", {"solver": s, "cone": c}) for s in ("PGS", "Newton", "CG") for c in ("pyramidal", "elliptic")]
# (PGS with elliptic cone blocks holds at most 64 rows: the table-size njmax of 73 is for the pyramidal asset)
CAP = {("PGS", "elliptic"): 64}
CASES += [("shadow_hand_like", {"solver": s}) for s in ("Newton", "CG")]
CASES += [("shadow_hand_grasp", {"solver": "Newton"})]
# the lane = env kernel (round 5): both compiled-in topologies, noise on, every LDS instantiation (4096 / 20000 / 40000 envs), sampled against the oracle
for name, n_le, k_le in (("franka_like", 4096, 300), ("franka_like", 20000, 100), ("franka_like", 40000, 100), ("lane_env_tree", 4096, 300)):
    m = mjcf.load_asset(name)
    rng = np.random.default_rng(7)
    qpos = np.tile(np.asarray(m["qpos0"], dtype=np.float64), (n_le, 1)) + rng.uniform(-0.5, 0.5, (n_le, m["nq"])) * np.where(np.asarray(m["jnt_type"]) == 3, 1.0, 0.04)
    qvel = rng.uniform(-1, 1, (n_le, m["nv"]))
    std = 20.0 if name == "franka_like" else 1.0
    b = engine.Batch(engine.CompiledModel(m), n_le)
    b.set_lane_env(1)
    b.set("qpos", qpos)
    b.set("qvel", qvel)
    b.set_ctrl_noise(std, 0.1, 4242, 0)
    b.step(k_le)
    gq, gv = b.get("qpos"), b.get("qvel")
    idx = np.unique(np.concatenate([np.arange(0, n_le, max(1, n_le // 512)), [n_le - 1]]))
    eq, ev = [], []
    for e in idx:  # (the global env id keys the Philox stream: one oracle rollout per sampled env)
        oq, ov, _ = pyoracle.rollout(m, qpos[e:e + 1], qvel[e:e + 1], k_le, noise_std=std, noise_rate=0.1, seed=4242, env_offset=int(e))
        eq.append(np.abs(gq[e] - oq[0]).max())
        ev.append(np.abs(gv[e] - ov[0]).max())
    print(json.dumps({"model": name, "kernel": "lane = env" if b.lane_env_info()[1] else "generic", "envs": n_le, "sampled": int(len(idx)), "steps": k_le, "auto_resets": b.warning_count(),
                      "max_err_qpos": float(np.max(eq)), "p99_err_qpos": float(np.percentile(eq, 99)), "max_err_qvel": float(np.max(ev)), "p99_err_qvel": float(np.percentile(ev, 99))}), flush=True)
    b.close()
for name, over in CASES:
    cap = CAP.get((over.get("solver"), over.get("cone")))
    m = mjcf.compile_xml_file(os.path.join(mjcf.ASSET_DIR, name + ".xml"), override=over, **({"nefcmax": cap} if cap else {}))
    n = args.envs if name == "franka_table" else max(16, args.envs // 8)
    if name == "franka_table":
        qpos, qvel = scenario_states(m, n, seed=123)
    else:
        qpos, qvel = (workloads.hand_power_grasp_states if name == "shadow_hand_grasp" else workloads.hand_grasp_states)(m, n, seed=123)
    b = engine.Batch(engine.CompiledModel(m), n)
    b.set("qpos", qpos)
    b.set("qvel", qvel)
    b.step(args.steps)
    gq, gv = b.get("qpos"), b.get("qvel")
    oq, ov, _ = pyoracle.rollout(m, qpos, qvel, args.steps, nthreads=os.cpu_count() or 1)
    eq, ev = np.abs(gq - oq).max(axis=1), np.abs(gv - ov).max(axis=1)
    print(json.dumps({"model": name, **over, "envs": n, "steps": args.steps, "auto_resets": b.warning_count(),
                      "max_err_qpos": float(eq.max()), "p99_err_qpos": float(np.percentile(eq, 99)),
                      "max_err_qvel": float(ev.max()), "p99_err_qvel": float(np.percentile(ev, 99))}))
    b.close()
