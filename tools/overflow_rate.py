#!/usr/bin/env python3
"""Contact / constraint-row overflow of the contact bench workloads: mjWARN_CONTACTFULL / mjWARN_CNSTRFULL events
per env-step over a 12 000-step rollout of bench.py's config 3 (and 5), launch by launch, with the ncon / nefc distribution at the
launch ends.  The capacities are the SURVEY.md §8 model table's (config 3: 16 contacts, 64 + 9 rows): they are what lets eight
lean frames share a CU's LDS (20 448 B each); this tool measures what that choice drops.
usage: overflow_rate.py [model] [envs] [launches] [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from bench import WORKLOADS, initial_state
from mujoco_ros_pkgs_amd import engine, mjcf

name = sys.argv[1] if len(sys.argv) > 1 else "franka_table"
n = int(sys.argv[2]) if len(sys.argv) > 2 else WORKLOADS[name][2]
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 12
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
m = mjcf.load_asset(name)
cm = engine.CompiledModel(m)
b = engine.Batch(cm, n)
qp, qv = initial_state(name, m, n, 1000)
b.set("qpos", qp)
b.set("qvel", qv)
b.set_ctrl_noise(WORKLOADS[name][1], 0.1, 12345, 0)
print(f"{name}: {n} envs, nconmax {m['nconmax']}, nefcmax {m['nefcmax']}, fused frame {b.lib.mjb_frame_bytes(cm.ptr, 1)} B")
prev_c = prev_r = 0
for it in range(launches):
    b.step(steps)
    c, r = b.warning("contactfull"), b.warning("cnstrfull")
    b.forward()
    nc, ne = b.get("ncon")[:, 0], b.get("nefc")[:, 0]
    print(f"launch {it:2d}: contactfull +{c - prev_c:4d}  cnstrfull +{r - prev_r:4d}   at its end: ncon mean {nc.mean():5.2f} p99 {int(np.percentile(nc, 99)):2d} max {nc.max():2d}"
          f" (== nconmax in {int((nc >= m['nconmax']).sum())} envs)   nefc mean {ne.mean():5.2f} max {ne.max():3d}")
    prev_c, prev_r = c, r
tot = n * steps * launches
print(f"total: {prev_c} contactfull, {prev_r} cnstrfull in {tot} env-steps -> {prev_c / tot:.2e} / {prev_r / tot:.2e} per env-step;"
      f" auto-resets {b.warning_count()}")
