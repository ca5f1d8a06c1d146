#!/usr/bin/env python3
"""Per-stage shader-cycle breakdown of the step kernel (profiling build libmjb_prof.so, env 0 / lane 0).
Usage: python tools/profile_stages.py [--lanes G] [--envs E] [--steps K]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mujoco_ros_pkgs_amd import binding  # noqa: E402

STAGES = ["kinematics", "com_pos", "crb", "factorM", "transm+sens_pos", "com_vel", "passive", "rne(+aref)", "sens_vel",
          "actuation", "acceleration", "constraint(PGS)", "sens_acc", "ctrl_noise", "forward(total)", "euler",
          "collision", "make_constraint", "project(B)", "-", "kin.A local poses", "kin.B chain", "kin.C normalise+inertial", "kin.D joints/geoms/sites",
          "nwt.setup (M, rows)", "nwt.warmstart", "nwt.grad+update /iter", "nwt.H (MFMA) /iter", "nwt.solve /iter", "nwt.linesearch /iter", "nwt.(loop overhead)", "nwt.chol /iter"]

ap = argparse.ArgumentParser()
ap.add_argument("--lanes", type=int, default=0)
ap.add_argument("--epb", type=int, default=0)
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--model", default="franka_like")
ap.add_argument("--solver", default="")
ap.add_argument("--sub", action="store_true", help="libmjb_prof_sub.so: slots 24-29 = collision / make_constraint sub-stages (PGS runs)")
a = ap.parse_args()

binding.LIB_PATH = os.environ.get("MJB_PROF_LIB") or os.path.join(ROOT, "mujoco_ros_pkgs_amd", "csrc", "libmjb_prof_sub.so" if a.sub else "libmjb_prof.so")
if a.sub:
    STAGES[19] = "pgs.setup (B row, b, warmstart)"
    STAGES[21], STAGES[22] = "pgs.sweeps per step [count, not cycles]", "pgs.rows per step [count, not cycles]"
    STAGES[30], STAGES[31] = "pgs.AR build", "pgs.warm residual + sweeps"
    STAGES[24:30] = ["col.cull+narrowphase", "col.offsets+params+stores", "mk.count+cut", "mk.row params (pass 2)", "mk.D + equality J",
                     "mk.contact J"]
from mujoco_ros_pkgs_amd import engine, mjcf  # noqa: E402

model = mjcf.load_asset(a.model)
if a.solver:
    model = mjcf.Model(dict(model))
    model["solver"] = {"PGS": 0, "Newton": 2}[a.solver]
cm = engine.CompiledModel(model)
b = engine.Batch(cm, a.envs)
b.set_launch(a.lanes, a.epb)
rng = np.random.default_rng(0)
from bench import WORKLOADS, initial_state  # noqa: E402
qp, qv = initial_state(a.model, model, a.envs, 1000)
b.set("qpos", qp)
b.set("qvel", qv)
b.set_ctrl_noise(WORKLOADS.get(a.model, ("", 1.0, 0))[1], 0.1, 12345, 0)
print("frame bytes:", 8 * b.lib.mjb_frame_doubles(cm.ptr) if hasattr(b.lib, "mjb_frame_doubles") else "?")
b.step(a.steps)
b.synchronize()
out = (C.c_uint64 * 64)()
b.lib.mjb_debug_profile(b.ptr, out, 1)
ms = b.time_steps(a.steps, 2)
b.lib.mjb_debug_profile(b.ptr, out, 1)
tot = 0
print(f"lanes={a.lanes} envs={a.envs} steps={a.steps}: {ms:.3f} ms/launch -> {a.envs*a.steps/ms/1e3:.1f} M env-steps/s")
for i, n in enumerate(STAGES):
    cnt = max(1, out[32 + i])
    cyc = out[i] / cnt
    if n != "forward(total)":
        tot += cyc
    print(f"  {n:18s} {cyc:10.0f} cycles/call  ({out[32+i]} calls)")
print(f"  sum (excl. total, incl. kin.* sub-phases twice)  {tot:10.0f} shader cycles/step  = {tot/2.4e3:.1f} us at 2.4 GHz")
