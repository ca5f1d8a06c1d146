#!/usr/bin/env python3
"""Per-stage shader-cycle breakdown of the step kernel (profiling build libmjb_prof*.so, env 0 / lane 0).

The profiling kernel records TWO probe ids per launch in 32 bytes of LDS (mjb_debug_profile_window) -- what the lean frames leave
of their last LDS granule -- so it runs the SAME kernel variant on the SAME frame at the SAME residency as the shipped library; the
tool repeats the launch (same initial state) with the window moved over the ids.  The header's env-steps/s is the profiling
build's own; compare it with the bench line.
Usage: python tools/profile_stages.py [--model M] [--envs E] [--steps K] [--sub|--nwt|--col|--mk]"""
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
ap.add_argument("--envs", type=int, default=0, help="default: the bench workload's batch")
ap.add_argument("--steps", type=int, default=0, help="default: the bench workload's steps per launch")
ap.add_argument("--model", default="franka_like")
ap.add_argument("--solver", default="")
ap.add_argument("--sub", action="store_true", help="libmjb_prof_sub.so: slots 24-29 = collision / make_constraint sub-stages (PGS runs)")
ap.add_argument("--nwt", action="store_true", help="libmjb_prof_nwt.so: slots 20-23 = parts of the Newton iteration's gradient step, 19 = line-search points")
ap.add_argument("--ls", action="store_true", help="libmjb_prof_ls.so: slots 20-22 = parts of the Newton line search")
ap.add_argument("--sm", action="store_true", help="libmjb_prof_sm.so: slots 20-28 = phases of com_pos / crb / com_vel / rne")
ap.add_argument("--col", action="store_true", help="libmjb_prof_col.so: slots 20-22 = collision: cull + register narrow phase / box-box / offsets + stores")
ap.add_argument("--mk", action="store_true", help="libmjb_prof_mk.so: slots 26-29 = parts of make_constraint")
ap.add_argument("--only", default="", help="comma-separated probe ids (default: all)")
a = ap.parse_args()

tag = "_sub" if a.sub else ("_nwt" if a.nwt else ("_ls" if a.ls else ("_sm" if a.sm else ("_col" if a.col else ("_mk" if a.mk else "")))))
binding.LIB_PATH = os.environ.get("MJB_PROF_LIB") or os.path.join(ROOT, "mujoco_ros_pkgs_amd", "csrc", f"libmjb_prof{tag}.so")
if a.sub:
    STAGES[19] = "pgs.setup (B row, b, warmstart)"
    STAGES[21], STAGES[22] = "pgs.sweeps per step [count, not cycles]", "pgs.rows per step [count, not cycles]"
    STAGES[20], STAGES[23] = "pgs.J' f (after the sweeps)", "pgs.M^-1 J' f + qacc stores"
    STAGES[30], STAGES[31] = "pgs.AR build", "pgs.warm residual + sweeps"
    STAGES[24:30] = ["col.cull+narrowphase", "col.offsets+params+stores", "mk.count+cut", "mk.row params (pass 2)", "mk.D + equality J",
                     "mk.contact J"]
if a.sm:
    STAGES[20:32] = ["com_pos.subtree com", "com_pos.cinert+cdof+tendon", "crb.accumulate", "crb.buf = I cdof", "crb.qM entries",
                     "com_vel.cvel", "com_vel.cdof_dot+actuator", "rne.cacc+body force", "rne.qfrc_bias", "-", "-", "-"]
if a.col:
    STAGES[20:24] = ["col.cull + register narrow phase", "col.box-box (one lane at a time)", "col.offsets + stores", "-"]
if a.nwt:
    STAGES[19] = "nwt.ls points /iter [count, not cycles]"
    STAGES[20:24] = ["nwt.g dots+park", "nwt.g cone_update", "nwt.g cost sums", "nwt.g J'f + stop test"]
if a.ls:
    STAGES[19] = "nwt.ls points /iter [count, not cycles]"
    STAGES[20:24] = ["nwt.ls M v | J v + park", "nwt.ls cone constants + Gauss terms", "nwt.ls trial points", "-"]
from mujoco_ros_pkgs_amd import engine, mjcf  # noqa: E402
from bench import WORKLOADS, initial_state  # noqa: E402

model = mjcf.load_asset(a.model)
if a.solver:
    model = mjcf.Model(dict(model))
    model["solver"] = {"PGS": 0, "Newton": 2}[a.solver]
wl = WORKLOADS.get(a.model, ("", 1.0, 4096, 200, 0))
envs, steps = a.envs or wl[2], a.steps or wl[3]
cm = engine.CompiledModel(model)
b = engine.Batch(cm, envs)
b.set_launch(a.lanes, a.epb)
qp, qv = initial_state(a.model, model, envs, 1000)
print("frame bytes (full / fused):", b.lib.mjb_frame_bytes(cm.ptr, 0), "/", b.lib.mjb_frame_bytes(cm.ptr, 1))
TOP = set(range(14)) | {15, 16, 17, 18}
ids = [int(t) for t in a.only.split(",")] if a.only else list(range(32))
out = (C.c_uint64 * 64)()
acc = np.zeros(64, dtype=np.uint64)
mss = []
warns = []
# One unrecorded launch first: the frame policy of kernel variant 4 looks at the PREVIOUS launch's row counters, so a batch's first long
# launch still runs on the 64-row frame (the power grasp: 3.5x slower) and would make the first window's two stages read high.
b.set("qpos", qp)
b.set("qvel", qv)
b.set_ctrl_noise(wl[1], 0.1, 12345, 0)
b.lib.mjb_debug_profile_window(b.ptr, 30)  # (ids 30, 31; nothing is read back)
b.step(steps)
for base in sorted({i & ~1 for i in ids}):
    b.reset()
    b.set("qpos", qp)
    b.set("qvel", qv)
    b.set_ctrl_noise(wl[1], 0.1, 12345, 0)
    b.lib.mjb_debug_profile_window(b.ptr, base)
    b.lib.mjb_debug_profile(b.ptr, out, 1)
    mss.append(b.time_steps(steps, 1))
    b.lib.mjb_debug_profile(b.ptr, out, 1)
    acc += np.frombuffer(out, dtype=np.uint64)
    warns.append([b.warning(w) for w in range(8)])
ms = float(np.median(mss))
tot = 0
print("mjData.warning[] per window (inertia, contactfull, cnstrfull, vgeomfull, badqpos, badqvel, badqacc, badctrl):", [list(np.diff([[0] * 8] + warns, axis=0)[i]) for i in range(len(warns))] if len(warns) <= 4 else ("total " + str(warns[-1])))
print(f"lanes={a.lanes} envs={envs} steps={steps}: {ms:.3f} ms/launch (median of {len(mss)} windows, min {min(mss):.3f} max {max(mss):.3f}) -> {envs*steps/ms/1e3:.2f} M env-steps/s")
for i, n in enumerate(STAGES):
    cnt = max(1, int(acc[32 + i]))
    cyc = int(acc[i]) / cnt
    if i in TOP:
        tot += cyc * cnt / steps
    print(f"  {n:26s} {cyc:10.0f} cycles/call  ({int(acc[32+i])} calls)")
print(f"  sum of the top-level stages (ids 0-13, 15-18; calls x cycles / steps)  {tot:10.0f} shader cycles/step = {tot/2.4e3:.1f} us at 2.4 GHz")
