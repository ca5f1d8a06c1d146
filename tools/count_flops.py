#!/usr/bin/env python3
"""Algorithmic flops per env-step of the BASELINE workloads = the CPU oracle's instrumented operation count (SURVEY.md 8d:
add / mul / div / sqrt = 1, fma = 2), from oracle/libmjo_count.so (oracle/mjo_count.h) on bench.py's own workload (same initial
states, same OU ctrl noise).  Writes profiles/r05_oracle_flops.json, which bench.py turns into `roofline.fp64.useful_*`.
Runs on the CPU (no GPU needed):  python tools/count_flops.py [--envs 16] [--steps 400]"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mujoco_ros_pkgs_amd import binding, mjcf  # noqa: E402


def count(name, nenv, nsteps, L):
    model = mjcf.load_asset(name)
    desc, keep = binding.make_desc(model)
    qpos, qvel = bench.initial_state(name, model, nenv, 1000)
    qpos, qvel = np.ascontiguousarray(qpos), np.ascontiguousarray(qvel)
    pd = C.POINTER(C.c_double)
    L.mjo_flops_reset()
    L.mjo_rollout(C.byref(desc), nenv, nsteps, qpos.ctypes.data_as(pd), qvel.ctypes.data_as(pd), None, None,
                  bench.WORKLOADS[name][1], 0.1, 12345, 0, 1)
    return L.mjo_flops_get() / float(nenv * nsteps), model


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=16)
    ap.add_argument("--steps", type=int, default=400)
    a = ap.parse_args()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libmjo_count.so"])
    L = C.CDLL(os.path.join(ROOT, "oracle", "libmjo_count.so"))
    L.mjo_flops_get.restype = C.c_ulonglong
    L.mjo_rollout.restype = C.c_int
    L.mjo_rollout.argtypes = [C.POINTER(binding.ModelDesc), C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                              C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_double, C.c_uint64, C.c_int64, C.c_int]
    out = {"definition": "oracle operation count: + - * / sqrt and libm calls = 1 each (a multiply-add = 2); comparisons, negation, "
                         "fabs / fmin / fmax, copies and integer work = 0 (oracle/mjo_count.h)",
           "sample": f"{a.envs} envs x {a.steps} steps of bench.py's workload (seed 1000 initial states, OU ctrl noise seed 12345)"}
    for name in ("franka_like", "franka_table", "shadow_hand_grasp", "shadow_hand_like"):
        per, model = count(name, a.envs, a.steps, L)
        out[name] = {"flops_per_env_step": per, "solver": {0: "PGS", 1: "CG", 2: "Newton"}[int(model["solver"])] if model["nefcmax"] else "none",
                     "nv": int(model["nv"]), "nefcmax": int(model["nefcmax"])}
        print(f"{name}: {per:.0f} flops / env-step")
    path = os.path.join(ROOT, "profiles", "r05_oracle_flops.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
