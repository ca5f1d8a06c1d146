#!/usr/bin/env python3
"""Where a split step's time goes (the control-callback path of the host runtime): wall time of each C-ABI call of
step1 -> packed D2H -> packed H2D -> step2 -> packed D2H on the Franka-like arm, callback set of 1 / 64 envs."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mujoco_ros_pkgs_amd import binding, engine, mjcf  # noqa: E402

model = mjcf.load_asset("franka_like")
cm = engine.CompiledModel(model)
b = engine.Batch(cm, 4096)
lib = b.lib
F = binding.Field.ids
pull1 = [F[n] for n in ("qpos", "qvel", "ctrl", "qacc", "qacc_warmstart", "qfrc_applied", "xfrc_applied", "sensordata", "time", "qfrc_passive",
                        "xpos", "xquat", "xmat", "xipos", "ximat", "cvel", "subtree_com", "site_xpos", "site_xmat", "geom_xpos", "geom_xmat",
                        "actuator_force", "qfrc_bias", "qfrc_actuator")]
push = [F[n] for n in ("qpos", "qvel", "ctrl", "qfrc_applied", "qfrc_passive")]
pull2 = [F[n] for n in ("qpos", "qvel", "ctrl", "qacc", "qacc_warmstart", "qfrc_applied", "sensordata", "time")]
arr = lambda l: (C.c_int * len(l))(*l)  # noqa: E731
block = np.zeros(1 << 19)
lib.mjb_host_register(block.ctypes.data_as(C.c_void_p), block.nbytes)
pd = C.POINTER(C.c_double)
for ncb in (1, 64):
    acc = np.zeros(6)
    N = 300
    for it in range(N + 20):
        t = [time.perf_counter()]
        lib.mjb_step1_prefix(b.ptr, ncb); t.append(time.perf_counter())
        lib.mjb_get_packed(b.ptr, len(pull1), arr(pull1), 0, ncb, block.ctypes.data_as(pd)); t.append(time.perf_counter())
        lib.mjb_step_rest(b.ptr, ncb)
        lib.mjb_set_packed(b.ptr, len(push), arr(push), 0, ncb, block.ctypes.data_as(pd)); t.append(time.perf_counter())
        lib.mjb_step2_prefix(b.ptr, ncb); t.append(time.perf_counter())
        lib.mjb_get_packed(b.ptr, len(pull2), arr(pull2), 0, ncb, block.ctypes.data_as(pd)); t.append(time.perf_counter())
        if it >= 20:
            acc[:5] += np.diff(t)
            acc[5] += t[-1] - t[0]
    names = ["step1_prefix (enqueue)", "get_packed 24 fields (+sync: waits for step1)", "step_rest + set_packed 5 fields (enqueue)", "step2_prefix (enqueue)", "get_packed 8 fields (+sync: waits for step2)", "total"]
    print(f"callback envs {ncb}: " + "; ".join(f"{n} {1e6 * a / N:.0f} us" for n, a in zip(names, acc)))
