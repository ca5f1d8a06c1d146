#!/usr/bin/env python3
"""lane_env_bench.py -- config 2's fused step at several batch sizes: the 16-lanes-per-env kernel beside the lane = env kernel
("run at 4096 / 16 384 / 65 536 / 262 144 envs beside the G = 16 kernel").  Same workload as bench.py's default
line (franka_like, OU ctrl noise std 43.5, Philox seed 12345), K steps per launch, env-steps/s from HIP-event timing of whole launches
(mjb_time_steps).  usage: python tools/lane_env_bench.py [K] [sizes...]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mujoco_ros_pkgs_amd import engine, mjcf  # noqa: E402
from tests.conftest import random_franka_state  # noqa: E402


def main():
    argv = [a for a in sys.argv[1:] if a != "--two-arm"]
    K = int(argv[0]) if argv else 200
    sizes = [int(a) for a in argv[1:]] or [4096, 16384, 65536, 262144]
    if "--two-arm" in sys.argv:  # a 14-dof model that is NOT compiled in: the kernel is built for it by hiprtc at the first launch
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
        from test_gpu_lane_env import two_arm_xml
        model = mjcf.compile_xml_string(two_arm_xml())
    else:
        model = mjcf.load_asset("franka_like")
    model["enableflags"] = int(model["enableflags"]) | 2
    cm = engine.CompiledModel(model)
    rows = []
    for nenv in sizes:
        if "--two-arm" in sys.argv:
            rng = np.random.default_rng(0)
            qpos, qvel = rng.uniform(-0.8, 0.8, (nenv, model["nq"])), rng.uniform(-0.5, 0.5, (nenv, model["nv"]))
        else:
            qpos, qvel = random_franka_state(model, nenv, 0)
        row = {"envs": nenv, "steps_per_launch": K}
        for mode, tag in ((0, "lanes16"), (1, "lane_env")):
            b = engine.Batch(cm, nenv)
            b.set_lane_env(mode)
            b.set("qpos", qpos)
            b.set("qvel", qvel)
            b.set_ctrl_noise(1.5 if "--two-arm" in sys.argv else 43.5, 0.1, 12345, 0)
            b.step(K)  # warm-up (also allocates the noise buffer)
            b.synchronize()
            ms = b.time_steps(K, 3)
            assert b.lane_env_info()[1] == (mode == 1)
            q = b.get("qpos")
            row[tag] = {"ms_per_launch": ms, "env_steps_per_s": nenv * K / (ms * 1e-3), "finite": bool(np.all(np.isfinite(q))),
                        "noise": b.noise_mode(), "resets": int(b.warning_count())}
            b.close()
        row["ratio"] = row["lane_env"]["env_steps_per_s"] / row["lanes16"]["env_steps_per_s"]
        rows.append(row)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
