#!/usr/bin/env python3
"""Throughput of the config-3 PGS kernel against resident waves per CU.  usage: occupancy_probe.py nconmax njmax [envs].  Capacities are shrunk until 8 envs fit one
CU's LDS, then MJB_DEBUG_LDS_BYTES (read once per process: run one process per setting) pads the request back up."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mujoco_ros_pkgs_amd import mjcf, engine
from bench import WORKLOADS, initial_state
name = "franka_table"
m = mjcf.Model(dict(mjcf.load_asset(name)))
m["nconmax"], m["nefcmax"] = int(sys.argv[1]), int(sys.argv[2])
nenv = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
cm = engine.CompiledModel(m)
b = engine.Batch(cm, nenv)
qp, qv = initial_state(name, m, nenv, 1000)
b.set("qpos", qp); b.set("qvel", qv)
b.set_ctrl_noise(WORKLOADS[name][1], 0.1, 12345, 0)
b.step(200); b.synchronize()
ms = b.time_steps(200, 3)
fb = cm.lib.mjb_frame_bytes(cm.ptr, 1)
b.forward()
print(f"nconmax {m['nconmax']} njmax {m['nefcmax']} frame {fb} B  lds floor {os.environ.get('MJB_DEBUG_LDS_BYTES','-')}  envs {nenv}: {ms:.2f} ms / 200 steps -> {nenv*200/ms/1e3:.2f} M env-steps/s  ncon mean {b.get('ncon').mean():.2f} nefc mean {b.get('nefc').mean():.2f}")
