"""Which envs are the long pole of a fused launch?  Steps the workload launch by launch and, after each, one step at a time for a few
steps reading solver_iter / nefc: distribution of the solver iterations over the envs.  usage: iter_hist.py [model] [envs] [launches]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from mujoco_ros_pkgs_amd import mjcf, engine
from bench import WORKLOADS, initial_state
name = sys.argv[1] if len(sys.argv) > 1 else "franka_table"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
nl = int(sys.argv[3]) if len(sys.argv) > 3 else 6
S = int(sys.argv[4]) if len(sys.argv) > 4 else 200
m = mjcf.load_asset(name); cm = engine.CompiledModel(m)
b = engine.Batch(cm, n)
b.set_keep_frame(True)
qp, qv = initial_state(name, m, n, 1000); b.set("qpos", qp); b.set("qvel", qv)
b.set_ctrl_noise(WORKLOADS[name][1], 0.1, 12345, 0)
for it in range(nl):
    ms = b.time_steps(S - 8, 1)
    tot = np.zeros(n)
    for k in range(8):
        b.step(1); b.forward()  # (the step's own count is overwritten by mj_checkAcc's flag: re-run the solver on the new state)
        si = b.get("solver_iter")[:, 0]; ne = b.get("nefc")[:, 0]
        tot += si
    tot /= 8
    print("launch %d (%.1f ms): solver_iter/step mean %.1f p50 %.0f p90 %.0f p99 %.0f max %.0f | envs at >=50 iters: %d, at max: %d | nefc mean %.1f max %d | work share of top 1%%: %.1f%%"
          % (it, ms, tot.mean(), np.percentile(tot, 50), np.percentile(tot, 90), np.percentile(tot, 99), tot.max(), (tot >= 50).sum(),
             (tot >= int(m["iterations"]) - 0.5).sum(), ne.mean(), ne.max(), 100 * np.sort(tot)[-n // 100:].sum() / tot.sum()))
