import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from mujoco_ros_pkgs_amd import engine, mjcf
from bench import initial_state
name = "shadow_hand_grasp"
m = mjcf.load_asset(name)
cm = engine.CompiledModel(m)
b = engine.Batch(cm, 1024)
qp, qv = initial_state(name, m, 1024, 1000)
b.set("qpos", qp); b.set("qvel", qv)
b.set_ctrl_noise(0.5, 0.1, 12345, 0)
b.step(1); b.synchronize()   # (module load, first-touch costs)
b.reset(); b.set("qpos", qp); b.set("qvel", qv)
ts = []
for k in range(5):
    t0 = time.perf_counter(); b.step(1000); b.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"launch {k}: {ts[-1] * 1e3:.1f} ms, fused frame {b.fused_frame()}", flush=True)
print("first / steady:", ts[0] / np.median(ts[2:]))
