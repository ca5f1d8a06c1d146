import sys, numpy as np
sys.path.insert(0, '/root/repo')
import bench
from mujoco_ros_pkgs_amd import engine, mjcf
m = mjcf.load_asset("franka_like")
cm = engine.CompiledModel(m); nenv=4096
b = engine.Batch(cm, nenv)
qpos,qvel = bench.initial_state("franka_like", m, nenv, seed=1000)
b.set("qpos", qpos); b.set("qvel", qvel); b.set_ctrl_noise(43.5, 0.1, 12345, 0)
prev_q=None; found=None
for it in range(1200):
    b.step(5)
    w=b.warning_count()
    v=b.get("qvel"); q=b.get("qpos")
    if w>0 and found is None:
        t=b.get("time")[:,0]; e=int(np.argmin(t)); found=(it,e)
        print("first reset at launch",it,"env",e,"time now",t[e])
        print("prev qvel", pv[e].round(1)); print("prev qpos", pq[e].round(2))
        break
    pv=v; pq=q
it,e=found
b2 = engine.Batch(cm, nenv); b2.set("qpos", qpos); b2.set("qvel", qvel); b2.set_ctrl_noise(43.5, 0.1, 12345, 0)
b2.step(5*it-10)
for k in range(16):
    b2.step(1); print("t=%.3f"%b2.get("time")[e,0], "qvel", b2.get("qvel")[e].round(1), "qpos78", b2.get("qpos")[e][7:].round(3), "ctrl", b2.get("ctrl")[e][7:].round(0))
