"""Distribution of the constraint-row / contact counts over the envs of a bench workload (why 64 rows of efc_J in the frame are enough
for config 5, how many config-3 env-steps exceed 32 rows).  usage: nefc_hist.py [model] [envs]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from mujoco_ros_pkgs_amd import mjcf, engine
from bench import WORKLOADS, initial_state
name=sys.argv[1] if len(sys.argv)>1 else "franka_table"
m=mjcf.load_asset(name); cm=engine.CompiledModel(m); n=int(sys.argv[2]) if len(sys.argv)>2 else 4096
b=engine.Batch(cm,n)
qp,qv=initial_state(name,m,n,1000); b.set("qpos",qp); b.set("qvel",qv)
b.set_ctrl_noise(WORKLOADS[name][1],0.1,12345,0)
for it in range(6):
    b.step(200); b.forward()
    ne=b.get("nefc")[:,0]; nc=b.get("ncon")[:,0]
    print(it, "nefc mean %.2f p50 %d p90 %d p99 %d max %d  >32: %.2f%%  >64: %.2f%%   ncon mean %.2f max %d"%(ne.mean(),np.percentile(ne,50),np.percentile(ne,90),np.percentile(ne,99),ne.max(),(ne>32).mean()*100,(ne>64).mean()*100,nc.mean(),nc.max()))
print(np.bincount(ne)[:210])
