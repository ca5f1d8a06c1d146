"""random_model_diag.py <seed> <env> [attr=value ...]: one env of one model of tests/test_gpu_random_models.py, stepped 15 times on the full frame and on the oracle:
qvel / qacc / efc_force differences, solver iterations, row and contact counts per step (runs on the GPU box)."""
import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import test_gpu_random_models as T
from mujoco_ros_pkgs_amd import mjcf, engine
from oracle import pyoracle as po
seed, env = int(sys.argv[1]), int(sys.argv[2])
xml = T.random_model(seed)
for a in sys.argv[3:]:
    k, v = a.split("=", 1)
    import re
    xml = re.sub(k + r'="[^"]*"', f'{k}="{v}"', xml, count=1)
m = mjcf.compile_xml_string(xml)
print(xml[:260])
print("solver", m["solver"], "cone", m["cone"], "integrator", m["integrator"], "nv", m["nv"], "nu", m["nu"], "na", m["na"], "nefcmax", m["nefcmax"], "nconmax", m["nconmax"], "ntendon", m["ntendon"], "neq", m["neq"], "iters", m["iterations"], "tol", m["tolerance"])
cm = engine.CompiledModel(m)
n = 16
rng = np.random.default_rng(500 + seed)
qpos = np.tile(np.asarray(m["qpos0"], float), (n, 1))
for j in range(m["njnt"]):
    a, t = int(m["jnt_qposadr"][j]), int(m["jnt_type"][j])
    if t >= 2:
        qpos[:, a] += rng.uniform(-0.3, 0.3, n) * (0.1 if t == 2 else 1.0)
    else:
        qa = a + (3 if t == 0 else 0)
        q = rng.normal(size=(n, 4)) * 0.3 + np.array([1, 0, 0, 0])
        qpos[:, qa:qa + 4] = q / np.linalg.norm(q, axis=1, keepdims=True)
        if t == 0:
            qpos[:, a + 2] += rng.uniform(-0.05, 0.1, n)
qvel = rng.uniform(-0.5, 0.5, (n, m["nv"]))
ctrl = rng.uniform(-1.2, 1.2, (n, m["nu"]))
act = rng.uniform(-0.1, 0.1, (n, m["na"]))
f = engine.Batch(cm, n); f.set_lane_env(0)
f.set("qpos", qpos); f.set("qvel", qvel); f.set("ctrl", ctrl)
if m["na"]: f.set("act", act)
d = po.OracleData(m); d.reset(); d.qpos[:] = qpos[env]; d.qvel[:] = qvel[env]; d.ctrl[:] = ctrl[env]
if m["na"]: d.act[:] = act[env]
for k in range(15):
    f.step1(); f.step2()
    d.step(1)
    fv = f.get("qvel")[env]
    nef = int(d.nefc[0])
    gf, of = f.get("efc_force")[env][:nef], np.array(d.efc_force)[:nef]
    print("step", k + 1, "dv %.2e" % np.abs(fv - np.array(d.qvel)).max(), "iters", int(f.get("solver_iter")[env, 0]), int(d.solver_iter[0]), "nefc", int(f.get("nefc")[env, 0]), nef,
          "ncon", int(f.get("ncon")[env, 0]), int(d.ncon[0]), "dforce %.2e" % (np.abs(gf - of).max() if nef else 0), "|f| %.2e" % (np.abs(of).max() if nef else 0),
          "types", sorted(set(int(t) for t in np.array(d.efc_type)[:nef])), "dqacc %.2e" % np.abs(f.get("qacc")[env] - np.array(d.qacc)).max())
