#!/usr/bin/env python3
"""split_step_check.py -- the split step (smooth half one env per lane + constraint half one env per wavefront) against the fused kernel and the oracle,
and its rate on the bench's config-3 workload.   python tools/split_step_check.py [model] [nenv] [K]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mujoco_ros_pkgs_amd import engine, mjcf


def states(name, model, n, seed=1000):
    if name == "franka_table":
        from bench import initial_state
        return initial_state(name, model, n, seed=seed)
    rng = np.random.default_rng(seed)
    qpos = np.tile(model["qpos0"], (n, 1))
    qvel = rng.uniform(-0.5, 0.5, (n, model["nv"]))
    for j in range(model["njnt"]):
        t, qa = int(model["jnt_type"][j]), int(model["jnt_qposadr"][j])
        if t == 0:
            qpos[:, qa:qa + 3] += rng.uniform(-0.05, 0.05, (n, 3))
            q = qpos[:, qa + 3:qa + 7] + rng.uniform(-0.3, 0.3, (n, 4))
            qpos[:, qa + 3:qa + 7] = q / np.linalg.norm(q, axis=1, keepdims=True) * rng.uniform(0.98, 1.02, (n, 1))  # (not exactly unit: mj_kinematics normalises)
        elif t == 1:
            q = qpos[:, qa:qa + 4] + rng.uniform(-0.4, 0.4, (n, 4))
            qpos[:, qa:qa + 4] = q / np.linalg.norm(q, axis=1, keepdims=True)
        else:
            qpos[:, qa] += rng.uniform(-0.3, 0.3, n) * (0.1 if t == 2 else 1.0)
    return qpos, qvel


def run(name="franka_table", n=256, K=20, noise=None, verbose=True):
    from oracle import pyoracle
    pyoracle.build()
    model = mjcf.Model(dict(mjcf.load_asset(name)))
    model["enableflags"] = int(model["enableflags"]) | 2  # mjENBL_ENERGY: mjData.energy of the last step
    cm = engine.CompiledModel(model)
    if noise is None:
        noise = 2.0 if name == "franka_table" else 0.3
    qpos, qvel = states(name, model, n)
    out = {}
    for mode in (0, 1):
        b = engine.Batch(cm, n)
        b.set_split_step(mode)
        b.set("qpos", qpos); b.set("qvel", qvel)
        b.set_ctrl_noise(noise, 0.1, 12345, 0)
        b.step(K)
        out[mode] = {k: b.get(k) for k in ("qpos", "qvel", "qacc", "qacc_warmstart", "sensordata", "time", "ctrl", "energy")}
        out[mode]["info"] = b.split_step_info()
        out[mode]["warn"] = b.warning_count()
        b.close()
    assert out[1]["info"][1], "the split step did not run: %r" % (out[1]["info"],)
    d = {k: float(np.abs(out[0][k] - out[1][k]).max()) for k in ("qpos", "qvel", "qacc", "qacc_warmstart", "sensordata", "time", "ctrl", "energy")}
    wo = {"qpos": 0.0, "qvel": 0.0, "sensordata": 0.0}
    for e in (0, n // 3, n - 1):
        oq, ov, osd = pyoracle.rollout(model, qpos[e:e + 1], qvel[e:e + 1], K, noise_std=noise, noise_rate=0.1, seed=12345, env_offset=int(e))
        wo["qpos"] = max(wo["qpos"], float(np.abs(out[1]["qpos"][e] - oq[0]).max()))
        wo["qvel"] = max(wo["qvel"], float(np.abs(out[1]["qvel"][e] - ov[0]).max()))
        wo["sensordata"] = max(wo["sensordata"], float(np.abs(out[1]["sensordata"][e] - osd[0]).max()))
    if verbose:
        print(f"{name} {n} envs x {K} steps: split vs fused " + " ".join(f"{k} {v:.2e}" for k, v in d.items()) + f" | split vs oracle " + " ".join(f"{k} {v:.2e}" for k, v in wo.items()) +
              f" | warnings {out[0]['warn']} / {out[1]['warn']} | slices {out[1]['info'][2]}")
    return d, wo, (out[0]["warn"], out[1]["warn"])


def rate(n=4096, K=1000, reps=3, mode=1):
    name = "franka_table"
    model = mjcf.load_asset(name)
    cm = engine.CompiledModel(model)
    qpos, qvel = states(name, model, n)
    b = engine.Batch(cm, n)
    b.set_split_step(mode)
    b.set("qpos", qpos); b.set("qvel", qvel)
    b.set_ctrl_noise(2.0, 0.1, 12345, 0)
    b.step(K); b.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); b.step(K); b.synchronize(); ts.append(time.perf_counter() - t0)
    t = min(ts)
    print(f"mode {mode} slices {os.environ.get('MJB_SPLIT_SLICES', '-')}: {n} envs x {K} steps: {t * 1e3:.1f} ms -> {n * K / t / 1e6:.2f} M env-steps/s (warnings {b.warning_count()})")
    b.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "rate":
        rate(int(sys.argv[2]) if len(sys.argv) > 2 else 4096, int(sys.argv[3]) if len(sys.argv) > 3 else 1000, mode=int(sys.argv[4]) if len(sys.argv) > 4 else 1)
    else:
        name = sys.argv[1] if len(sys.argv) > 1 else "franka_table"
        n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
        for K in ([int(sys.argv[3])] if len(sys.argv) > 3 else [1, 10, 60]):
            run(name, n, K)
