#!/usr/bin/env python3
"""mixed_condim_hand.py [pattern] -- the power-grasp hand with its six condim="4" attributes rewritten by a pattern (e.g. 346436: cones of dimension 3, 4 and 6 in one env-step,
hcd = 6), on the fused frames against the full frame (state copied before every step) and the oracle (one step, sampled envs).  Exploration of the cone-block layouts beyond
the shipped models' uniform condim 4."""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mujoco_ros_pkgs_amd import engine, mjcf, workloads
from oracle import pyoracle


def run(pat, n=128, steps=60, verbose=True):
    xml = open(os.path.join(mjcf.ASSET_DIR, "shadow_hand_grasp.xml")).read()
    it = iter(pat * 10)
    xml = re.sub(r'condim="4"', lambda mm: f'condim="{next(it)}"', xml)
    m = mjcf.compile_xml_string(xml)
    cm = engine.CompiledModel(m)
    qpos, qvel = workloads.hand_power_grasp_states(m, n, seed=3)
    pyoracle.build()
    out = []
    for wide in (False, True):
        A, B = engine.Batch(cm, n), engine.Batch(cm, n)
        B.set_keep_frame(True)
        if wide:
            A.set("qpos", qpos); A.set("qvel", qvel); A.step(100); A.step(100); A.reset()
        fid = A.fused_frame()
        for b in (A, B):
            b.set("qpos", qpos); b.set("qvel", qvel)
        worst, dims, rows = 0.0, set(), []
        for s in range(steps):
            for k in ("qpos", "qvel", "qacc_warmstart", "time"):
                A.set(k, B.get(k))
            A.step(1); B.step(1)
            worst = max(worst, float(np.abs(A.get("qvel") - B.get("qvel")).max()))
            dims |= set(B.get("contact_dim").reshape(-1).astype(int).tolist())
            rows.append(B.get("nefc")[:, 0].copy())
        rows = np.concatenate(rows)
        st = {k: B.get(k) for k in ("qpos", "qvel", "qacc_warmstart")}
        B.step(1)
        gv = B.get("qvel")
        d = pyoracle.OracleData(m, fast=False)
        wo = 0.0
        for e in range(0, n, 16):
            d.reset(); d.qpos[:] = st["qpos"][e]; d.qvel[:] = st["qvel"][e]; d.qacc_warmstart[:] = st["qacc_warmstart"][e]; d.step()
            wo = max(wo, float(np.abs(gv[e] - d.qvel).max()))
        resets = (A.warning_count(), B.warning_count())
        if verbose:
            print(f"pattern {pat}: fused frame {fid}; dims present {sorted(dims - {0})}; rows mean {rows.mean():.1f} max {rows.max()}; fused vs full worst |dqvel| {worst:.3e}; "
                  f"full frame vs oracle (8 envs, one step) {wo:.3e}; resets fused / full {resets[0]} / {resets[1]}; contactfull {A.warning('contactfull')} / {B.warning('contactfull')} "
                  f"cnstrfull {A.warning('cnstrfull')} / {B.warning('cnstrfull')}", flush=True)
        out.append(dict(frame=fid[0], dims=sorted(dims - {0}), worst=worst, vs_oracle=wo, resets=resets))
        A.close(); B.close()
    return out


if __name__ == "__main__":
    run(sys.argv[1] if len(sys.argv) > 1 else "346436")
