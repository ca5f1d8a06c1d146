#!/usr/bin/env python3
"""Write MuJoCo golden vectors for the oracle (and, through it, the HIP engine): the OFFLINE parity pin.

The arithmetic of the path is libmujoco 2.3.7 (mujoco_ros/CMakeLists.txt:61; call sites mujoco_ros/src/mujoco_env.cpp:498,552,593),
absent from the build container and from the GPU box, so parity is "unpinned" there.  Run this ONCE on any machine that has the
release tree --

    MUJOCO_DIR=/path/to/mujoco-2.3.7 python tools/make_mujoco_golden.py        # writes tests/golden/mujoco_<version>/<model>.npz

-- and commit the eight small .npz files: tests/test_golden_vectors.py then pins the oracle (CPU suite) and the HIP engine (-m gpu)
against real mj_forward / mj_step everywhere, with no live dependency.  The files hold DATA only (inputs and MuJoCo's outputs).

Per model (the three BASELINE assets + the five reference worlds under tests/golden/):
  const_*            compiled-model constants mj_loadXML derived (oracle/mujoco_ref.c: mjref_model)
  qpos, qvel, ctrl   [8][..] seeded input states (the generator below; seed = 5)
  fwd_<field>_<s>    after mj_forward on state s: qacc qfrc_bias qM qLD qacc_smooth qfrc_passive xpos cvel efc_{J,pos,D,R,aref,vel,force}
                     contact_{dist,pos,frame} sensordata, fwd_sizes_<s> = (ncon, nefc)
  ctrl_seq           [100][nu] the recorded ctrl sequence of the rollouts
  roll_qpos_<K>_<s>, roll_qvel_<K>_<s>   (qpos, qvel) after K = 1 / 10 / 100 mj_steps from state s (s = 0, 1) under ctrl_seq

`--self-check DIR` writes the same files FROM THE ORACLE (not a pin: it only exercises the reader; tests use it in a temp dir)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
WORLDS = [("asset", "franka_like"), ("asset", "franka_table"), ("asset", "shadow_hand_like"), ("golden", "pendulum_world"),
          ("golden", "empty_world"), ("golden", "equality_world"), ("golden", "sensors_world"), ("golden", "mocap_world")]
CONST = ("qpos0", "body_mass", "body_inertia", "body_subtreemass", "dof_invweight0", "body_invweight0", "geom_rbound")
FWD = ("qacc", "qfrc_bias", "qM", "qLD", "qacc_smooth", "qfrc_passive", "xpos", "cvel", "efc_J", "efc_pos", "efc_D", "efc_R",
       "efc_aref", "efc_vel", "efc_force", "contact_dist", "contact_pos", "contact_frame")
NSTATE, ROLL_STATES, ROLLS, SEED = 8, 2, (1, 10, 100), 5


def world_path(kind, name):
    if kind == "asset":
        return os.path.join(ROOT, "mujoco_ros_pkgs_amd", "assets", name + ".xml")
    return os.path.join(GOLDEN, name + ".xml")


def seeded_inputs(model):
    """The input states: qpos0 with every hinge / slide coordinate moved by U(-0.05, 0.05), qvel U(-0.1, 0.1), ctrl U(-1, 1)."""
    rng = np.random.default_rng(SEED)
    nq, nv, nu = model["nq"], model["nv"], model["nu"]
    qpos = np.tile(np.asarray(model["qpos0"], dtype=np.float64), (NSTATE, 1))
    for s in range(NSTATE):
        for j in range(model["njnt"]):
            if model["jnt_type"][j] >= 2:
                qpos[s, model["jnt_qposadr"][j]] += rng.uniform(-0.05, 0.05)
    qvel = rng.uniform(-0.1, 0.1, (NSTATE, nv))
    ctrl = rng.uniform(-1, 1, (NSTATE, nu))
    ctrl_seq = rng.uniform(-1, 1, (max(ROLLS), nu))
    return qpos.reshape(NSTATE, nq), qvel, ctrl, ctrl_seq


class MujocoSource:
    label = "MuJoCo"

    def __init__(self, path, model):
        from oracle import mujoco_ref
        self.sim = mujoco_ref.RefSim(path)
        assert (self.sim.nq, self.sim.nv, self.sim.nu) == (model["nq"], model["nv"], model["nu"])

    def const(self, f):
        return self.sim.model(f)

    def forward(self, q, v, c):
        self.sim.reset()
        self.sim.set_state(q, v, c if len(c) else None)
        self.sim.forward()
        out = {f: self.sim.get(f) for f in FWD}
        out["sensordata"] = self.sim.state()[3]
        s = self.sim.sizes()
        out["sizes"] = np.array([s["ncon"], s["nefc"]])
        return out

    def rollout(self, q, v, ctrl_seq, K):
        self.sim.reset()
        self.sim.set_state(q, v, None)
        self.sim.step(K, ctrl_seq[:K] if ctrl_seq.shape[1] else None)
        qq, vv, _, _ = self.sim.state()
        return qq, vv


class OracleSource:
    """--self-check: the same file layout from the oracle."""
    label = "oracle (SELF-CHECK, not a pin)"

    def __init__(self, path, model):
        from oracle import pyoracle
        self.po, self.model = pyoracle, model

    def const(self, f):
        return np.asarray(self.model[f], dtype=np.float64).reshape(-1)

    def _data(self, q, v, c):
        d = self.po.OracleData(self.model)
        d.qpos[:] = q
        d.qvel[:] = v
        if self.model["nu"] and c is not None:
            d.ctrl[:] = c
        return d

    def forward(self, q, v, c):
        d = self._data(q, v, c)
        d.forward()
        nv, ncon, nefc = self.model["nv"], int(d.ncon[0]), int(d.nefc[0])
        cut = {"efc_J": nefc * nv, "contact_dist": ncon, "contact_pos": 3 * ncon, "contact_frame": 9 * ncon}
        out = {}
        for f in FWD:
            a = np.asarray(getattr(d, f), dtype=np.float64).reshape(-1)
            out[f] = a[:cut.get(f, nefc if f.startswith("efc_") else a.size)].copy()
        out["sensordata"] = np.asarray(d.sensordata, dtype=np.float64).copy()
        out["sizes"] = np.array([ncon, nefc])
        return out

    def rollout(self, q, v, ctrl_seq, K):
        d = self._data(q, v, None)
        for k in range(K):
            if self.model["nu"]:
                d.ctrl[:] = ctrl_seq[k]
            d.step(1)
        return np.array(d.qpos), np.array(d.qvel)


def write_one(src_cls, kind, name, outdir, version):
    from mujoco_ros_pkgs_amd import mjcf
    path = world_path(kind, name)
    model = mjcf.compile_xml_file(path)
    src = src_cls(path, model)
    qpos, qvel, ctrl, ctrl_seq = seeded_inputs(model)
    out = {"version": np.array(version), "source": np.array(src.label), "qpos": qpos, "qvel": qvel, "ctrl": ctrl, "ctrl_seq": ctrl_seq}
    for f in CONST:
        out["const_" + f] = src.const(f)
    for s in range(NSTATE):
        for f, a in src.forward(qpos[s], qvel[s], ctrl[s]).items():
            out[f"fwd_{f}_{s}"] = np.asarray(a)
    for s in range(ROLL_STATES):
        for K in ROLLS:
            q, v = src.rollout(qpos[s], qvel[s], ctrl_seq, K)
            out[f"roll_qpos_{K}_{s}"], out[f"roll_qvel_{K}_{s}"] = q, v
    os.makedirs(outdir, exist_ok=True)
    np.savez_compressed(os.path.join(outdir, name + ".npz"), **out)
    return os.path.getsize(os.path.join(outdir, name + ".npz"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--self-check", default="", metavar="DIR", help="write the files from the ORACLE into DIR (exercises the reader; not a pin)")
    ap.add_argument("--only", default="", help="comma-separated model names")
    a = ap.parse_args()
    only = set(a.only.split(",")) if a.only else None
    if a.self_check:
        src_cls, version, outdir = OracleSource, "selfcheck", a.self_check
    else:
        from oracle import mujoco_ref
        if not mujoco_ref.available():
            print("MuJoCo golden vectors: NOT WRITTEN (library absent) -- set MUJOCO_DIR to a MuJoCo release tree (include/mujoco/mujoco.h, lib/libmujoco.so*)")
            return 2
        src_cls, version = MujocoSource, mujoco_ref.version()
        outdir = os.path.join(GOLDEN, "mujoco_" + version)
        if version != "2.3.7":
            print(f"note: the reference pins MuJoCo 2.3.7 (mujoco_ros/CMakeLists.txt:61); this tree is {version}")
    for kind, name in WORLDS:
        if only and name not in only:
            continue
        n = write_one(src_cls, kind, name, outdir, version)
        print(f"{name}: {n} bytes -> {os.path.join(outdir, name + '.npz')}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
