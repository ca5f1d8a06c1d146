#!/usr/bin/env python3
"""aligned_pile_hunt.py [count]: piles of free boxes / capsules / spheres whose poses are EXACT -- quarter-turn orientations, positions and sizes on a 5 mm grid --
so that faces, edges and capsule axes are parallel to the last bit: the tie cases of the narrow phase (SAT axis choice, clipping on coincident planes, flat minimiser
sets).  Contact lists of the HIP path against the oracle's on the full frame; runs on the GPU box."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mujoco_ros_pkgs_amd import engine, mjcf  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

S2 = 0.7071067811865476
QUATS = ["1 0 0 0", f"{S2} {S2} 0 0", f"{S2} 0 {S2} 0", f"{S2} 0 0 {S2}", "0 1 0 0", "0 0 1 0", "0.5 0.5 0.5 0.5", f"{S2} -{S2} 0 0"]


def pile(seed):
    rng = np.random.default_rng(50_000 + seed)
    nb = int(rng.integers(2, 7))
    bodies = []
    for b in range(nb):
        kind = rng.choice(["box", "box", "capsule", "sphere"])
        g5 = lambda lo, hi: 0.005 * int(rng.integers(lo, hi))
        if kind == "box":
            g = f'<geom type="box" size="{g5(4, 14):.3f} {g5(4, 14):.3f} {g5(4, 10):.3f}" mass="0.3"/>'
        elif kind == "capsule":
            g = f'<geom type="capsule" size="{g5(4, 8):.3f} {g5(6, 16):.3f}" mass="0.3"/>'
        else:
            g = f'<geom type="sphere" size="{g5(6, 12):.3f}" mass="0.3"/>'
        bodies.append(f'<body name="p{b}" pos="{g5(-12, 13):.3f} {g5(-12, 13):.3f} {g5(4, 40):.3f}" quat="{QUATS[int(rng.integers(0, len(QUATS)))]}"><freejoint/>{g}</body>')
    return (f'<mujoco model="aligned{seed}"><compiler angle="radian"/><option timestep="0.002" solver="Newton" cone="elliptic" iterations="40" tolerance="0"/>'
            f'<size nconmax="48" njmax="250"/><worldbody><geom name="floor" type="plane" size="3 3 0.1"/><geom name="slab" type="box" size="0.2 0.2 0.02" pos="0 0 0.02"/>'
            f'{"".join(bodies)}</worldbody></mujoco>')


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    bad = ran = ncon_tot = 0
    for seed in range(count):
        m = mjcf.compile_xml_string(pile(seed))
        try:
            cm = engine.CompiledModel(m)
        except engine.EngineError:
            continue
        ran += 1
        b = engine.Batch(cm, 1)
        b.forward()
        d = po.OracleData(m)
        d.reset()
        d.forward()
        k, kg = int(d.ncon[0]), int(b.get("ncon")[0, 0])
        ncon_tot += k
        msg = ""
        if k != kg:
            msg = f"ncon gpu {kg} oracle {k}"
        else:
            og, gg = np.array(d.contact_geom)[:2 * k], b.get("contact_geom")[0][:2 * k]
            if not np.array_equal(og, gg):
                msg = "contact geoms differ"
            else:
                e1 = np.abs(b.get("contact_dist")[0][:k] - np.array(d.contact_dist)[:k]).max() if k else 0
                e2 = np.abs(b.get("contact_pos")[0][:3 * k] - np.array(d.contact_pos)[:3 * k]).max() if k else 0
                e3 = np.abs(b.get("contact_frame")[0][:9 * k] - np.array(d.contact_frame)[:9 * k]).max() if k else 0
                if max(e1, e2, e3) > 1e-12:
                    msg = f"dist {e1:.1e} pos {e2:.1e} frame {e3:.1e}"
        if msg:
            bad += 1
            T = {0: "plane", 2: "sphere", 3: "capsule", 6: "box"}
            ty = [T[int(t)] for t in m["geom_type"]]
            gl = [(int(a), int(c)) for a, c in np.array(d.contact_geom)[:2 * k].reshape(-1, 2)]
            gg = [(int(a), int(c)) for a, c in b.get("contact_geom")[0][:2 * kg].reshape(-1, 2)]
            import collections
            co, cg = collections.Counter(gl), collections.Counter(gg)
            diff = [f"{ty[a]}-{ty[c]} oracle {co[(a, c)]} gpu {cg[(a, c)]}" for (a, c) in sorted(set(co) | set(cg)) if co[(a, c)] != cg[(a, c)]]
            if not diff:
                fo, fg = np.array(d.contact_frame)[:9 * k].reshape(-1, 9), b.get("contact_frame")[0][:9 * k].reshape(-1, 9)
                po_, pg = np.array(d.contact_pos)[:3 * k].reshape(-1, 3), b.get("contact_pos")[0][:3 * k].reshape(-1, 3)
                for c in range(k):
                    if np.abs(fo[c] - fg[c]).max() > 1e-12 or np.abs(po_[c] - pg[c]).max() > 1e-12:
                        diff.append(f"{ty[gl[c][0]]}-{ty[gl[c][1]]} contact {c}: normal oracle {fo[c][:3].round(3).tolist()} gpu {fg[c][:3].round(3).tolist()} dpos {np.abs(po_[c] - pg[c]).max():.1e}")
            print(f"seed {seed}: {msg}  {diff[:4]}", flush=True)
        b.close()
    print(f"{ran} aligned piles, {ncon_tot} contacts, mismatching models: {bad}")


if __name__ == "__main__":
    main()
