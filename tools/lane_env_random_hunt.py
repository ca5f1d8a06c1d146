#!/usr/bin/env python3
"""lane_env_random_hunt.py [count]: random hinge / slide trees without constraints -- the topologies the lane = env kernel is BUILT for at run time (hiprtc, ~3 s per
topology and LDS budget) -- stepped by that kernel and by the generic one from the same states: the topology generator + kernel template against the table-driven kernel.
Runs on the GPU box; one line per model, a summary at the end."""
import sys
import os
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mujoco_ros_pkgs_amd import engine, mjcf  # noqa: E402


def tree(seed):
    rng = np.random.default_rng(seed)
    nbody = int(rng.integers(2, 11))
    children = {i: [] for i in range(-1, nbody)}
    body = {}
    joints = []
    for b in range(nbody):
        parent = -1 if b == 0 else int(rng.integers(-1 if rng.random() < 0.15 else 0, b))
        children[parent].append(b)
        kind = rng.choice(["hinge", "hinge", "slide"])
        ax = rng.normal(size=3)
        ax /= np.linalg.norm(ax)
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        extra = ""
        if rng.random() < 0.3:
            extra += f' stiffness="{rng.uniform(1, 30):.3f}" springref="{rng.uniform(-0.3, 0.3):.3f}"'
        if rng.random() < 0.3:
            extra += f' pos="{rng.uniform(-0.03, 0.03):.3f} {rng.uniform(-0.03, 0.03):.3f} 0"'
        jx = f'<joint name="j{b}" type="{kind}" axis="{ax[0]:.4f} {ax[1]:.4f} {ax[2]:.4f}" damping="{rng.uniform(0.05, 2):.3f}" armature="{rng.uniform(0.001, 0.05):.4f}"{extra}/>'
        iq = rng.normal(size=4)
        iq /= np.linalg.norm(iq)
        inert = (f'<inertial pos="{rng.uniform(-0.1, 0.1):.3f} {rng.uniform(-0.05, 0.05):.3f} {rng.uniform(-0.05, 0.1):.3f}" quat="{iq[0]:.4f} {iq[1]:.4f} {iq[2]:.4f} {iq[3]:.4f}" '
                 f'mass="{rng.uniform(0.1, 2):.3f}" diaginertia="{rng.uniform(0.001, 0.02):.4f} {rng.uniform(0.001, 0.02):.4f} {rng.uniform(0.001, 0.02):.4f}"/>')
        site = f'<site name="s{b}" pos="0.05 0 0.02" quat="{iq[1]:.4f} {iq[0]:.4f} {iq[3]:.4f} {iq[2]:.4f}"/>' if rng.random() < 0.4 else ""
        pos = f'{rng.uniform(-0.3, 0.3):.3f} {rng.uniform(-0.3, 0.3):.3f} {rng.uniform(0, 0.4):.3f}'
        body[b] = (f'<body name="b{b}" pos="{pos}" quat="{q[0]:.4f} {q[1]:.4f} {q[2]:.4f} {q[3]:.4f}">{inert}{jx}{site}', "</body>")
        joints.append((f"j{b}", kind, bool(site)))

    def emit(b):
        o, c = body[b]
        return o + "".join(emit(k) for k in children[b]) + c
    acts, sens = [], []
    for k, (jn, kind, has_site) in enumerate(joints):
        r = rng.random()
        if r < 0.3:
            acts.append(f'<motor name="a{k}" joint="{jn}" gear="{rng.uniform(0.5, 3):.3f}" ctrllimited="true" ctrlrange="-2 2"/>')
        elif r < 0.5:
            acts.append(f'<position name="a{k}" joint="{jn}" kp="{rng.uniform(5, 60):.3f}"/>')
        elif r < 0.6:
            acts.append(f'<velocity name="a{k}" joint="{jn}" kv="{rng.uniform(0.5, 5):.3f}" forcelimited="true" forcerange="-4 4"/>')
        if rng.random() < 0.4:
            sens.append(f'<jointpos joint="{jn}"/><jointvel joint="{jn}"/>')
        if has_site and rng.random() < 0.7:
            sens.append(f'<framepos objtype="site" objname="s{k}"/><framequat objtype="site" objname="s{k}"/>')
    for a in acts[:3]:
        nm = a.split('name="')[1].split('"')[0]
        sens.append(f'<actuatorfrc actuator="{nm}"/>')
    return (f'<mujoco model="le{seed}"><compiler angle="radian"/><option timestep="0.002" integrator="Euler"><flag contact="disable"/></option>'
            f'<worldbody>{"".join(emit(k) for k in children[-1])}</worldbody><actuator>{"".join(acts)}</actuator><sensor>{"".join(sens)}<clock/></sensor></mujoco>')


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    bad = used_n = 0
    for seed in range(count):
        m = mjcf.compile_xml_string(tree(seed))
        cm = engine.CompiledModel(m)
        n = 192
        rng = np.random.default_rng(900 + seed)
        qpos = np.tile(np.asarray(m["qpos0"], float), (n, 1)) + rng.uniform(-0.3, 0.3, (n, m["nq"]))
        qvel = rng.uniform(-1, 1, (n, m["nv"]))
        ctrl = rng.uniform(-2.5, 2.5, (n, m["nu"]))
        out = []
        for mode in (0, 1):
            b = engine.Batch(cm, n)
            b.set_lane_env(mode)
            b.set("qpos", qpos); b.set("qvel", qvel); b.set("ctrl", ctrl)
            b.step(60)
            out.append((b.get("qpos"), b.get("qvel"), b.get("sensordata"), b.lane_env_info()[1], b.lane_env_error() if mode else ""))
            b.close()
        dq, dv = np.abs(out[0][0] - out[1][0]).max(), np.abs(out[0][1] - out[1][1]).max()
        ds = np.abs(out[0][2] - out[1][2]).max() if out[0][2].size else 0.0
        ok = dq <= 1e-10 and dv <= 1e-8 and ds <= 1e-9 and np.isfinite(out[1][0]).all()
        used_n += int(bool(out[1][3]))
        bad += int(not ok)
        print(f"seed {seed}: nbody {m['nbody']} nv {m['nv']} nu {m['nu']} nsens {m['nsensor']} lane_env {bool(out[1][3])} dq {dq:.1e} dv {dv:.1e} dsens {ds:.1e} {'ok' if ok else 'MISMATCH'} {out[1][4][:80]}", flush=True)
    print(f"{count} models, lane = env kernel used on {used_n}, mismatches {bad}")


main()
