"""Differential test over RANDOM models: seeded MJCF trees mixing everything the loader takes -- hinge / slide / ball / free joints with limits,
springs and dampers, capsule / sphere / box geoms over a floor, fixed tendons (limited), joint / tendon actuators of every supported kind
(motor, position, velocity, general with filter / integrator dynamics, intvelocity), a joint equality, explicit contact pairs, sensors -- stepped by
the HIP path and by the oracle from the same states.  What the hand-written worlds of the other tests do not cover is the COMBINATIONS."""
import os

import numpy as np
import pytest

from mujoco_ros_pkgs_amd import mjcf


def random_model(seed):
    rng = np.random.default_rng(seed)
    solver = ["Newton", "PGS", "CG"][seed % 3]
    cone = ["pyramidal", "elliptic"][(seed // 3) % 2]
    nbody = int(rng.integers(3, 10)) if "MJB_RANDOM_BODIES" not in os.environ else int(rng.integers(10, int(os.environ["MJB_RANDOM_BODIES"])))   # (the hunt for the nv 17 .. 64 code paths)
    vel_servo = False
    bodies, joints, scalar = [], [], []
    xml_body = {}
    children = {i: [] for i in range(-1, nbody)}
    for b in range(nbody):
        parent = -1 if b == 0 else int(rng.integers(-1, b))
        children[parent].append(b)
        kind = rng.choice(["hinge", "hinge", "slide", "ball", "free"] if parent == -1 else ["hinge", "hinge", "slide", "ball"])
        name = f"j{b}"
        ax = rng.normal(size=3)
        ax /= np.linalg.norm(ax)
        att = f'name="{name}" damping="{rng.uniform(0.02, 0.3):.3f}" armature="{rng.uniform(0.001, 0.02):.4f}"'
        if kind == "free":
            jx = f'<freejoint name="{name}"/>'
        elif kind == "ball":
            lim = f' limited="true" range="0 {rng.uniform(0.5, 1.2):.3f}"' if rng.random() < 0.5 else ""
            jx = f'<joint type="ball" {att}{lim}/>'
        else:
            rg = (-0.05, 0.05) if kind == "slide" else (-rng.uniform(0.3, 1.0), rng.uniform(0.3, 1.0))
            lim = f' limited="true" range="{rg[0]:.3f} {rg[1]:.3f}" margin="{rng.choice([0, 0.01])}"' if rng.random() < 0.6 else ""
            st = f' stiffness="{rng.uniform(0, 3):.3f}" springref="{rng.uniform(-0.1, 0.1):.3f}"' if rng.random() < 0.3 else ""
            fl = f' frictionloss="{rng.uniform(0.01, 0.1):.3f}"' if rng.random() < 0.15 else ""
            jx = f'<joint type="{kind}" axis="{ax[0]:.4f} {ax[1]:.4f} {ax[2]:.4f}" {att}{lim}{st}{fl}/>'
            scalar.append(name)
        joints.append((name, kind))
        gk = rng.choice(["capsule", "sphere", "box"])
        L = rng.uniform(0.08, 0.2)
        if gk == "capsule":
            gx = f'<geom name="g{b}" type="capsule" fromto="0 0 0 {L:.3f} 0 0" size="{rng.uniform(0.015, 0.03):.3f}" mass="{rng.uniform(0.1, 0.8):.3f}"/>'
        elif gk == "sphere":
            gx = f'<geom name="g{b}" type="sphere" size="{rng.uniform(0.03, 0.05):.3f}" pos="{L / 2:.3f} 0 0" mass="{rng.uniform(0.1, 0.8):.3f}"/>'
        else:
            gx = f'<geom name="g{b}" type="box" size="{L / 2:.3f} {rng.uniform(0.02, 0.04):.3f} {rng.uniform(0.02, 0.04):.3f}" pos="{L / 2:.3f} 0 0" mass="{rng.uniform(0.1, 0.8):.3f}"/>'
        extra = ""
        if rng.random() < 0.4:
            cd = rng.choice([1, 3, 3, 4, 6])
            extra += f' condim="{cd}" friction="{rng.uniform(0.3, 1.2):.3f} {rng.uniform(0.001, 0.02):.4f} {rng.uniform(0.0001, 0.002):.5f}"'
        if rng.random() < 0.25:
            extra += f' margin="{rng.choice([0.002, 0.005]):.3f}" gap="{rng.choice([0, 0.001]):.3f}"'
        if rng.random() < 0.2:
            extra += f' priority="{rng.integers(0, 3)}" solmix="{rng.uniform(0.5, 2):.3f}" solref="{rng.uniform(0.01, 0.03):.4f} {rng.uniform(0.7, 1.2):.3f}"'
        gx = gx.replace("/>", extra + "/>")
        if rng.random() < 0.3:   # a second geom on the body
            g2k = rng.choice(["sphere", "box", "capsule"])
            off = f'{rng.uniform(0, L):.3f} {rng.uniform(-0.03, 0.03):.3f} {rng.uniform(-0.03, 0.03):.3f}'
            if g2k == "sphere":
                gx += f'<geom name="h{b}" type="sphere" size="{rng.uniform(0.015, 0.03):.3f}" pos="{off}" mass="0.05"/>'
            elif g2k == "box":
                gx += f'<geom name="h{b}" type="box" size="{rng.uniform(0.015, 0.04):.3f} {rng.uniform(0.015, 0.03):.3f} {rng.uniform(0.01, 0.03):.3f}" pos="{off}" euler="{rng.uniform(-1, 1):.3f} {rng.uniform(-1, 1):.3f} 0" mass="0.05"/>'
            else:
                gx += f'<geom name="h{b}" type="capsule" size="{rng.uniform(0.01, 0.02):.3f} {rng.uniform(0.02, 0.05):.3f}" pos="{off}" euler="{rng.uniform(-1, 1):.3f} 0 {rng.uniform(-1, 1):.3f}" mass="0.05"/>'
        site = f'<site name="s{b}" pos="{L / 2:.3f} 0 0.01"/>'
        pos = (rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(0.12, 0.5)) if parent == -1 else (L, 0, rng.uniform(-0.02, 0.02))
        xml_body[b] = (f'<body name="b{b}" pos="{pos[0]:.3f} {pos[1]:.3f} {pos[2]:.3f}">{jx}{gx}{site}', "</body>")

    def emit(b):
        o, c = xml_body[b]
        return o + "".join(emit(k) for k in children[b]) + c
    world = "".join(emit(k) for k in children[-1])
    tendons, acts, eqs, pairs, sens = [], [], [], [], []
    if len(scalar) >= 2 and rng.random() < 0.7:
        a, b2 = rng.choice(len(scalar), 2, replace=False)
        lim = ' limited="true" range="-0.4 0.4"' if rng.random() < 0.5 else ""
        if rng.random() < 0.25:
            lim += f' frictionloss="{rng.uniform(0.01, 0.1):.3f}"'
        if rng.random() < 0.25:
            lim += f' stiffness="{rng.uniform(0.5, 3):.3f}"'
        tendons.append(f'<fixed name="t0"{lim}><joint joint="{scalar[a]}" coef="{rng.uniform(0.5, 1.5):.3f}"/><joint joint="{scalar[b2]}" coef="{rng.uniform(-1.5, -0.5):.3f}"/></fixed>')
    for k, jn in enumerate(scalar):
        r = rng.random()
        if r < 0.2:
            acts.append(f'<motor name="a{k}" joint="{jn}" gear="{rng.uniform(0.5, 2):.3f}" ctrllimited="true" ctrlrange="-1 1"/>')
        elif r < 0.35:
            acts.append(f'<position name="a{k}" joint="{jn}" kp="{rng.uniform(2, 10):.3f}"/>')
        elif r < 0.45:
            acts.append(f'<velocity name="a{k}" joint="{jn}" kv="{rng.uniform(0.1, 1):.3f}"/>')
            vel_servo = True
        elif r < 0.6:
            acts.append(f'<general name="a{k}" joint="{jn}" dyntype="{rng.choice(["filter", "integrator"])}" dynprm="{rng.uniform(0.02, 0.2):.3f}" gainprm="{rng.uniform(0.5, 2):.3f}" '
                        f'actlimited="true" actrange="-0.5 0.5" forcelimited="true" forcerange="-2 2"/>')
        elif r < 0.7:
            acts.append(f'<intvelocity name="a{k}" joint="{jn}" kp="{rng.uniform(2, 10):.3f}" actrange="-0.3 0.3"/>')
    if tendons and rng.random() < 0.7:
        acts.append(f'<motor name="at" tendon="t0" gear="{rng.uniform(0.3, 1):.3f}"/>')
    if len(scalar) >= 2 and rng.random() < 0.3:
        eqs.append(f'<joint joint1="{scalar[0]}" joint2="{scalar[-1]}" polycoef="0 {rng.uniform(0.5, 1):.3f} 0 0 0"/>')
    if nbody >= 4 and rng.random() < 0.2:
        a, b2 = rng.choice(np.arange(1, nbody), 2, replace=False)
        eqs.append(f'<connect body1="b{a}" body2="b{b2}" anchor="0.02 0 0"/>' if rng.random() < 0.5 else f'<weld body1="b{a}" body2="b{b2}"/>')
    if rng.random() < 0.6:
        g = int(rng.integers(0, nbody))
        pairs.append(f'<pair geom1="floor" geom2="g{g}" condim="{rng.choice([1, 3, 4])}" friction="{rng.uniform(0.3, 1):.3f} {rng.uniform(0.3, 1):.3f} 0.01 0.001 0.001"/>')
    for k, (jn, kind) in enumerate(joints):
        if kind in ("hinge", "slide") and rng.random() < 0.5:
            sens.append(f'<jointpos joint="{jn}"/><jointvel joint="{jn}"/>')
        if kind == "ball":
            sens.append(f'<ballquat joint="{jn}"/>')
    sens.append('<framepos objtype="site" objname="s0"/><velocimeter site="s0"/><subtreelinvel body="b0"/>')
    # a handful of the other sensor types, on random sites / bodies / joints (relative frames included)
    for _ in range(int(rng.integers(2, 8))):
        sb, sb2 = int(rng.integers(0, nbody)), int(rng.integers(0, nbody))
        kind_s = rng.choice(["touch", "accelerometer", "gyro", "force", "torque", "magnetometer", "rangefinder", "framequat", "framexaxis", "framezaxis", "framelinvel",
                             "frameangvel", "framelinacc", "frameangacc", "subtreecom", "subtreeangmom", "framepos_rel", "framelinvel_rel", "limit", "tendon", "actpos", "jactfrc"])
        if kind_s in ("touch", "accelerometer", "gyro", "force", "torque", "magnetometer", "rangefinder"):
            sens.append(f'<{kind_s} site="s{sb}"/>')
        elif kind_s in ("framequat", "framexaxis", "framezaxis", "framelinvel", "frameangvel", "framelinacc", "frameangacc"):
            ot = rng.choice(["site", "body", "xbody", "geom"])
            on = {"site": f"s{sb}", "body": f"b{sb}", "xbody": f"b{sb}", "geom": f"g{sb}"}[ot]
            sens.append(f'<{kind_s} objtype="{ot}" objname="{on}"/>')
        elif kind_s in ("subtreecom", "subtreeangmom"):
            sens.append(f'<{kind_s} body="b{sb}"/>')
        elif kind_s == "framepos_rel":
            sens.append(f'<framepos objtype="site" objname="s{sb}" reftype="body" refname="b{sb2}"/><framequat objtype="body" objname="b{sb}" reftype="site" refname="s{sb2}"/>')
        elif kind_s == "framelinvel_rel":
            sens.append(f'<framelinvel objtype="site" objname="s{sb}" reftype="site" refname="s{sb2}"/><frameangvel objtype="xbody" objname="b{sb}" reftype="body" refname="b{sb2}"/>')
        elif kind_s == "limit" and scalar:
            jn = scalar[int(rng.integers(0, len(scalar)))]
            sens.append(f'<jointlimitpos joint="{jn}"/><jointlimitvel joint="{jn}"/><jointlimitfrc joint="{jn}"/>')
        elif kind_s == "tendon" and tendons:
            sens.append('<tendonpos tendon="t0"/><tendonvel tendon="t0"/><tendonlimitpos tendon="t0"/><tendonlimitfrc tendon="t0"/>')
        elif kind_s == "actpos" and acts:
            nm = acts[int(rng.integers(0, len(acts)))].split('name="')[1].split('"')[0]
            sens.append(f'<actuatorpos actuator="{nm}"/><actuatorvel actuator="{nm}"/>')
        elif kind_s == "jactfrc" and scalar:
            sens.append(f'<jointactuatorfrc joint="{scalar[int(rng.integers(0, len(scalar)))]}"/>')
    for k, a in enumerate(acts[:2]):
        nm = a.split('name="')[1].split('"')[0]
        sens.append(f'<actuatorfrc actuator="{nm}"/>')
    integ = ["Euler", "RK4", "implicitfast"][(seed // 6) % 3]
    if integ == "implicitfast" and vel_servo and any("tendon=" in a for a in acts):
        integ = "Euler"
    xml = f'''<mujoco model="random{seed}"><compiler angle="radian"/>
<option timestep="0.002" solver="{solver}" cone="{cone}" integrator="{integ}" iterations="{100 if solver == "CG" else 40}" tolerance="{"1e-10" if solver == "CG" else "0"}"/>
<size nconmax="16" njmax="120"/>
<worldbody><geom name="floor" type="plane" size="3 3 0.1"/>{world}</worldbody>
<tendon>{"".join(tendons)}</tendon><actuator>{"".join(acts)}</actuator><equality>{"".join(eqs)}</equality>
<contact>{"".join(pairs)}</contact><sensor>{"".join(sens)}</sensor></mujoco>'''
    return xml


SEEDS = list(range(int(os.environ.get("MJB_RANDOM_MODELS", "64"))))   # (MJB_RANDOM_MODELS=1000: the long hunt, profiles/r06_random_models.txt)


def test_random_models_load_and_step_on_the_oracle(oracle_built):
    """(CPU) every generated model compiles and the oracle takes 20 finite steps from a perturbed state: the generator's own sanity."""
    kinds = set()
    for seed in SEEDS:
        m = mjcf.compile_xml_string(random_model(seed))
        kinds.update(int(t) for t in m["jnt_type"])
        d = oracle_built.OracleData(m)
        d.reset()
        rng = np.random.default_rng(1000 + seed)
        d.qvel[:] = rng.uniform(-0.5, 0.5, m["nv"])
        d.ctrl[:] = rng.uniform(-1, 1, m["nu"])
        d.step(20)
        assert np.isfinite(np.array(d.qpos)).all() and np.isfinite(np.array(d.qvel)).all(), seed
    assert kinds == {0, 1, 2, 3}


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SEEDS)
def test_gpu_random_model_matches_oracle(oracle_built, seed):
    from mujoco_ros_pkgs_amd import engine
    m = mjcf.compile_xml_string(random_model(seed))
    try:
        cm = engine.CompiledModel(m)
    except engine.EngineError as ex:
        if "one env per wavefront" in str(ex) or "exceeds one CU" in str(ex):
            pytest.skip(str(ex))
        raise
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
    d = oracle_built.OracleData(m)
    solver = int(m["solver"])
    for nstep in (1, 15):
        b = engine.Batch(cm, n)
        b.set_lane_env(0)
        b.set("qpos", qpos); b.set("qvel", qvel); b.set("ctrl", ctrl)
        if m["na"]:
            b.set("act", act)
        b.step(nstep)
        got = {k: b.get(k) for k in ("qpos", "qvel", "act", "sensordata")}
        resets = b.warning_count()
        b.close()
        # (CG converges to its tolerance RELATIVE to the problem's scale: welded, interpenetrating bodies carry constraint forces of 1e4 N here)
        tol = (1e-6 if solver == 1 else 1e-10) if nstep == 1 else (1e-4 if solver == 1 else 1e-6)
        bad_rows, worst = 0, []
        for e in range(n):
            d.reset(); d.qpos[:] = qpos[e]; d.qvel[:] = qvel[e]; d.ctrl[:] = ctrl[e]
            if m["na"]:
                d.act[:] = act[e]
            capped = False
            for _ in range(nstep):
                d.step(1)
                capped = capped or (solver == 1 and int(d.solver_iter[0]) >= int(m["iterations"]))
            if capped:      # CG cut at its iteration cap is not converged: what it returns depends on the last bit of every operation
                continue
            for k in got:
                r = np.array(getattr(d, k))
                if r.size == 0:
                    continue
                err = np.abs(got[k][e] - r).max() / (1 + np.abs(r).max())
                lim = tol * (100 if k in ("qvel", "sensordata") else 1)
                if err > lim:
                    bad_rows += 1
                    worst.append((e, k, float(err)))
                    # (the models ask for tolerance 0 -- every solver iteration is run, in both implementations -- because a stop test that sits
                    #  on its threshold ends the two one iteration apart: seed 982 at tolerance 1e-10, 4 such ties in 16 envs, each with
                    #  mjData.solver_iter one apart and ~1e-8 in qvel, every env with equal counts at 1e-14.  profiles/r06_random_models.txt)
                    assert err <= lim * 1e3, (seed, nstep, e, k, err)
        # (CG: nonlinear conjugate gradients cut at a fixed count are not converged and amplify the last bit, so CG models keep a real tolerance,
        #  and with it the stop-test ties of DESIGN.md §2: up to two envs of sixteen may sit one iteration apart)
        assert bad_rows <= (6 if solver == 1 else 0), (seed, nstep, bad_rows, worst)
        assert resets == 0 or not np.isfinite(np.array(d.qpos)).all(), (seed, resets)
    # the split step around the callback point (full frame in the HBM workspace) is the fused step (lean frame in LDS), bit for bit
    outs = []
    for split in (False, True):
        b = engine.Batch(cm, n)
        b.set_lane_env(0)
        b.set("qpos", qpos); b.set("qvel", qvel); b.set("ctrl", ctrl)
        if m["na"]:
            b.set("act", act)
        if split:
            for _ in range(3):
                b.step1()
                b.step2()
        else:
            b.step(3)
        outs.append((b.get("qpos"), b.get("qvel"), b.get("act")))
        b.close()
    for x, y in zip(*outs):
        assert np.array_equal(x, y), (seed, "split != fused", float(np.abs(x - y).max()))
    # the other ways of cutting a step at the callback points -- chained halves (mjb_step21_prefix) and, under RK4, the four evaluations
    # (mjb_step2_rk_prefix) -- are the same step too
    b = engine.Batch(cm, n)
    b.set_lane_env(0)
    b.set("qpos", qpos); b.set("qvel", qvel); b.set("ctrl", ctrl)
    if m["na"]:
        b.set("act", act)
    assert b.lib.mjb_step1_prefix(b.ptr, n) == 0
    for k in range(3):
        assert (b.lib.mjb_step21_prefix if k < 2 else b.lib.mjb_step2_prefix)(b.ptr, n) == 0
    assert np.array_equal(b.get("qpos"), outs[0][0]) and np.array_equal(b.get("act"), outs[0][2]), (seed, "chained halves")
    b.close()
    if int(m["integrator"]) == 1:
        b = engine.Batch(cm, n)
        b.set_lane_env(0)
        b.set("qpos", qpos); b.set("qvel", qvel); b.set("ctrl", ctrl)
        if m["na"]:
            b.set("act", act)
        for k in range(3):
            assert b.lib.mjb_step1_prefix(b.ptr, n) == 0
            for rk in range(4):
                assert b.lib.mjb_step2_rk_prefix(b.ptr, n, rk) == 0
        assert np.array_equal(b.get("qpos"), outs[0][0]) and np.array_equal(b.get("act"), outs[0][2]), (seed, "rk4 evaluations")
        b.close()
    # the device-side DefaultRobotHWSim::writeSim stage (EFFORT / POSITION / VELOCITY / *_PID on random scalar joints) against the oracle's, fused steps
    sj = [j for j in range(m["njnt"]) if int(m["jnt_type"][j]) >= 2]
    if solver != 1 and sj:
        from mujoco_ros_pkgs_amd import binding
        import test_hwsim as H
        pick = [int(j) for j in rng.choice(sj, min(len(sj), 4), replace=False)]
        spec = []
        for j in pick:
            meth = str(rng.choice(list(binding.HW_METHODS)))
            spec.append(dict(joint=j, method=meth, kind=str(rng.choice(["revolute", "continuous", "prismatic"])), p=float(rng.uniform(5, 60)), i=float(rng.uniform(0, 5)),
                             d=float(rng.uniform(0, 2)), i_max=2.0, i_min=-2.0, antiwindup=int(rng.integers(0, 2)), effort_limit=float(rng.uniform(2, 20)),
                             lower=float(m["jnt_range"][j][0]) if m["jnt_limited"][j] else -1.0, upper=float(m["jnt_range"][j][1]) if m["jnt_limited"][j] else 1.0))
        cfg = H._oracle_cfg(spec)
        k = 4
        cp, cv, ce = rng.uniform(-0.3, 0.3, (k, len(spec))), rng.uniform(-0.3, 0.3, (k, len(spec))), rng.uniform(-2, 2, (k, len(spec)))
        b = engine.Batch(cm, k)
        b.set_lane_env(0)
        b.hwsim_configure(spec)
        b.hwsim_set_command("position", cp); b.hwsim_set_command("velocity", cv); b.hwsim_set_command("effort", ce)
        b.set("qpos", qpos[:k])
        b.step(12)
        qh, vh = b.get("qpos"), b.get("qvel")
        b.close()
        for e in range(k):
            dh, _ = H._oracle_rollout(oracle_built, m, cfg, qpos[e], cp[e], cv[e], ce[e], 12)
            assert np.abs(qh[e] - np.array(dh.qpos)).max() <= 1e-7 and np.abs(vh[e] - np.array(dh.qvel)).max() <= 1e-5 * (1 + np.abs(np.array(dh.qvel)).max()), \
                (seed, "hwsim", e, [s_["method"] for s_ in spec], float(np.abs(qh[e] - np.array(dh.qpos)).max()), float(np.abs(vh[e] - np.array(dh.qvel)).max()))
    # the sensors plugin's device-side packing (value / cutoff + per-axis Gaussian noise by set_flag, float32 messages) against the oracle's, random noise models
    if m["nsensor"]:
        ns = m["nsensor"]
        flag, mean, sigma = np.zeros(ns, np.int32), np.zeros((ns, 3)), np.zeros((ns, 3))
        for i in range(ns):
            dim = int(m["sensor_dim"][i])
            if rng.random() < 0.5:
                flag[i] = int(rng.integers(1, 8)) if dim >= 3 else 1
                nb = bin(int(flag[i])).count("1")
                mean[i, :nb], sigma[i, :nb] = rng.uniform(-0.5, 0.5, nb), rng.uniform(0.01, 0.2, nb)
        b = engine.Batch(cm, n)
        b.set_lane_env(0)
        b.set("qpos", qpos); b.set("qvel", qvel); b.set("ctrl", ctrl)
        b.set_ctrl_noise(0.0, 0.1, 0, 500)
        b.step(2)
        for i in range(ns):
            if flag[i]:
                nb = bin(int(flag[i])).count("1")
                b.sensor_set_noise(i, int(flag[i]), mean[i][:nb], sigma[i][:nb])
        b.sensor_pack(seed=7 + seed)
        sd, vv, tt = b.get("sensordata"), b.sensor_messages("value"), b.sensor_messages("truth")
        b.close()
        for e in (0, n - 1):
            ov, ot = oracle_built.sensor_pack(m, sd[e], flag, mean.ravel(), sigma.ravel(), 7 + seed, 500 + e, 2)
            assert np.array_equal(tt[e], ot), (seed, "sensor pack truth", e)
            assert np.abs(vv[e] - ov).max() <= 1e-6 * (1 + np.abs(ov).max()), (seed, "sensor pack value", e, float(np.abs(vv[e] - ov).max()))
    # the reference's ctrl-noise injector on the device (mujoco_env.cpp:469-481; Philox-keyed OU process) against the oracle's, activations starting at rest
    if solver != 1 and m["nu"] > 0:
        b = engine.Batch(cm, n)
        b.set_lane_env(0)
        b.set("qpos", qpos); b.set("qvel", qvel)
        b.set_ctrl_noise(0.7, 0.1, 4242 + seed, 0)
        b.step(6)
        qn, vn = b.get("qpos"), b.get("qvel")
        b.close()
        oq, ov, _ = oracle_built.rollout(m, qpos, qvel, 6, noise_std=0.7, noise_rate=0.1, seed=4242 + seed)
        assert np.abs(qn - oq).max() <= 1e-8 and np.abs(vn - ov).max() <= 1e-6 * (1 + np.abs(ov).max()), (seed, "ctrl noise", float(np.abs(qn - oq).max()), float(np.abs(vn - ov).max()))
    # per-env gravity and geom friction (mjb_set_env_*: the reference's setGravity / setGeomProperties services, per env) against the oracle on a model
    # that carries the env's values in the file; an explicit pair's stated friction stays the pair's
    if solver != 1:
        k = 4
        grav = np.tile(np.asarray(m["gravity"], float), (k, 1)) + rng.uniform(-2, 2, (k, 3))
        fric = np.tile(np.asarray(m["geom_friction"], float), (k, 1, 1)) * rng.uniform(0.3, 1.5, (k, m["ngeom"], 1))
        b = engine.Batch(cm, k)
        b.set_lane_env(0)
        b.set("qpos", qpos[:k]); b.set("qvel", qvel[:k]); b.set("ctrl", ctrl[:k])
        if m["na"]:
            b.set("act", act[:k])
        b.set_env_gravity(grav)
        b.set_env_geom_friction(fric)
        b.step(5)
        q5, v5 = b.get("qpos"), b.get("qvel")
        b.close()
        for e in range(k):
            me = mjcf.Model(dict(m))
            me["gravity"] = grav[e].copy()
            me["geom_friction"] = fric[e].copy()
            de = oracle_built.OracleData(me)
            de.reset(); de.qpos[:] = qpos[e]; de.qvel[:] = qvel[e]; de.ctrl[:] = ctrl[e]
            if m["na"]:
                de.act[:] = act[e]
            de.step(5)
            assert np.abs(q5[e] - np.array(de.qpos)).max() <= 1e-8 and np.abs(v5[e] - np.array(de.qvel)).max() <= 1e-6 * (1 + np.abs(np.array(de.qvel)).max()), \
                (seed, "env overrides", e, float(np.abs(q5[e] - np.array(de.qpos)).max()), float(np.abs(v5[e] - np.array(de.qvel)).max()))


def random_pile(seed):
    """Free bodies dropped into one another over a floor: many contacts of every primitive pair (box - box above all), row counts up to the
    solvers' capacities -- the lean / wide / row-capped frames of the Newton kernels and the two-rows-per-lane PGS."""
    rng = np.random.default_rng(10_000 + seed)
    solver = ["Newton", "PGS", "Newton"][seed % 3]
    cone = ["pyramidal", "elliptic"][(seed // 3) % 2]
    nb = int(rng.integers(3, 9 if solver == "Newton" else 6))
    bodies = []
    for b in range(nb):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        kind = rng.choice(["box", "box", "capsule", "sphere"])
        if kind == "box":
            g = f'<geom type="box" size="{rng.uniform(0.03, 0.07):.3f} {rng.uniform(0.03, 0.07):.3f} {rng.uniform(0.02, 0.05):.3f}" mass="{rng.uniform(0.1, 0.5):.3f}"/>'
        elif kind == "capsule":
            g = f'<geom type="capsule" size="{rng.uniform(0.02, 0.04):.3f} {rng.uniform(0.03, 0.08):.3f}" mass="{rng.uniform(0.1, 0.5):.3f}"/>'
        else:
            g = f'<geom type="sphere" size="{rng.uniform(0.03, 0.06):.3f}" mass="{rng.uniform(0.1, 0.5):.3f}"/>'
        cd = rng.choice([1, 3, 3, 4, 6])
        g = g.replace("/>", f' condim="{cd}"/>')
        bodies.append(f'<body name="p{b}" pos="{rng.uniform(-0.08, 0.08):.3f} {rng.uniform(-0.08, 0.08):.3f} {rng.uniform(0.03, 0.25):.3f}" '
                      f'quat="{q[0]:.4f} {q[1]:.4f} {q[2]:.4f} {q[3]:.4f}"><freejoint/>{g}</body>')
    ncon = int(rng.choice([16, 32, 48]))
    return f'''<mujoco model="pile{seed}"><compiler angle="radian"/>
<option timestep="0.002" solver="{solver}" cone="{cone}" iterations="40" tolerance="0"/>
<size nconmax="{ncon}" njmax="{int(rng.choice([64, 128, 250]))}"/>
<worldbody><geom name="floor" type="plane" size="3 3 0.1"/><geom name="wall" type="box" size="0.3 0.02 0.1" pos="0 0.15 0.1"/>{"".join(bodies)}</worldbody></mujoco>'''


PILES = list(range(int(os.environ.get("MJB_RANDOM_PILES", "32"))))


def test_random_piles_load(oracle_built):
    for seed in PILES[:8]:
        m = mjcf.compile_xml_string(random_pile(seed))
        d = oracle_built.OracleData(m)
        d.reset()
        d.step(3)
        assert np.isfinite(np.array(d.qpos)).all() and int(d.ncon[0]) > 0, seed


@pytest.mark.gpu
@pytest.mark.parametrize("seed", PILES)
def test_gpu_random_pile_matches_oracle(oracle_built, seed):
    from mujoco_ros_pkgs_amd import engine
    m = mjcf.compile_xml_string(random_pile(seed))
    try:
        cm = engine.CompiledModel(m)
    except engine.EngineError as ex:
        if "one env per wavefront" in str(ex) or "exceeds one CU" in str(ex):
            pytest.skip(str(ex))
        raise
    n = 8
    rng = np.random.default_rng(700 + seed)
    qpos = np.tile(np.asarray(m["qpos0"], float), (n, 1))
    nb = m["nq"] // 7
    for b in range(nb):
        qpos[:, 7 * b:7 * b + 3] += rng.uniform(-0.02, 0.02, (n, 3))
        q = qpos[:, 7 * b + 3:7 * b + 7] + rng.normal(size=(n, 4)) * 0.1
        qpos[:, 7 * b + 3:7 * b + 7] = q / np.linalg.norm(q, axis=1, keepdims=True)
    qvel = rng.uniform(-0.3, 0.3, (n, m["nv"]))
    d = oracle_built.OracleData(m)
    # the contact lists themselves, entry by entry, on the full frame
    b = engine.Batch(cm, n)
    b.set("qpos", qpos); b.set("qvel", qvel)
    b.forward()
    ncon, geom, dist, pos, frame, nefc = (b.get(k) for k in ("ncon", "contact_geom", "contact_dist", "contact_pos", "contact_frame", "nefc"))
    rows = 0
    for e in range(n):
        d.reset(); d.qpos[:] = qpos[e]; d.qvel[:] = qvel[e]
        d.forward()
        k = int(d.ncon[0])
        assert int(ncon[e][0]) == k, (seed, e, int(ncon[e][0]), k)
        assert int(nefc[e][0]) == int(d.nefc[0]), (seed, e, int(nefc[e][0]), int(d.nefc[0]))
        np.testing.assert_array_equal(geom[e][:2 * k], np.array(d.contact_geom)[:2 * k])
        np.testing.assert_allclose(dist[e][:k], np.array(d.contact_dist)[:k], rtol=0, atol=1e-12)
        np.testing.assert_allclose(pos[e][:3 * k], np.array(d.contact_pos)[:3 * k], rtol=0, atol=1e-12)
        np.testing.assert_allclose(frame[e][:9 * k], np.array(d.contact_frame)[:9 * k], rtol=0, atol=1e-11)
        rows = max(rows, int(d.nefc[0]))
    b.close()
    for nstep in (1, 10):
        b = engine.Batch(cm, n)
        b.set("qpos", qpos); b.set("qvel", qvel)
        b.step(nstep)
        q, v = b.get("qpos"), b.get("qvel")
        b.close()
        tol = 1e-9 if nstep == 1 else 1e-5    # (deep random overlaps: constraint forces of 1e3 - 1e4 N; ten steps of a tumbling pile amplify the last bit)
        for e in range(n):
            d.reset(); d.qpos[:] = qpos[e]; d.qvel[:] = qvel[e]
            d.step(nstep)
            if d.warning(6) or d.warning(4) or d.warning(5):     # mj_check* reset the env in the oracle: the kernel's reset is tested elsewhere
                continue
            eq = np.abs(q[e] - np.array(d.qpos)).max()
            ev = np.abs(v[e] - np.array(d.qvel)).max() / (1 + np.abs(np.array(d.qvel)).max())
            assert eq <= tol and ev <= 100 * tol, (seed, nstep, e, eq, ev, rows)
    # per-env geom sizes (setGeomProperties per env: mjb_set_env_geom_size; the bounding radii stay the model's, as in the reference)
    size = np.tile(np.asarray(m["geom_size"], float), (n, 1, 1))
    size[:, 2:, :] *= rng.uniform(0.85, 1.0, (n, m["ngeom"] - 2, 1))
    b = engine.Batch(cm, n)
    b.set("qpos", qpos); b.set("qvel", qvel)
    b.set_env_geom_size(size)
    b.step(3)
    q3 = b.get("qpos")
    b.close()
    for e in range(n):
        d.reset(); d.qpos[:] = qpos[e]; d.qvel[:] = qvel[e]
        d.set_geom_size(size[e])
        d.step(3)
        if not (d.warning(6) or d.warning(4) or d.warning(5)):
            assert np.abs(q3[e] - np.array(d.qpos)).max() <= 1e-7, (seed, "geom sizes", e, float(np.abs(q3[e] - np.array(d.qpos)).max()))
    d.set_geom_size(None)


@pytest.mark.gpu
def test_gpu_aligned_piles_resolve_ties_like_the_oracle(oracle_built):
    """Scenes built on a grid (quarter-turn orientations, positions and sizes on 5 mm) put the narrow phase on its knife edges: equal face axes,
    vertices ON side planes, surfaces at distance == margin.  MJB_TIE / MJO_TIE move those edges off the grid, so that the two implementations agree
    there too (tools/aligned_pile_hunt.py: 18 of 400 models differed before, 0 of 6000 after)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import aligned_pile_hunt as A
    from mujoco_ros_pkgs_amd import engine
    total = 0
    for seed in range(60):
        m = mjcf.compile_xml_string(A.pile(seed))
        try:
            cm = engine.CompiledModel(m)
        except engine.EngineError:
            continue
        b = engine.Batch(cm, 1)
        b.forward()
        d = oracle_built.OracleData(m)
        d.reset()
        d.forward()
        k = int(d.ncon[0])
        total += k
        assert int(b.get("ncon")[0, 0]) == k, seed
        np.testing.assert_array_equal(b.get("contact_geom")[0][:2 * k], np.array(d.contact_geom)[:2 * k], err_msg=str(seed))
        np.testing.assert_allclose(b.get("contact_dist")[0][:k], np.array(d.contact_dist)[:k], rtol=0, atol=1e-12, err_msg=str(seed))
        np.testing.assert_allclose(b.get("contact_pos")[0][:3 * k], np.array(d.contact_pos)[:3 * k], rtol=0, atol=1e-12, err_msg=str(seed))
        np.testing.assert_allclose(b.get("contact_frame")[0][:9 * k], np.array(d.contact_frame)[:9 * k], rtol=0, atol=1e-12, err_msg=str(seed))
        b.close()
    assert total > 300
