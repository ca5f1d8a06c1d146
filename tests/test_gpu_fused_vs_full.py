"""The fused (compact) frames of the step kernels against the full frame, on every world of the repo under every solver / cone it takes.

Every parity test that reads a frame field (efc_*, contact_*, qM ...) runs on the FULL frame: fused launches keep the frame in LDS under a
leaner layout (arrays overlaid by lifetime, cone blocks laid out by row, a capped share of efc_J) and store only the state.  Round 5 found a
defect that lived in that difference alone (cones of mixed dimension, tests/test_gpu_mixed_condim.py).  This test closes the gap generically:
batch B steps on the full frame (`keep_frame`), its state is copied into batch A before every step, A takes the same step on its fused
frame; the two run the same arithmetic and must agree to rounding (the lane = env kernel, which replaces the fused step of an eligible
unconstrained model, to 1e-11).  mj_step sites: mujoco_env.cpp:498,552,593."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))

CASES = [("franka_like", {}), ("lane_env_tree", {}), ("pendulum_world", {}), ("empty_world", {}), ("sensors_world", {}), ("mocap_world", {}),
         ("equality_world", {}), ("equality_world", {"solver": "Newton"}), ("equality_world", {"solver": "CG"}),
         ("franka_table", {}), ("franka_table", {"cone": "elliptic"}), ("franka_table", {"solver": "Newton"}),
         ("franka_table", {"solver": "Newton", "cone": "elliptic"}), ("franka_table", {"solver": "CG", "cone": "elliptic"}),
         ("shadow_hand_like", {}), ("shadow_hand_like", {"cone": "pyramidal"}), ("shadow_hand_like", {"solver": "CG"}),
         ("shadow_hand_grasp", {}), ("shadow_hand_grasp", {"cone": "pyramidal"}),
         ("franka_like", {"integrator": "RK4"}), ("franka_table", {"integrator": "RK4"}), ("franka_table", {"integrator": "RK4", "solver": "Newton", "cone": "elliptic"}),
         ("shadow_hand_like", {"integrator": "RK4"}), ("equality_world", {"integrator": "RK4"})]


@pytest.mark.parametrize("name,over", CASES, ids=[n + ("" if not o else "-" + "-".join(o.values())) for n, o in CASES])
def test_fused_frame_steps_like_the_full_frame(name, over, oracle_built):
    import make_mujoco_golden as gold
    from mujoco_ros_pkgs_amd import engine, mjcf
    kind = dict((n, k) for k, n in gold.WORLDS)[name]
    try:
        model = mjcf.compile_xml_file(gold.world_path(kind, name), override=over) if over else mjcf.compile_xml_file(gold.world_path(kind, name))
    except Exception as e:  # (a combination mjb_compile refuses, e.g. PGS with elliptic blocks beyond 64 rows)
        pytest.skip(f"not compiled: {e}")
    q, v, c, _ = gold.seeded_inputs(model)
    n = 64
    rng = np.random.default_rng(11)
    rep = n // q.shape[0]
    qp, qv = np.repeat(q, rep, axis=0), np.repeat(v, rep, axis=0)
    for j in range(model["njnt"]):
        if model["jnt_type"][j] >= 2:  # hinge / slide coordinates: a different perturbation per copy (quaternions stay as they are)
            qp[:, model["jnt_qposadr"][j]] += 1e-3 * rng.standard_normal(n)
    qv += 1e-2 * rng.standard_normal(qv.shape)
    try:
        cm = engine.CompiledModel(model)
    except Exception as e:
        pytest.skip(f"not compiled: {e}")
    A, B = engine.Batch(cm, n), engine.Batch(cm, n)
    B.set_keep_frame(True)
    for b in (A, B):
        b.set("qpos", qp)
        b.set("qvel", qv)
        if model["nu"]:
            b.set_ctrl_noise(0.5, 0.1, 99, 0)
    state = ["qpos", "qvel", "qacc_warmstart", "time"] + (["ctrlnoise"] if model["nu"] else []) + (["act"] if model.get("na", 0) else [])
    worst, rows = 0.0, set()
    for s in range(60):
        for k in state:
            A.set(k, B.get(k))
        A.step(1)
        B.step(1)
        if model.get("nefcmax", 0) > 0:
            rows.update(np.unique(B.get("nefc")[:, 0].astype(int)).tolist())
        if model["nv"]:
            worst = max(worst, float(np.abs(A.get("qvel") - B.get("qvel")).max()), float(np.abs(A.get("qpos") - B.get("qpos")).max()))
        assert np.array_equal(A.get("time"), B.get("time"))
    assert A.warning_count() == B.warning_count()
    assert worst <= 1e-11, (name, over, worst, sorted(rows)[-5:])
    # ... and free-running: one fused launch of 30 steps against 30 full-frame steps from the same state (contact dynamics amplify rounding)
    for b in (A, B):
        b.reset()
        b.set("qpos", qp)
        b.set("qvel", qv)
    A.step(30)
    for _ in range(30):
        B.step(1)
    assert (model["nv"] == 0 or np.abs(A.get("qpos") - B.get("qpos")).max() <= 1e-7) and A.warning_count() == B.warning_count()
    A.close()
    B.close()
