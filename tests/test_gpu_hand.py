"""GPU parity on BASELINE config 5's model: Shadow-Hand-like 24-DoF hand + free cube held in a half-closed grasp
(capsule-box and capsule-capsule contacts, joint limits, elliptic cones, Newton with 2 / 4 constraint rows per
lane and the Hessian on v_mfma_f64_16x16x4_f64).  Tolerances as in test_gpu_contact.py: inputs of the solver
1e-10, solver outputs 1e-6 relative (MFMA accumulates J'WJ in a different order than the oracle's loops)."""
import numpy as np
import pytest

from test_gpu_contact import PRE, ROWS, _close, binding_dim

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["rows<=128", "rows<=256", "rows<=256, J beyond the frame's share in HBM",
                                        "rows<=256, 24 rows in the frame: HBM and frame paths mixed"])
def setup(request, oracle_built):
    import os
    from mujoco_ros_pkgs_amd import engine, mjcf, workloads
    kw = {"nefcmax": 128} if request.param == "rows<=128" else {"nefcmax": 160}
    model = mjcf.load_asset("shadow_hand_like", **kw)
    # Above 128 rows of capacity the fused step's frame holds the first 64 rows of efc_J and an env-step with more rows reads J
    # from the env's block in HBM.  The grasp states have 20 - 50 rows: the third variant lowers the frame's share
    # to 8 rows (MJB_DEBUG_JROWS, read when the model is compiled) so that every env-step takes the HBM path.
    # (round 3: the cap covers EVERY per-row array of the fused frame, and an env-step beyond it keeps all of its row data in the
    #  HBM block; the fourth variant puts the cap at 24 rows, inside the grasp states' range, so that one rollout mixes both paths)
    hbm = "HBM" in request.param
    if hbm:
        os.environ["MJB_DEBUG_JROWS"] = "24" if "mixed" in request.param else "8"
    try:
        cm = engine.CompiledModel(model)
    finally:
        os.environ.pop("MJB_DEBUG_JROWS", None)
    if hbm:
        full, fused = cm.lib.mjb_frame_bytes(cm.ptr, 0), cm.lib.mjb_frame_bytes(cm.ptr, 1)
        assert fused < full - 8 * (160 - 64) * model["nv"], (full, fused)
    # settle on the CPU oracle so that the states carry finger / palm / cube contacts
    qpos, qvel = workloads.hand_grasp_states(model, 24, seed=4)
    qpos, qvel, _ = oracle_built.rollout(model, qpos, qvel, 150, noise_std=0.1, noise_rate=0.1, seed=5, nthreads=8)
    return model, cm, engine, oracle_built, qpos, qvel


def test_hand_constraint_stages_match_oracle(setup):
    model, cm, engine, po, qpos, qvel = setup
    nenv, nv = qpos.shape[0], model["nv"]
    ctrl = np.random.default_rng(2).uniform(-0.2, 0.2, (nenv, model["nu"]))
    b = engine.Batch(cm, nenv)
    b.set("qpos", qpos)
    b.set("qvel", qvel)
    b.set("ctrl", ctrl)
    b.forward()
    rows = [r for r in ROWS if r != "efc_b"]
    got = {f: b.get(f) for f in PRE + rows + ["efc_J", "efc_KBIP", "efc_force", "qacc", "qfrc_constraint", "ncon", "nefc",
                                               "contact_geom", "contact_dim", "efc_type", "efc_id", "sensordata"]}
    d = po.OracleData(model)
    seen_con, pairs = 0, set()
    for e in range(nenv):
        d.reset()
        d.qpos[:] = qpos[e]
        d.qvel[:] = qvel[e]
        d.ctrl[:] = ctrl[e]
        d.forward()
        ncon, nefc = int(d.ncon[0]), int(d.nefc[0])
        assert got["ncon"][e, 0] == ncon and got["nefc"][e, 0] == nefc, f"env {e}: counts {got['ncon'][e, 0]}/{ncon}"
        seen_con += ncon
        for c in range(ncon):
            pairs.add((int(model["geom_type"][d.contact_geom[2 * c]]), int(model["geom_type"][d.contact_geom[2 * c + 1]])))
        assert np.array_equal(got["contact_geom"][e][:2 * ncon], d.contact_geom[:2 * ncon])
        assert np.array_equal(got["efc_type"][e][:nefc], d.efc_type[:nefc])
        assert np.array_equal(got["efc_id"][e][:nefc], d.efc_id[:nefc])
        for f in PRE:
            w = binding_dim(model, f) // model["nconmax"]
            _close(got[f][e][:w * ncon], d.field(f)[:w * ncon], 1e-10, f"{f} env {e}")
        for f in rows:
            _close(got[f][e][:nefc], d.field(f)[:nefc], 1e-10, f"{f} env {e}")
        _close(got["efc_J"][e][:nv * nefc], d.efc_J[:nv * nefc], 1e-10, f"efc_J env {e}")
        _close(got["efc_force"][e][:nefc], d.efc_force[:nefc], 1e-6, f"efc_force env {e}")
        _close(got["qfrc_constraint"][e], d.qfrc_constraint, 1e-6, f"qfrc_constraint env {e}")
        _close(got["qacc"][e], d.qacc, 1e-6, f"qacc env {e}")
        _close(got["sensordata"][e], d.sensordata, 1e-10, f"sensordata env {e}")
    assert seen_con >= 3 * nenv and (3, 6) in pairs, f"scenario did not exercise capsule-box contacts ({seen_con}, {pairs})"
    b.close()


def test_hand_rollout_matches_oracle(setup):
    model, cm, engine, po, qpos, qvel = setup
    b = engine.Batch(cm, qpos.shape[0])
    b.set("qpos", qpos)
    b.set("qvel", qvel)
    b.set_ctrl_noise(0.1, 0.1, 12345, 0)
    b.step(1)
    oq, ov, _ = po.rollout(model, qpos, qvel, 1, noise_std=0.1, noise_rate=0.1, seed=12345)
    _close(b.get("qpos"), oq, 1e-10, "qpos after 1 step")
    _close(b.get("qvel"), ov, 1e-7, "qvel after 1 step")
    b.step(19)
    oq, ov, os_ = po.rollout(model, qpos, qvel, 20, noise_std=0.1, noise_rate=0.1, seed=12345)
    _close(b.get("qpos"), oq, 1e-8, "qpos after 20 steps")
    _close(b.get("qvel"), ov, 1e-6, "qvel after 20 steps")
    b.close()


def test_hand_holds_the_cube(setup):
    """Size-independent property at a larger batch: after 300 noisy steps every env still has the cube in the hand."""
    model, cm, engine, po, qpos, qvel = setup
    from mujoco_ros_pkgs_amd import workloads
    nenv = 512
    q0, v0 = workloads.hand_grasp_states(model, nenv, seed=9)
    b = engine.Batch(cm, nenv)
    b.set("qpos", q0)
    b.set("qvel", v0)
    b.set_ctrl_noise(0.1, 0.1, 3, 0)
    b.step(300)
    q = b.get("qpos")
    assert np.all(np.isfinite(q)) and b.warning_count() == 0
    assert np.all(q[:, 2] > 0.11) and np.all(np.abs(q[:, 0] - 0.07) < 0.05), "cube left the palm"
    b.close()
