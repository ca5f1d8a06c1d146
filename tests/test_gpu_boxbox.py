"""Box - box narrow phase on the GPU against the oracle, including the 8-contact face overlap (mjc_BoxBox returns up to 8 contacts;
SURVEY.md 8a row A5) and edge-edge contacts, with the contacts' constraint rows solved by Newton."""
import numpy as np
import pytest

from mujoco_ros_pkgs_amd import mjcf
from test_oracle_contact import BOXBOX

pytestmark = pytest.mark.gpu


def test_box_box_up_to_eight_contacts(oracle_built):
    from mujoco_ros_pkgs_amd import engine
    m = mjcf.compile_xml_string(BOXBOX.format(pos="0 0 0.109", rot="", bx=0.09, by=0.09))
    assert m["nconmax"] >= 8
    nenv = 24
    rng = np.random.default_rng(6)
    qpos = np.tile(np.asarray(m["qpos0"], dtype=np.float64), (nenv, 1))
    yaw = rng.uniform(-np.pi, np.pi, nenv)
    yaw[0] = np.pi / 4                                   # the octagon
    tilt = rng.uniform(-0.03, 0.03, (nenv, 2))
    tilt[:8] = 0
    qpos[:, 0:2] = rng.uniform(-0.05, 0.05, (nenv, 2))
    qpos[0, 0:2] = 0
    qpos[:, 2] = 0.109 + rng.uniform(-0.002, 0.002, nenv)
    qpos[0, 2] = 0.109
    for e in range(nenv):
        cy, sy = np.cos(yaw[e] / 2), np.sin(yaw[e] / 2)
        qz = np.array([cy, 0, 0, sy])
        a = np.array([np.cos(tilt[e, 0] / 2), np.sin(tilt[e, 0] / 2), 0, 0])
        b = np.array([np.cos(tilt[e, 1] / 2), 0, np.sin(tilt[e, 1] / 2), 0])

        def qmul(p, q):
            return np.array([p[0] * q[0] - p[1] * q[1] - p[2] * q[2] - p[3] * q[3], p[0] * q[1] + p[1] * q[0] + p[2] * q[3] - p[3] * q[2],
                             p[0] * q[2] - p[1] * q[3] + p[2] * q[0] + p[3] * q[1], p[0] * q[3] + p[1] * q[2] - p[2] * q[1] + p[3] * q[0]])
        qpos[e, 3:7] = qmul(qz, qmul(a, b))
    qvel = rng.uniform(-0.05, 0.05, (nenv, m["nv"]))
    bt = engine.Batch(engine.CompiledModel(m), nenv)
    bt.set("qpos", qpos)
    bt.set("qvel", qvel)
    bt.forward()
    ncon, dist, pos, frame, qacc = (bt.get(f) for f in ("ncon", "contact_dist", "contact_pos", "contact_frame", "qacc"))
    d = oracle_built.OracleData(m)
    counts = set()
    for e in range(nenv):
        d.reset(); d.qpos[:] = qpos[e]; d.qvel[:] = qvel[e]; d.forward()
        n = int(d.ncon[0])
        counts.add(n)
        assert ncon[e, 0] == n, f"env {e}: {ncon[e, 0]} vs {n}"
        assert np.allclose(dist[e][:n], d.contact_dist[:n], rtol=0, atol=1e-11)
        assert np.allclose(pos[e][:3 * n], d.contact_pos[:3 * n], rtol=0, atol=1e-10)
        assert np.allclose(frame[e][:9 * n], d.contact_frame[:9 * n], rtol=0, atol=1e-10)
        assert np.allclose(qacc[e], d.qacc, rtol=1e-6, atol=1e-6 * (1 + np.abs(d.qacc).max()))
    assert ncon[0, 0] == 8 and max(counts) == 8 and len(counts) >= 3, counts
    bt.step(30)
    oq, ov, _ = oracle_built.rollout(m, qpos, qvel, 30)
    assert np.allclose(bt.get("qpos"), oq, rtol=0, atol=1e-9) and np.allclose(bt.get("qvel"), ov, rtol=0, atol=1e-7)
    bt.close()
