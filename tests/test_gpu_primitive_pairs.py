"""Every remaining primitive pair of the narrow phase on the GPU against the oracle: random poses around
contact and the degenerate ones where a tie or a sign at the rounding level could make the two disagree -- a capsule lying flat on
/ standing on a plane, a box flat / on an edge / on a corner on a plane, a sphere centred on a box face, edge, vertex or inside the
box, concentric spheres, a sphere on a capsule's axis.  Same contact count, distance to 1e-10, point and frame to 1e-8.
(capsule - box, capsule - capsule: tests/test_gpu_capsule_box.py; box - box: tests/test_gpu_boxbox.py.)"""
import numpy as np
import pytest

from mujoco_ros_pkgs_amd import mjcf

pytestmark = pytest.mark.gpu
MARGIN = 0.02
GEOM = {
    "plane": 'type="plane" size="1 1 0.1"',
    "sphere": 'type="sphere" size="0.05"',
    "capsule": 'type="capsule" size="0.03 0.08"',
    "box": 'type="box" size="0.07 0.05 0.04"',
}
REACH = {"sphere": 0.05, "capsule": 0.11, "box": 0.1}
XML = """
<mujoco><compiler angle="radian"/>
<option timestep="0.001" cone="elliptic" solver="Newton"/>
<worldbody>
  <body name="fix" pos="0 0 0"><geom name="fix" {g1} margin="{mg}"/></body>
  <body name="mov" pos="0 0 0.5"><freejoint/><geom name="mov" {g2} mass="0.2" margin="{mg}"/></body>
</worldbody></mujoco>
"""


def _q(axis, ang):
    a = np.asarray(axis, float)
    a = a / np.linalg.norm(a)
    return np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * a])


def _poses(t1, t2, rng):
    P = []
    r1 = 0.0 if t1 == "plane" else REACH[t1]
    for _ in range(250):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        if t1 == "plane":
            p = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(-0.02, REACH[t2] + MARGIN + 0.02)])
        else:
            u = rng.normal(size=3)
            u /= np.linalg.norm(u)
            p = u * rng.uniform(0.0, r1 + REACH[t2] + MARGIN + 0.02)
        P.append((p, q))
    I = np.array([1.0, 0, 0, 0])
    # (stand-offs: penetrating, touching, inside the margin, and either side of the margin's edge -- not ON it: there the contact's
    #  existence is a comparison of two equal numbers up to rounding)
    for dz in (-0.004, 0.0, 0.003, MARGIN - 1e-6, MARGIN + 1e-6):
        if (t1, t2) == ("plane", "sphere"):
            P.append((np.array([0.1, -0.2, 0.05 + dz]), I))
        if (t1, t2) == ("plane", "capsule"):
            P += [(np.array([0, 0, 0.03 + dz]), _q((0, 1, 0), np.pi / 2)), (np.array([0, 0, 0.11 + dz]), I),
                  (np.array([0, 0, 0.03 + dz]), _q((1, 0, 0), np.pi / 2)), (np.array([0, 0, 0.03 + 0.08 * np.sin(0.3) + dz]), _q((0, 1, 0), np.pi / 2 - 0.3))]
        if (t1, t2) == ("plane", "box"):
            P += [(np.array([0, 0, 0.04 + dz]), I), (np.array([0, 0, 0.07 + dz]), _q((0, 1, 0), np.pi / 2)),
                  (np.array([0, 0, (0.07 + 0.04) / np.sqrt(2) + dz]), _q((0, 1, 0), np.pi / 4)),
                  (np.array([0, 0, 0.04 + dz]), _q((0, 0, 1), 0.7)), (np.array([0, 0, 0.09 + dz]), _q((1, 1, 0), 0.9))]
        if (t1, t2) == ("sphere", "sphere"):
            P += [(np.array([0, 0, 0.1 + dz]), I), (np.array([0.1 + dz, 0, 0]), I)]
        if (t1, t2) == ("sphere", "capsule"):
            P += [(np.array([0.08 + dz, 0, 0]), I), (np.array([0, 0, 0.16 + dz]), I), (np.array([0.08 + dz, 0, 0.08]), I)]
        if (t1, t2) == ("sphere", "box"):
            P += [(np.array([0, 0, 0.09 + dz]), I), (np.array([0.12 + dz, 0, 0]), I), (np.array([0.07 + 0.03 + dz, 0.05 + 0.03 + dz, 0]), I),
                  (np.array([0.07, 0.05, 0.04]) + (0.05 + dz) / np.sqrt(3), I), (np.array([0, 0, 0.09 + dz]), _q((0, 0, 1), 0.6))]
    if (t1, t2) == ("sphere", "sphere"):
        P.append((np.zeros(3), I))                               # concentric
    if (t1, t2) == ("sphere", "capsule"):
        P.append((np.zeros(3), I))                               # the sphere's centre on the capsule's axis
    if (t1, t2) == ("sphere", "box"):
        P += [(np.zeros(3), I), (np.array([0.01, 0.0, 0.0]), I), (np.array([0.06, 0.045, 0.035]), I)]   # centre inside the box
    return P


@pytest.mark.parametrize("t1,t2", [("plane", "sphere"), ("plane", "capsule"), ("plane", "box"), ("sphere", "sphere"),
                                   ("sphere", "capsule"), ("sphere", "box")])
def test_gpu_primitive_pair_matches_oracle(t1, t2, oracle_built):
    from mujoco_ros_pkgs_amd import engine
    m = mjcf.compile_xml_string(XML.format(g1=GEOM[t1], g2=GEOM[t2], mg=MARGIN))
    P = _poses(t1, t2, np.random.default_rng(100 * list(GEOM).index(t1) + list(GEOM).index(t2)))
    nenv = len(P)
    qpos = np.array([np.concatenate([p, q]) for p, q in P])
    b = engine.Batch(engine.CompiledModel(m), nenv)
    b.set("qpos", qpos)
    b.forward()
    ncon, dist, pos, frame = b.get("ncon"), b.get("contact_dist"), b.get("contact_pos"), b.get("contact_frame")
    d = oracle_built.OracleData(m)
    hit = 0
    for e in range(nenv):
        d.reset()
        d.qpos[:] = qpos[e]
        d.forward()
        n = int(d.ncon[0])
        assert ncon[e, 0] == n, f"{t1}-{t2} pose {e}: {ncon[e, 0]} contacts, oracle {n}"
        hit += n > 0
        np.testing.assert_allclose(dist[e][:n], d.contact_dist[:n], rtol=0, atol=1e-10, err_msg=f"pose {e}")
        np.testing.assert_allclose(pos[e][:3 * n], d.contact_pos[:3 * n], rtol=0, atol=1e-8, err_msg=f"pose {e}")
        np.testing.assert_allclose(frame[e][:9 * n], d.contact_frame[:9 * n], rtol=0, atol=1e-8, err_msg=f"pose {e}")
    assert 30 <= hit < nenv, (hit, nenv)
    b.close()
