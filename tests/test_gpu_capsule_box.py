"""GPU capsule - box narrow phase against the oracle's.  The two find the minimiser set of the axis-to-box distance differently --
the oracle bisects its (piecewise-linear, non-decreasing) slope twice, the GPU takes the zero set from the slope's six sorted
breakpoints (csrc/mjb_constraint.h capsule_box) -- so agreement on random AND on degenerate poses (axis parallel to a face: a whole
stretch of minimisers and two contacts; parallel to an edge; through the box centre; touching a vertex) pins both: same contact
count, distances to 1e-10, points and normals to 1e-8."""
import numpy as np
import pytest

from mujoco_ros_pkgs_amd import mjcf

pytestmark = pytest.mark.gpu

SLAB = np.array([0.15, 0.12, 0.05])
RAD, HALF, MARGIN = 0.02, 0.1, 0.03
XML = f"""
<mujoco><compiler angle="radian"/>
<option timestep="0.001" cone="elliptic" solver="Newton"/>
<worldbody>
  <body name="slab" pos="0 0 0"><geom name="slab" type="box" size="{SLAB[0]} {SLAB[1]} {SLAB[2]}" margin="{MARGIN}"/></body>
  <body name="mov" pos="0 0 0.5"><freejoint/><geom name="mov" type="capsule" size="{RAD} {HALF}" mass="0.2" margin="{MARGIN}"/></body>
</worldbody></mujoco>
"""


def _axis_to_quat(a):
    """unit quaternion turning the capsule's z axis onto direction a"""
    a = np.asarray(a, float) / np.linalg.norm(a)
    z = np.array([0.0, 0.0, 1.0])
    c = float(z @ a)
    if c < -1 + 1e-12:
        return np.array([0.0, 1.0, 0.0, 0.0])
    v = np.cross(z, a)
    q = np.array([1 + c, v[0], v[1], v[2]])
    return q / np.linalg.norm(q)


def _poses():
    rng = np.random.default_rng(11)
    P = []
    for _ in range(400):      # random poses around the slab's surface
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        u = rng.normal(size=3)
        u /= np.linalg.norm(u)
        s = 1.0 / np.max(np.abs(u) / SLAB)
        P.append((u * s + u * rng.uniform(-0.02, RAD + MARGIN + 0.02), q))
    top = SLAB[2] + RAD
    for dz in (-0.01, 0.0, 0.004, 0.02):    # lying flat on the top face: along x, along y, diagonal, overhanging an edge
        for ax, off in (((1, 0, 0), (0, 0)), ((0, 1, 0), (0.03, 0)), ((1, 1, 0), (0, 0.01)), ((1, 0, 0), (0.1, 0)), ((0, 1, 0), (0, 0.09))):
            P.append((np.array([off[0], off[1], top + dz]), _axis_to_quat(ax)))
    for dz in (-0.005, 0.003):              # standing on the face, axis through the centre; along an edge; pointing at a vertex
        P.append((np.array([0, 0, SLAB[2] + HALF + RAD + dz]), _axis_to_quat((0, 0, 1))))
        P.append((np.array([SLAB[0] + RAD / np.sqrt(2) + dz, 0, SLAB[2] + RAD / np.sqrt(2) + dz]), _axis_to_quat((0, 1, 0))))
        d = np.array([1.0, 1.0, 1.0]) / np.sqrt(3)
        P.append((SLAB + d * (HALF + RAD + dz), _axis_to_quat(d)))
    P.append((np.zeros(3), _axis_to_quat((1, 0, 0))))     # axis through the box centre (deep inside)
    return P


def test_gpu_capsule_box_matches_oracle(oracle_built):
    from mujoco_ros_pkgs_amd import engine
    m = mjcf.compile_xml_string(XML)
    P = _poses()
    nenv = len(P)
    qpos = np.array([np.concatenate([p, q]) for p, q in P])
    b = engine.Batch(engine.CompiledModel(m), nenv)
    b.set("qpos", qpos)
    b.forward()
    ncon, dist, pos, frame = b.get("ncon"), b.get("contact_dist"), b.get("contact_pos"), b.get("contact_frame")
    d = oracle_built.OracleData(m)
    two, one, none = 0, 0, 0
    for e in range(nenv):
        d.reset()
        d.qpos[:] = qpos[e]
        d.forward()
        n = int(d.ncon[0])
        assert ncon[e, 0] == n, f"pose {e}: {ncon[e, 0]} contacts, oracle {n}"
        two, one, none = two + (n == 2), one + (n == 1), none + (n == 0)
        np.testing.assert_allclose(dist[e][:n], d.contact_dist[:n], rtol=0, atol=1e-10, err_msg=f"pose {e}")
        np.testing.assert_allclose(pos[e][:3 * n], d.contact_pos[:3 * n], rtol=0, atol=1e-8, err_msg=f"pose {e}")
        np.testing.assert_allclose(frame[e][:9 * n], d.contact_frame[:9 * n], rtol=0, atol=1e-8, err_msg=f"pose {e}")
    assert two >= 15 and one >= 50 and none >= 3, (two, one, none)   # every outcome of the routine was exercised
    b.close()


XML_CC = f"""
<mujoco><compiler angle="radian"/>
<option timestep="0.001" cone="elliptic" solver="Newton"/>
<worldbody>
  <body name="fix" pos="0 0 0"><geom name="fix" type="capsule" size="0.025 0.12" margin="{MARGIN}"/></body>
  <body name="mov" pos="0 0 0.5"><freejoint/><geom name="mov" type="capsule" size="{RAD} {HALF}" mass="0.2" margin="{MARGIN}"/></body>
</worldbody></mujoco>
"""


def test_gpu_capsule_capsule_matches_oracle(oracle_built):
    """The same for capsule - capsule (closest points of two segments; the parallel case takes its own branch with up to two
    contacts): random poses, exactly parallel side by side / offset / overlapping partly, end to end on one line, crossing at right
    angles, and nearly parallel (1e-9 .. 1e-5 rad) where the determinant of the 2 x 2 system is at the rounding level."""
    from mujoco_ros_pkgs_amd import engine
    m = mjcf.compile_xml_string(XML_CC)
    rng = np.random.default_rng(5)
    P = []
    for _ in range(300):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        u = rng.normal(size=3)
        u /= np.linalg.norm(u)
        P.append((u * rng.uniform(0.0, 0.025 + RAD + MARGIN + 0.03) + np.array([0, 0, rng.uniform(-0.15, 0.15)]), q))
    gap = 0.025 + RAD
    for dz in (-0.01, 0.0, 0.01, 0.05):
        for zoff in (0.0, 0.05, 0.15, 0.21):                      # parallel: aligned, shifted, partly / barely overlapping
            P.append((np.array([gap + dz, 0, zoff]), np.array([1.0, 0, 0, 0])))
        P.append((np.array([0, 0, 0.12 + HALF + gap + dz]), np.array([1.0, 0, 0, 0])))        # end to end on one line
        P.append((np.array([gap + dz, 0, 0]), _axis_to_quat((0, 1, 0))))                    # crossing at right angles
        for ang in (1e-9, 1e-8, 1e-7, 1e-6, 1e-5):                 # nearly parallel
            P.append((np.array([gap + dz, 0, 0.03]), _axis_to_quat((np.sin(ang), 0, np.cos(ang)))))
    nenv = len(P)
    qpos = np.array([np.concatenate([p, q]) for p, q in P])
    b = engine.Batch(engine.CompiledModel(m), nenv)
    b.set("qpos", qpos)
    b.forward()
    ncon, dist, pos, frame = b.get("ncon"), b.get("contact_dist"), b.get("contact_pos"), b.get("contact_frame")
    d = oracle_built.OracleData(m)
    counts = [0, 0, 0]
    for e in range(nenv):
        d.reset()
        d.qpos[:] = qpos[e]
        d.forward()
        n = int(d.ncon[0])
        assert ncon[e, 0] == n, f"pose {e}: {ncon[e, 0]} contacts, oracle {n}"
        counts[min(n, 2)] += 1
        np.testing.assert_allclose(dist[e][:n], d.contact_dist[:n], rtol=0, atol=1e-10, err_msg=f"pose {e}")
        np.testing.assert_allclose(pos[e][:3 * n], d.contact_pos[:3 * n], rtol=0, atol=1e-7, err_msg=f"pose {e}")
        np.testing.assert_allclose(frame[e][:9 * n], d.contact_frame[:9 * n], rtol=0, atol=1e-7, err_msg=f"pose {e}")
    assert counts[0] >= 3 and counts[1] >= 50 and counts[2] >= 4, counts
    b.close()
