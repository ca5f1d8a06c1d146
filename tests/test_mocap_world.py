"""The reference's mocap_world.xml as shipped (mujoco_ros_mocap_plugin/assets/mocap_world.xml, fixture copy under
tests/golden/): two mocap boxes, a free box welded to one of them, box-box and plane-box contacts, elliptic cones,
Newton.  CPU: the oracle loads and runs it; GPU: the HIP path agrees with the oracle, incl. the box-box narrow phase
on randomly posed boxes."""
import os

import numpy as np
import pytest

from mujoco_ros_pkgs_amd import mjcf

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _model():
    return mjcf.compile_xml_file(os.path.join(GOLDEN, "mocap_world.xml"))


def test_mocap_world_loads_and_runs_on_the_oracle(oracle_built):
    m = _model()
    assert (m["nmocap"], m["neq"], m["nq"], m["nbody"]) == (2, 1, 7, 4) and m["solver"] == 2 and m["cone"] == 1
    types = {(int(m["geom_type"][a]), int(m["geom_type"][b])) for a, b in m["collpair_geom"]}
    assert (6, 6) in types and (0, 6) in types          # box-box (mocap box vs the free box) and plane-box
    d = oracle_built.OracleData(m)
    d.step(300)
    assert np.all(np.isfinite(d.qpos)) and d.ncon[0] > 0
    # dragging the welded mocap body drags the box
    d.mocap_pos[3:6] = [0.8, 0.3, 0.6]
    d.step(2000)
    assert np.linalg.norm(np.array(d.qpos[:3]) - [0.8, 0.3, 0.6]) < 0.05


@pytest.mark.gpu
def test_gpu_mocap_world_matches_oracle(oracle_built):
    from mujoco_ros_pkgs_amd import engine
    m = _model()
    cm = engine.CompiledModel(m)
    nenv = 40
    rng = np.random.default_rng(5)
    qpos = np.tile(np.asarray(m["qpos0"], dtype=np.float64), (nenv, 1))
    qpos[:, :3] += rng.uniform(-0.3, 0.3, (nenv, 3))       # the free box pushed into / around the mocap box and the ground
    qpos[:, 3:7] = rng.normal(size=(nenv, 4))
    qpos[:, 3:7] /= np.linalg.norm(qpos[:, 3:7], axis=1, keepdims=True)
    b = engine.Batch(cm, nenv)
    b.set("qpos", qpos)
    b.forward()
    got = {f: b.get(f) for f in ("ncon", "contact_dist", "contact_pos", "contact_frame", "contact_geom", "nefc", "efc_J",
                                 "efc_force", "qacc")}
    d = oracle_built.OracleData(m)
    pairs = set()
    for e in range(nenv):
        d.reset()
        d.qpos[:] = qpos[e]
        d.forward()
        ncon, nefc = int(d.ncon[0]), int(d.nefc[0])
        assert got["ncon"][e, 0] == ncon and got["nefc"][e, 0] == nefc, f"env {e}"
        for c in range(ncon):
            pairs.add((int(m["geom_type"][d.contact_geom[2 * c]]), int(m["geom_type"][d.contact_geom[2 * c + 1]])))
        assert np.array_equal(got["contact_geom"][e][:2 * ncon], d.contact_geom[:2 * ncon])
        np.testing.assert_allclose(got["contact_dist"][e][:ncon], d.contact_dist[:ncon], rtol=0, atol=1e-10)
        np.testing.assert_allclose(got["contact_pos"][e][:3 * ncon], d.contact_pos[:3 * ncon], rtol=0, atol=1e-10)
        np.testing.assert_allclose(got["contact_frame"][e][:9 * ncon], d.contact_frame[:9 * ncon], rtol=0, atol=1e-10)
        np.testing.assert_allclose(got["efc_J"][e][:m["nv"] * nefc], d.efc_J[:m["nv"] * nefc], rtol=0, atol=1e-9)
        np.testing.assert_allclose(got["qacc"][e], d.qacc, rtol=1e-6, atol=1e-6 * (1 + np.abs(d.qacc).max()))
    assert (6, 6) in pairs and (0, 6) in pairs
    # short rollout from the shipped initial state
    c = engine.Batch(cm, 4)
    c.step(200)
    d.reset()
    d.step(200)
    np.testing.assert_allclose(c.get("qpos")[0], d.qpos, rtol=0, atol=1e-6)
    b.close()
    c.close()
