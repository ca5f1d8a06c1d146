"""MuJoCo golden vectors (tools/make_mujoco_golden.py): the OFFLINE pin of the oracle -- and, `-m gpu`, of the HIP engine -- against
real mj_forward / mj_step of libmujoco (2.3.7 in the reference, mujoco_ros/CMakeLists.txt:61), for the three BASELINE assets and the
five reference worlds.  The files are written once on any machine that has the MuJoCo release tree and committed under
tests/golden/mujoco_<version>/; where they are missing the tests below SKIP with the literal "MuJoCo golden vectors: ABSENT" and the
oracle stays "parity unpinned" (oracle/mjo.h).  The reader itself is exercised in every run on files the tool writes
from the oracle into a temp dir (`--self-check`: not a pin, and says so)."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

from mujoco_ros_pkgs_amd import mjcf

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_mujoco_golden as gold  # noqa: E402

ABSENT = "MuJoCo golden vectors: ABSENT"
NAMES = [n for _, n in gold.WORLDS]


def golden_dir():
    env = os.environ.get("MJB_GOLDEN_DIR", "")
    if env:
        return env if glob.glob(os.path.join(env, "*.npz")) else None
    dirs = sorted(glob.glob(os.path.join(HERE, "golden", "mujoco_*")))
    dirs = [d for d in dirs if glob.glob(os.path.join(d, "*.npz"))]
    return dirs[-1] if dirs else None


# tolerances: those of the live comparison (tests/test_mujoco_parity.py)
TOL = dict(rtol=1e-9, atol=1e-10)
TOL_FORCE = dict(rtol=1e-6, atol=1e-8)
TOL_QACC = dict(rtol=1e-7, atol=1e-8)
ROLL = {1: (dict(rtol=1e-9, atol=1e-10), dict(rtol=1e-8, atol=1e-9)), 10: (dict(rtol=1e-8, atol=1e-9), dict(rtol=1e-7, atol=1e-8)),
        100: (dict(rtol=1e-6, atol=1e-7), dict(rtol=1e-5, atol=1e-6))}


def check_against(path, name, src, const=True, fields=gold.FWD + ("sensordata",)):
    """`src`: something with forward(q, v, c) -> {field: array} and rollout(q, v, ctrl_seq, K) -> (qpos, qvel) -- the oracle
    (make_mujoco_golden.OracleSource) or the HIP engine (EngineSource below)."""
    z = np.load(path, allow_pickle=False)
    kind = dict((n, k) for k, n in gold.WORLDS)[name]
    model = mjcf.compile_xml_file(gold.world_path(kind, name))
    q, v, c, seq = gold.seeded_inputs(model)
    assert np.array_equal(q, z["qpos"]) and np.array_equal(v, z["qvel"]) and np.array_equal(seq, z["ctrl_seq"]), "input states drifted from the file's"
    if const:
        for f in gold.CONST:
            mine = np.asarray(model[f], dtype=np.float64).reshape(-1)
            assert mine.shape == z["const_" + f].shape and np.allclose(mine, z["const_" + f], rtol=1e-9, atol=1e-12), f"{name}: model constant {f}"
    s_obj = src(gold.world_path(kind, name), model)
    for s in range(gold.NSTATE):
        out = s_obj.forward(q[s], v[s], c[s])
        assert tuple(out["sizes"]) == tuple(z[f"fwd_sizes_{s}"]), f"{name} state {s}: (ncon, nefc) {tuple(out['sizes'])} vs {tuple(z[f'fwd_sizes_{s}'])}"
        for f in fields:
            if f not in out:
                continue
            ref = z[f"fwd_{f}_{s}"]
            tol = TOL_FORCE if f == "efc_force" else (TOL_QACC if f == "qacc" else TOL)
            assert out[f].shape == ref.shape and np.allclose(out[f], ref, **tol), f"{name} state {s}: {f} (max |d| {np.abs(out[f] - ref).max() if ref.size else 0:.2e})"
    for s in range(gold.ROLL_STATES):
        for K in gold.ROLLS:
            qq, vv = s_obj.rollout(q[s], v[s], seq, K)
            assert np.allclose(qq, z[f"roll_qpos_{K}_{s}"], **ROLL[K][0]), f"{name} state {s}: qpos after {K} steps"
            assert np.allclose(vv, z[f"roll_qvel_{K}_{s}"], **ROLL[K][1]), f"{name} state {s}: qvel after {K} steps"
    return str(z["source"])


class EngineSource:
    """The HIP engine behind the same two calls (one env; `mjb_forward` + field reads, `mjb_step` with ctrl set per step)."""

    def __init__(self, path, model):
        from mujoco_ros_pkgs_amd import engine
        self.model = model
        self.cm = engine.CompiledModel(model)
        self.b = engine.Batch(self.cm, 1)

    def _set(self, q, v, c):
        self.b.reset()
        self.b.set("qpos", q[None])
        self.b.set("qvel", v[None])
        if self.model["nu"] and c is not None:
            self.b.set("ctrl", np.asarray(c)[None])

    def forward(self, q, v, c):
        self._set(q, v, c)
        self.b.forward()
        nv = self.model["nv"]
        ncon, nefc = int(self.b.get("ncon")[0, 0]), int(self.b.get("nefc")[0, 0])
        cut = {"efc_J": nefc * nv, "contact_dist": ncon, "contact_pos": 3 * ncon, "contact_frame": 9 * ncon}
        out = {"sizes": np.array([ncon, nefc]), "sensordata": self.b.get("sensordata")[0]}
        for f in gold.FWD:
            try:
                a = self.b.get(f)[0].reshape(-1)
            except Exception:
                continue   # (a field the engine's full frame does not carry for this model)
            out[f] = a[:cut.get(f, nefc if f.startswith("efc_") else a.size)].copy()
        return out

    def rollout(self, q, v, ctrl_seq, K):
        self._set(q, v, None)
        for k in range(K):
            if self.model["nu"]:
                self.b.set("ctrl", ctrl_seq[k][None])
            self.b.step(1)
        return self.b.get("qpos")[0], self.b.get("qvel")[0]


@pytest.fixture(scope="session")
def selfcheck_dir(tmp_path_factory, oracle_built):
    d = str(tmp_path_factory.mktemp("golden_selfcheck"))
    # every world of the tool, so that the first real MuJoCo file of ANY of them meets a reader that has been through it
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_mujoco_golden.py"), "--self-check", d])
    return d


def test_generator_reports_absence_or_writes(tmp_path):
    from oracle import mujoco_ref
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_mujoco_golden.py"), "--only", "none"], capture_output=True, text=True)
    if mujoco_ref.available():
        assert r.returncode == 0
    else:
        assert r.returncode == 2 and "NOT WRITTEN (library absent)" in r.stdout
    print(ABSENT if golden_dir() is None else "MuJoCo golden vectors: " + golden_dir())


@pytest.mark.parametrize("name", NAMES)
def test_reader_on_self_check_files(selfcheck_dir, oracle_built, name):
    """Plumbing only: the oracle against files the tool wrote FROM the oracle (labelled so inside the file)."""
    label = check_against(os.path.join(selfcheck_dir, name + ".npz"), name, gold.OracleSource, const=False)
    assert "SELF-CHECK" in label


@pytest.mark.parametrize("name", NAMES)
def test_contact_models_have_contact_states(selfcheck_dir, name):
    """The writer's guarantee, read back from the files: a model with contacts is in contact in >= 3 of its 8 forward states."""
    z = np.load(os.path.join(selfcheck_dir, name + ".npz"), allow_pickle=False)
    kind = dict((n, k) for k, n in gold.WORLDS)[name]
    model = mjcf.compile_xml_file(gold.world_path(kind, name))
    sizes = [tuple(int(x) for x in z[f"fwd_sizes_{s}"]) for s in range(gold.NSTATE)]
    assert gold.contact_states(model, sizes), sizes
    if model["nconmax"] > 0 and any(model["jnt_type"][j] == 0 for j in range(model["njnt"])):
        assert all(z[f"fwd_contact_dist_{s}"].size == sizes[s][0] for s in range(gold.NSTATE))
        assert any(z[f"fwd_efc_force_{s}"].size and np.abs(z[f"fwd_efc_force_{s}"]).max() > 0 for s in range(gold.NCONTACT)), "no contact force in any contact state"


@pytest.mark.skipif(golden_dir() is None, reason=ABSENT)
@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_mujoco_golden_vectors(oracle_built, name):
    label = check_against(os.path.join(golden_dir(), name + ".npz"), name, gold.OracleSource)
    assert label == "MuJoCo"


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_engine_reader_on_self_check_files(selfcheck_dir, oracle_built, name):
    """The HIP engine through the same reader, against the oracle-written files (GPU vs oracle on these worlds, whatever the
    machine): keeps the `-m gpu` consumer of the golden files from rotting while the files themselves are absent."""
    check_against(os.path.join(selfcheck_dir, name + ".npz"), name, EngineSource, const=False)


@pytest.mark.gpu
@pytest.mark.skipif(golden_dir() is None, reason=ABSENT)
@pytest.mark.parametrize("name", NAMES)
def test_engine_matches_mujoco_golden_vectors(oracle_built, name):
    check_against(os.path.join(golden_dir(), name + ".npz"), name, EngineSource, const=False)
