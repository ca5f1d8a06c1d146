"""Real MuJoCo, when the machine has it -- TEST / BENCH INFRASTRUCTURE (see oracle/mjo.h).

The reference's arithmetic is libmujoco 2.3.7 (`mj_step`, mujoco_ros/src/mujoco_env.cpp:498,552,593), absent from the build
container and from the GPU box.  If ``$MUJOCO_DIR`` points at a MuJoCo release tree
(``include/mujoco/mujoco.h`` + ``lib/libmujoco.so*``), this module compiles oracle/mujoco_ref.c against it into
oracle/_ref/libmjref.so and exposes real `mj_step` for (a) tests/test_mujoco_parity.py -- oracle vs MuJoCo on the shipped
worlds and the BASELINE models -- and (b) bench.py's ``cpu_baseline.mujoco`` leg.  Otherwise ``available()`` is False and
every consumer prints the literal "NOT MEASURED (library absent)": nothing is faked."""
from __future__ import annotations

import ctypes as C
import glob
import os
import subprocess
import time

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_REF = os.path.join(_DIR, "_ref")
ABSENT = "NOT MEASURED (library absent)"
_lib = None


def _find():
    root = os.environ.get("MUJOCO_DIR", "")
    if not root:
        return None
    hdr = os.path.join(root, "include", "mujoco", "mujoco.h")
    libs = sorted(glob.glob(os.path.join(root, "lib", "libmujoco.so*")))
    if not os.path.exists(hdr) or not libs:
        return None
    return root, libs[0]


def available():
    return _find() is not None


def load():
    """Build (once) and load the shim; raises RuntimeError(ABSENT) when MuJoCo is not on the machine."""
    global _lib
    if _lib is not None:
        return _lib
    found = _find()
    if not found:
        raise RuntimeError(ABSENT)
    root, libpath = found
    os.makedirs(_REF, exist_ok=True)
    so = os.path.join(_REF, "libmjref.so")
    src = os.path.join(_DIR, "mujoco_ref.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-I", os.path.join(root, "include"), "-o", so, src, libpath,
                               "-Wl,-rpath," + os.path.dirname(libpath)])
    L = C.CDLL(so)
    vp, ci, pd = C.c_void_p, C.c_int, C.POINTER(C.c_double)
    L.mjref_version.restype = C.c_char_p
    L.mjref_load.restype = vp
    L.mjref_load.argtypes = [C.c_char_p, C.c_char_p, ci]
    L.mjref_free.argtypes = [vp]
    L.mjref_sizes.argtypes = [vp, C.POINTER(ci)]
    L.mjref_reset.argtypes = [vp]
    L.mjref_set_state.argtypes = [vp, pd, pd, pd]
    L.mjref_get_state.argtypes = [vp, pd, pd, pd, pd]
    L.mjref_forward.argtypes = [vp]
    L.mjref_step.argtypes = [vp, ci, pd]
    L.mjref_get.restype = ci
    L.mjref_get.argtypes = [vp, C.c_char_p, pd, ci]
    L.mjref_model.restype = ci
    L.mjref_model.argtypes = [vp, C.c_char_p, pd, ci]
    _lib = L
    return L


def version():
    return load().mjref_version().decode()


class RefSim:
    """One real mjModel + mjData loaded from an MJCF file."""

    def __init__(self, xml_path):
        self.L = load()
        err = C.create_string_buffer(1000)
        self.ptr = self.L.mjref_load(xml_path.encode(), err, 1000)
        if not self.ptr:
            raise RuntimeError("mj_loadXML failed: " + err.value.decode())
        s = (C.c_int * 8)()
        self.L.mjref_sizes(self.ptr, s)
        self.nq, self.nv, self.nu, self.nsensordata = s[0], s[1], s[2], s[3]

    def __del__(self):
        try:
            if self.ptr:
                self.L.mjref_free(self.ptr)
                self.ptr = None
        except Exception:
            pass

    @staticmethod
    def _p(a):
        return None if a is None else np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(C.POINTER(C.c_double))

    def reset(self):
        self.L.mjref_reset(self.ptr)

    def set_state(self, qpos=None, qvel=None, ctrl=None):
        keep = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in (qpos, qvel, ctrl)]
        self.L.mjref_set_state(self.ptr, *[self._p(a) for a in keep])

    def state(self):
        q, v, a, s = np.zeros(self.nq), np.zeros(self.nv), np.zeros(self.nv), np.zeros(max(1, self.nsensordata))
        self.L.mjref_get_state(self.ptr, *[x.ctypes.data_as(C.POINTER(C.c_double)) for x in (q, v, a, s)])
        return q, v, a, s[:self.nsensordata]

    def forward(self):
        self.L.mjref_forward(self.ptr)

    def step(self, n=1, ctrl_seq=None):
        cs = None if ctrl_seq is None else np.ascontiguousarray(ctrl_seq, dtype=np.float64)
        assert cs is None or cs.shape == (n, self.nu)
        self.L.mjref_step(self.ptr, int(n), self._p(cs))

    def _get(self, fn, name, cap=1 << 16):
        out = np.zeros(cap)
        n = fn(self.ptr, name.encode(), out.ctypes.data_as(C.POINTER(C.c_double)), cap)
        if n > cap:
            return self._get(fn, name, n)
        return out[:n].copy()

    def get(self, name):
        return self._get(self.L.mjref_get, name)

    def model(self, name):
        return self._get(self.L.mjref_model, name)

    def sizes(self):
        s = (C.c_int * 8)()
        self.L.mjref_sizes(self.ptr, s)
        return dict(zip(("nq", "nv", "nu", "nsensordata", "nbody", "ngeom", "ncon", "nefc"), list(s)))


def asset_path(name):
    return os.path.join(os.path.dirname(_DIR), "mujoco_ros_pkgs_amd", "assets", name + ".xml")


def time_reference(name, noise_std, target_s=6.0):
    """bench.py's ``cpu_baseline.mujoco``: real mj_step, one env, one thread pinned to one core, OU ctrl noise drawn with
    the engine's Philox stream (same inputs as the GPU run's env 0)."""
    if not available():
        return ABSENT
    from oracle import pyoracle
    sim = RefSim(asset_path(name))
    L = pyoracle.lib()
    dt_model = 0.002
    nsteps = 2000

    def noise(n):
        rate = np.exp(-dt_model / 0.1)
        scale = noise_std * np.sqrt(1 - rate * rate)
        z = np.array([[L.mjo_normal(12345, 0, s, i) for i in range(sim.nu)] for s in range(n)])
        out = np.zeros((n, sim.nu))
        cur = np.zeros(sim.nu)
        for s in range(n):
            cur = rate * cur + scale * z[s]
            out[s] = cur
        return out

    old = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    if old:
        os.sched_setaffinity(0, {sorted(old)[0]})
    try:
        seq = noise(nsteps)
        sim.reset()
        t0 = time.perf_counter()
        sim.step(nsteps, seq)
        rate = nsteps / (time.perf_counter() - t0)
        reps = max(1, int(rate * target_s / nsteps))
        t0 = time.perf_counter()
        for _ in range(reps):
            sim.step(nsteps, seq)
        dt = time.perf_counter() - t0
    finally:
        if old:
            os.sched_setaffinity(0, old)
    return {"value": reps * nsteps / dt, "unit": "env-steps/s", "cores": 1, "kind": "reference", "version": version(),
            "sample": f"real mj_step, 1 env x {reps * nsteps} steps of {name}, one pinned thread, {dt:.1f} s"}
