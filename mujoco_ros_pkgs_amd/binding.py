"""ctypes mirror of the C-ABI in ``include/mjb.h`` (libmjb.so).

The struct layouts are generated from the same X-macro tables the C side includes
(``include/mjb_model_fields.def``, ``include/mjb_data_fields.def``), so the two cannot drift.
There is no Python/CPU fallback: if the shared library is missing, importing the engine raises.
"""
from __future__ import annotations

import ctypes as C
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(_HERE)
INCLUDE = os.path.join(REPO, "include")
LIB_PATH = os.path.join(_HERE, "csrc", "libmjb.so")

_MODEL_RE = re.compile(r"^\s*MJB_(SIZE|OPT_I|OPT_D|ARR_I|ARR_D)\(\s*([A-Za-z0-9_]+)\s*(?:,\s*([A-Za-z0-9_]+)\s*)?(?:,\s*([A-Za-z0-9_]+)\s*)?\)")
_DATA_RE = re.compile(r"^\s*MJB_(DS|DD|DD2|DI)\(\s*([A-Za-z0-9_]+)\s*,\s*([A-Za-z0-9_]+)\s*,\s*([A-Za-z0-9_]+)\s*\)")


def _parse(path, rx):
    out = []
    with open(path) as f:
        for line in f:
            mm = rx.match(line)
            if mm:
                out.append(mm.groups())
    return out


MODEL_FIELDS = _parse(os.path.join(INCLUDE, "mjb_model_fields.def"), _MODEL_RE)
DATA_FIELDS = _parse(os.path.join(INCLUDE, "mjb_data_fields.def"), _DATA_RE)

SIZE_NAMES = [n for k, n, _, _ in MODEL_FIELDS if k == "SIZE"]


def _model_struct_fields():
    fl = []
    for kind, name, a, b in MODEL_FIELDS:
        if kind in ("SIZE", "OPT_I"):
            fl.append((name, C.c_int))
        elif kind == "OPT_D":
            fl.append((name, C.c_double * int(a)))
        elif kind == "ARR_I":
            fl.append((name, C.POINTER(C.c_int)))
        else:
            fl.append((name, C.POINTER(C.c_double)))
    return fl


class ModelDesc(C.Structure):
    _fields_ = _model_struct_fields()


class Field:
    """Data-field ids and metadata (order of mjb_data_fields.def == enum mjb_field)."""

    names = [n for _, n, _, _ in DATA_FIELDS]
    ids = {n: i for i, n in enumerate(names)}
    kinds = {n: k for k, n, _, _ in DATA_FIELDS}

    @staticmethod
    def dim(model, name):
        for k, n, rows, cols in DATA_FIELDS:
            if n == name:
                r = 1 if rows == "one" else int(model[rows])
                c = int(model[cols]) if k == "DD2" else int(cols)
                return r * c
        raise KeyError(name)


def make_desc(model):
    """Build a ``ModelDesc`` pointing at (contiguous copies of) the arrays of a compiled model dict.
    Returns (desc, keepalive)."""
    d = ModelDesc()
    keep = []
    for kind, name, a, b in MODEL_FIELDS:
        if kind in ("SIZE", "OPT_I"):
            setattr(d, name, int(model[name]))
        elif kind == "OPT_D":
            v = np.atleast_1d(np.asarray(model[name], dtype=np.float64))
            arr = getattr(d, name)
            for i in range(int(a)):
                arr[i] = float(v[i])
        else:
            rows = int(model[a])
            cols = int(b)
            dt = np.int32 if kind == "ARR_I" else np.float64
            arr = np.ascontiguousarray(np.asarray(model[name], dtype=dt).reshape(-1))
            if arr.size != rows * cols:
                raise ValueError(f"model field {name}: expected {rows}x{cols}, got {arr.size}")
            if arr.size == 0:
                arr = np.zeros(1, dt)
            keep.append(arr)
            ct = C.c_int if kind == "ARR_I" else C.c_double
            setattr(d, name, arr.ctypes.data_as(C.POINTER(ct)))
    return d, keep


class HwsimJoint(C.Structure):
    """mjb_hwsim_joint of include/mjb.h"""
    _fields_ = [("joint", C.c_int), ("method", C.c_int), ("kind", C.c_int), ("antiwindup", C.c_int),
                ("p", C.c_double), ("i", C.c_double), ("d", C.c_double), ("i_max", C.c_double), ("i_min", C.c_double),
                ("effort_limit", C.c_double), ("lower", C.c_double), ("upper", C.c_double)]


WARN = {"inertia": 0, "contactfull": 1, "cnstrfull": 2, "vgeomfull": 3, "badqpos": 4, "badqvel": 5, "badqacc": 6, "badctrl": 7}
# layout of the mjb_metrics vector: [0:8] additive, [8:16] maxima
METRIC_NAMES = ["env_steps", "auto_resets", "contactfull", "cnstrfull", "energy_potential", "energy_kinetic", "nenv", "sum_reserved",
                "max_abs_qacc", "max_abs_qvel", "max_time", "max_r3", "max_r4", "max_r5", "max_r6", "max_r7"]

HW_METHODS = {"effort": 0, "position": 1, "position_pid": 2, "velocity": 3, "velocity_pid": 4}
HW_KINDS = {"revolute": 0, "continuous": 1, "prismatic": 2}

_lib = None


def check_desc_size(lib, fn, path):
    """A library built from another revision of include/mjb_model_fields.def would read the model desc at the wrong offsets: refuse it."""
    try:
        f = getattr(lib, fn)
    except AttributeError:
        raise OSError(f"{path} is stale (no {fn}): run `python -c 'import __graft_entry__ as g; g.build()'`")
    f.restype = C.c_int
    if f() != C.sizeof(ModelDesc):
        raise OSError(f"{path} was built with another mjb_model_desc ({f()} bytes, include/mjb_model_fields.def gives {C.sizeof(ModelDesc)}): "
                      "run `python -c 'import __graft_entry__ as g; g.build()'`")


def load_library(path=None):
    """Load libmjb.so and declare every entry point of include/mjb.h.  Raises OSError if missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("MJB_LIBRARY") or LIB_PATH  # MJB_LIBRARY: an alternative build (e.g. libmjb_prof.so)
    if not os.path.exists(p):
        raise OSError(
            f"{p} not found: the HIP engine is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback).")
    lib = C.CDLL(p, mode=C.RTLD_GLOBAL)
    check_desc_size(lib, "mjb_model_desc_size", p)
    vp, ci, cd = C.c_void_p, C.c_int, C.c_double
    sig = {
        "mjb_last_error": (C.c_char_p, []),
        "mjb_model_desc_size": (ci, []),
        "mjb_version": (ci, []),
        "mjb_device_count": (ci, []),
        "mjb_compile": (vp, [C.POINTER(ModelDesc)]),
        "mjb_free_model": (None, [vp]),
        "mjb_field_size": (ci, [vp, ci]),
        "mjb_field_is_int": (ci, [ci]),
        "mjb_field_is_state": (ci, [ci]),
        "mjb_field_name": (C.c_char_p, [ci]),
        "mjb_frame_doubles": (ci, [vp]),
        "mjb_frame_bytes": (ci, [vp, ci]),
        "mjb_frame_offset": (ci, [vp, ci, ci]),
        "mjb_make_batch": (vp, [vp, ci, ci]),
        "mjb_free_batch": (None, [vp]),
        "mjb_nenv": (ci, [vp]),
        "mjb_set_launch": (ci, [vp, ci, ci]),
        "mjb_set_keep_frame": (ci, [vp, ci]),
        "mjb_set_env_gravity": (ci, [vp, ci, ci, C.POINTER(cd)]),
        "mjb_set_env_geom_friction": (ci, [vp, ci, ci, C.POINTER(cd)]),
        "mjb_set_env_geom_size": (ci, [vp, ci, ci, C.POINTER(cd)]),
        "mjb_set_env_geom_type": (ci, [vp, ci, ci, C.POINTER(ci)]),
        "mjb_set_env_equality": (ci, [vp, ci, ci, C.POINTER(cd)]),
        "mjb_env_mass_stride": (ci, [vp]),
        "mjb_set_env_mass_params": (ci, [vp, ci, ci, C.POINTER(cd)]),
        "mjb_step1_prefix": (ci, [vp, ci]),
        "mjb_step_rest": (ci, [vp, ci]),
        "mjb_step2_prefix": (ci, [vp, ci]),
        "mjb_step21_prefix": (ci, [vp, ci]),
        "mjb_step2_rk_prefix": (ci, [vp, ci, ci]),
        "mjb_get_packed": (ci, [vp, ci, C.POINTER(ci), ci, ci, C.POINTER(cd)]),
        "mjb_set_packed": (ci, [vp, ci, C.POINTER(ci), ci, ci, C.POINTER(cd)]),
        "mjb_derive_mass_params": (ci, [vp, C.POINTER(cd), C.POINTER(cd), C.POINTER(cd)]),
        "mjb_set_env_body_mass": (ci, [vp, ci, ci, C.POINTER(cd), C.POINTER(cd)]),
        "mjb_hwsim_configure": (ci, [vp, ci, C.POINTER(HwsimJoint)]),
        "mjb_hwsim_set_command": (ci, [vp, ci, ci, ci, C.POINTER(cd)]),
        "mjb_hwsim_command_ptr": (vp, [vp, ci]),
        "mjb_hwsim_estop": (ci, [vp, ci]),
        "mjb_hwsim_set_period": (ci, [vp, cd]),
        "mjb_sensor_set_noise": (ci, [vp, ci, ci, C.POINTER(cd), C.POINTER(cd)]),
        "mjb_sensor_pack": (ci, [vp, C.c_uint64]),
        "mjb_sensor_get": (ci, [vp, ci, ci, ci, C.POINTER(C.c_float)]),
        "mjb_sensor_device_ptr": (vp, [vp, ci]),
        "mjb_step": (ci, [vp, ci]),
        "mjb_step1": (ci, [vp]),
        "mjb_step2": (ci, [vp]),
        "mjb_forward": (ci, [vp]),
        "mjb_reset": (ci, [vp, C.POINTER(C.c_uint8)]),
        "mjb_get": (ci, [vp, ci, ci, ci, C.POINTER(cd)]),
        "mjb_set": (ci, [vp, ci, ci, ci, C.POINTER(cd)]),
        "mjb_get_int": (ci, [vp, ci, ci, ci, C.POINTER(ci)]),
        "mjb_register_collision": (ci, [vp, ci, ci, ci]),
        "mjb_get_many": (ci, [vp, ci, C.POINTER(ci), ci, ci, C.POINTER(C.POINTER(cd))]),
        "mjb_set_many": (ci, [vp, ci, C.POINTER(ci), ci, ci, C.POINTER(C.POINTER(cd))]),
        "mjb_host_register": (ci, [vp, C.c_ulonglong]),
        "mjb_host_unregister": (ci, [vp]),
        "mjb_device_ptr": (vp, [vp, ci]),
        "mjb_set_ctrl_noise": (ci, [vp, cd, cd, C.c_uint64, C.c_int64]),
        "mjb_noise_mode": (ci, [vp]),
        "mjb_fused_frame": (ci, [vp]),
        "mjb_set_lane_env": (ci, [vp, ci]),
        "mjb_set_sensors_every_step": (ci, [vp, ci]),
        "mjb_lane_env_info": (ci, [vp, C.POINTER(ci)]),
        "mjb_lane_env_error": (C.c_char_p, []),
        "mjb_lane_env_set_form": (ci, [ci]),
        "mjb_lane_env_last_form": (ci, []),
        "mjb_model_lane_env": (ci, [vp]),
        "mjb_lane_env_jit_counts": (None, [C.POINTER(ci), C.POINTER(ci)]),
        "mjb_set_split_step": (ci, [vp, ci]),
        "mjb_split_step_info": (ci, [vp, C.POINTER(ci), C.POINTER(ci)]),
        "mjb_model_split_step": (ci, [vp]),
        "mjb_set_stats": (ci, [vp, ci]),
        "mjb_get_stats": (ci, [vp, C.POINTER(C.c_ulonglong)]),
        "mjb_get_stream": (vp, [vp]),
        "mjb_set_stream": (ci, [vp, vp]),
        "mjb_synchronize": (ci, [vp]),
        "mjb_time_steps": (ci, [vp, ci, ci, C.POINTER(cd)]),
        "mjb_warning_count": (ci, [vp, C.POINTER(C.c_uint64)]),
        "mjb_warning": (ci, [vp, ci, C.POINTER(C.c_uint64)]),
        "mjb_metrics": (ci, [vp, C.POINTER(cd)]),
        "mjb_metrics_device": (vp, [vp]),
        "mjb_debug_profile": (ci, [vp, C.POINTER(C.c_uint64), ci]),
        "mjb_debug_profile_window": (ci, [vp, ci]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    lib._mjb_symbols = sorted(sig)
    if path is None:
        _lib = lib
    return lib


def header_symbols():
    """Every function name declared in include/mjb.h (for the symbol-export test)."""
    txt = open(os.path.join(INCLUDE, "mjb.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mjb_[a-z0-9_]+)\s*\(", txt)))
