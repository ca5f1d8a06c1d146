"""Python face of the C-ABI engine (libmjb.so): thin, typed wrappers -- no compute happens here.

``CompiledModel`` <-> ``mjb_model``  (takes the place of the reference's ``mjModelPtr model_``,
/root/reference mujoco_ros/include/mujoco_ros/mujoco_env.h:298), ``Batch`` <-> ``mjb_batch`` (N x
``mjDataPtr data_``, mujoco_env.h:300).  Every method maps 1:1 onto an entry point of include/mjb.h and
raises ``EngineError`` with ``mjb_last_error()`` on a non-zero return.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import binding
from .binding import Field


class EngineError(RuntimeError):
    pass


def _lib():
    return binding.load_library()


def _check(rc, what):
    if rc != 0:
        raise EngineError(f"{what} failed ({rc}): {_lib().mjb_last_error().decode()}")


def device_count():
    return _lib().mjb_device_count()


class CompiledModel:
    def __init__(self, model):
        self.model = model
        self.lib = _lib()
        desc, self._keep = binding.make_desc(model)
        self.ptr = self.lib.mjb_compile(C.byref(desc))
        if not self.ptr:
            raise EngineError("mjb_compile failed: " + self.lib.mjb_last_error().decode())

    def field_size(self, name):
        return self.lib.mjb_field_size(self.ptr, Field.ids[name])

    def derive_mass_params(self, body_mass, body_inertia=None):
        """mj_setConst's constants for new body masses (mjb_derive_mass_params, host-side C++): the packed block of
        mjb_set_env_mass_params.  Needs no device."""
        nb = int(self.model["nbody"])
        bm = np.ascontiguousarray(np.asarray(body_mass, dtype=np.float64).reshape(nb))
        bi = None if body_inertia is None else np.ascontiguousarray(np.asarray(body_inertia, dtype=np.float64).reshape(nb, 3))
        out = np.zeros(self.lib.mjb_env_mass_stride(self.ptr))
        pd = C.POINTER(C.c_double)
        _check(self.lib.mjb_derive_mass_params(self.ptr, bm.ctypes.data_as(pd), None if bi is None else bi.ctypes.data_as(pd),
                                               out.ctypes.data_as(pd)), "mjb_derive_mass_params")
        return out

    @property
    def frame_doubles(self):
        return self.lib.mjb_frame_doubles(self.ptr)

    def close(self):
        if getattr(self, "ptr", None):
            self.lib.mjb_free_model(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Batch:
    """N env instances of one model on one GPU."""

    def __init__(self, cmodel: CompiledModel, nenv: int, device: int = 0):
        self.cm = cmodel
        self.lib = cmodel.lib
        self.nenv = int(nenv)
        self.ptr = self.lib.mjb_make_batch(cmodel.ptr, self.nenv, int(device))
        if not self.ptr:
            raise EngineError("mjb_make_batch failed: " + self.lib.mjb_last_error().decode())

    # ---- stepping
    def step(self, nsteps=1):
        _check(self.lib.mjb_step(self.ptr, int(nsteps)), "mjb_step")

    def step1(self):
        _check(self.lib.mjb_step1(self.ptr), "mjb_step1")

    def step2(self):
        _check(self.lib.mjb_step2(self.ptr), "mjb_step2")

    def forward(self):
        _check(self.lib.mjb_forward(self.ptr), "mjb_forward")

    def reset(self, mask=None):
        if mask is None:
            _check(self.lib.mjb_reset(self.ptr, None), "mjb_reset")
        else:
            m = np.ascontiguousarray(mask, dtype=np.uint8)
            if m.shape != (self.nenv,):
                raise ValueError("mask must have shape (nenv,)")
            _check(self.lib.mjb_reset(self.ptr, m.ctypes.data_as(C.POINTER(C.c_uint8))), "mjb_reset")

    def synchronize(self):
        _check(self.lib.mjb_synchronize(self.ptr), "mjb_synchronize")

    def set_launch(self, lanes_per_env=0, envs_per_block=0):
        _check(self.lib.mjb_set_launch(self.ptr, lanes_per_env, envs_per_block), "mjb_set_launch")

    def warning_count(self):
        """Number of auto-resets (mj_checkPos / Vel / Acc warnings) since the batch was made."""
        n = C.c_uint64(0)
        _check(self.lib.mjb_warning_count(self.ptr, C.byref(n)), "mjb_warning_count")
        return int(n.value)

    def warning(self, which):
        """mjData.warning[which].number summed over the envs (mjb_warning; which = binding.WARN[...] or its int value)."""
        n = C.c_uint64(0)
        _check(self.lib.mjb_warning(self.ptr, binding.WARN.get(which, which), C.byref(n)), "mjb_warning")
        return int(n.value)

    def metrics(self):
        """The 16-double metrics vector of include/mjb.h (mjb_metrics) as a dict + the raw array."""
        out = np.zeros(16)
        _check(self.lib.mjb_metrics(self.ptr, out.ctypes.data_as(C.POINTER(C.c_double))), "mjb_metrics")
        return dict(zip(binding.METRIC_NAMES, out)), out

    def metrics_device_ptr(self):
        p = self.lib.mjb_metrics_device(self.ptr)
        if not p:
            raise EngineError(self.lib.mjb_last_error().decode())
        return p

    def register_collision(self, geom_type1, geom_type2, func):
        """registerCollisionFunction's device-side form: func = 0 default, 1 none, 2 bounding spheres (mjb_register_collision)."""
        _check(self.lib.mjb_register_collision(self.ptr, int(geom_type1), int(geom_type2), int(func)), "mjb_register_collision")

    # ---- per-env model parameters ----
    def set_env_gravity(self, gravity, lo=0, hi=None):
        hi = self.nenv if hi is None else hi
        a = np.ascontiguousarray(gravity, dtype=np.float64).reshape(hi - lo, 3)
        _check(self.lib.mjb_set_env_gravity(self.ptr, lo, hi, a.ctypes.data_as(C.POINTER(C.c_double))), "mjb_set_env_gravity")

    def set_env_geom_friction(self, friction, lo=0, hi=None):
        hi = self.nenv if hi is None else hi
        a = np.ascontiguousarray(friction, dtype=np.float64).reshape(hi - lo, self.cm.model["ngeom"] * 3)
        _check(self.lib.mjb_set_env_geom_friction(self.ptr, lo, hi, a.ctypes.data_as(C.POINTER(C.c_double))),
               "mjb_set_env_geom_friction")

    def set_env_geom_size(self, size, lo=0, hi=None):
        hi = self.nenv if hi is None else hi
        a = np.ascontiguousarray(size, dtype=np.float64).reshape(hi - lo, self.cm.model["ngeom"] * 3)
        _check(self.lib.mjb_set_env_geom_size(self.ptr, lo, hi, a.ctypes.data_as(C.POINTER(C.c_double))), "mjb_set_env_geom_size")

    def set_env_geom_type(self, types, lo=0, hi=None):
        hi = self.nenv if hi is None else hi
        a = np.ascontiguousarray(types, dtype=np.int32).reshape(hi - lo, self.cm.model["ngeom"])
        _check(self.lib.mjb_set_env_geom_type(self.ptr, lo, hi, a.ctypes.data_as(C.POINTER(C.c_int))), "mjb_set_env_geom_type")

    def set_env_equality(self, active=None, data=None, solref=None, solimp=None, lo=0, hi=None):
        """Per-env equality parameters (setEqualityConstraintParameters per env).  Each argument is None (the model's
        values) or an array [hi-lo, neq(, 11 | 2 | 5)]."""
        hi = self.nenv if hi is None else hi
        m, n, neq = self.cm.model, hi - lo, int(self.cm.model["neq"])
        p = np.zeros((n, neq, 19))
        p[:, :, 0] = np.asarray(m["eq_active"], dtype=np.float64) if active is None else np.asarray(active, dtype=np.float64).reshape(n, neq)
        p[:, :, 1:12] = np.asarray(m["eq_data"]).reshape(neq, 11) if data is None else np.asarray(data, dtype=np.float64).reshape(n, neq, 11)
        p[:, :, 12:14] = np.asarray(m["eq_solref"]).reshape(neq, 2) if solref is None else np.asarray(solref, dtype=np.float64).reshape(n, neq, 2)
        p[:, :, 14:19] = np.asarray(m["eq_solimp"]).reshape(neq, 5) if solimp is None else np.asarray(solimp, dtype=np.float64).reshape(n, neq, 5)
        p = np.ascontiguousarray(p)
        _check(self.lib.mjb_set_env_equality(self.ptr, lo, hi, p.ctypes.data_as(C.POINTER(C.c_double))), "mjb_set_env_equality")

    def set_env_body_mass(self, body_mass, body_inertia=None, lo=0, hi=None):
        """Per-env body masses [hi-lo, nbody] (and optionally principal inertias [hi-lo, nbody, 3]): mjb_set_env_body_mass
        derives what mj_setConst would (in C++, inside libmjb) and uploads the packed constants."""
        hi = self.nenv if hi is None else hi
        n = hi - lo
        nb = int(self.cm.model["nbody"])
        bm = np.ascontiguousarray(np.asarray(body_mass, dtype=np.float64).reshape(n, nb))
        bi = None if body_inertia is None else np.ascontiguousarray(np.asarray(body_inertia, dtype=np.float64).reshape(n, nb, 3))
        pd = C.POINTER(C.c_double)
        _check(self.lib.mjb_set_env_body_mass(self.ptr, lo, hi, bm.ctypes.data_as(pd), None if bi is None else bi.ctypes.data_as(pd)),
               "mjb_set_env_body_mass")

    # ---- device-side DefaultRobotHWSim (mjb_hwsim_*) ----
    def hwsim_configure(self, joints):
        """joints: list of dicts(joint=<id>, method=..., kind=..., p, i, d, i_max, i_min, antiwindup, effort_limit, lower, upper)."""
        arr = (binding.HwsimJoint * max(1, len(joints)))()
        for k, j in enumerate(joints):
            arr[k] = binding.HwsimJoint(int(j["joint"]), binding.HW_METHODS[j.get("method", "effort")],
                                        binding.HW_KINDS[j.get("kind", "revolute")], int(j.get("antiwindup", 0)),
                                        float(j.get("p", 0)), float(j.get("i", 0)), float(j.get("d", 0)),
                                        float(j.get("i_max", 0)), float(j.get("i_min", 0)), float(j.get("effort_limit", 0)),
                                        float(j.get("lower", 0)), float(j.get("upper", 0)))
        _check(self.lib.mjb_hwsim_configure(self.ptr, len(joints), arr), "mjb_hwsim_configure")
        self._hw_n = len(joints)

    def hwsim_set_command(self, which, cmd, lo=0, hi=None):
        import numpy as np
        hi = self.nenv if hi is None else hi
        a = np.ascontiguousarray(cmd, dtype=np.float64).reshape(hi - lo, self._hw_n)
        _check(self.lib.mjb_hwsim_set_command(self.ptr, {"position": 0, "velocity": 1, "effort": 2}[which], lo, hi,
                                              a.ctypes.data_as(C.POINTER(C.c_double))), "mjb_hwsim_set_command")

    def hwsim_set_period(self, control_period):
        """Controller cadence of MujocoRosControlPlugin::controlCallback (mjb_hwsim_set_period); <= 0 switches it off."""
        _check(self.lib.mjb_hwsim_set_period(self.ptr, float(control_period)), "mjb_hwsim_set_period")

    def hwsim_estop(self, active):
        _check(self.lib.mjb_hwsim_estop(self.ptr, 1 if active else 0), "mjb_hwsim_estop")

    # ---- sensors-plugin equivalent (mjb_sensor_*) ----
    def sensor_set_noise(self, sensor, set_flag, mean=(0, 0, 0), sigma=(0, 0, 0)):
        """registerNoiseModels: bit k of set_flag = noise on component k; the n-th set bit uses mean[n], sigma[n]."""
        import numpy as np
        mu = np.zeros(3)
        sg = np.zeros(3)
        mu[:len(mean)] = mean
        sg[:len(sigma)] = sigma
        pd = C.POINTER(C.c_double)
        _check(self.lib.mjb_sensor_set_noise(self.ptr, int(sensor), int(set_flag), mu.ctypes.data_as(pd), sg.ctypes.data_as(pd)),
               "mjb_sensor_set_noise")

    def sensor_pack(self, seed=0):
        _check(self.lib.mjb_sensor_pack(self.ptr, int(seed)), "mjb_sensor_pack")

    def sensor_messages(self, which="value", lo=0, hi=None):
        """float32 [hi-lo, nsensordata]: 'value' (with noise models applied) or 'truth' (sensordata / cutoff)."""
        import numpy as np
        hi = self.nenv if hi is None else hi
        out = np.empty((hi - lo, self.cm.model["nsensordata"]), dtype=np.float32)
        _check(self.lib.mjb_sensor_get(self.ptr, 0 if which == "value" else 1, lo, hi, out.ctypes.data_as(C.POINTER(C.c_float))),
               "mjb_sensor_get")
        return out

    def set_keep_frame(self, on=True):
        """Fused step() also leaves the derived fields of its last step readable through get()."""
        _check(self.lib.mjb_set_keep_frame(self.ptr, 1 if on else 0), "mjb_set_keep_frame")

    def set_ctrl_noise(self, std, rate, seed=0, env_offset=0):
        _check(self.lib.mjb_set_ctrl_noise(self.ptr, float(std), float(rate), int(seed), int(env_offset)),
               "mjb_set_ctrl_noise")

    def set_stats(self, on=True):
        """Start (and zero) / stop the device-side workload statistics of the constrained kernels (mjb_set_stats)."""
        _check(self.lib.mjb_set_stats(self.ptr, 1 if on else 0), "mjb_set_stats")

    def stats(self):
        """dict(evaluations, solver_iter_mean, ncon_{mean,p50,p99,max}, nefc_{mean,p50,p99,max}, rows_gt64_share, nefc_hist, ncon_hist)."""
        out = (C.c_ulonglong * 388)()
        _check(self.lib.mjb_get_stats(self.ptr, out), "mjb_get_stats")
        a = np.array(out[:], dtype=np.float64)
        n = a[0]
        he, hc = a[2:259], a[259:388]

        def summary(h):
            tot = h.sum()
            if tot <= 0:
                return 0.0, 0, 0, 0
            c = np.cumsum(h)
            q = lambda f: int(np.searchsorted(c, f * tot, side="left"))
            return float((h * np.arange(len(h))).sum() / tot), q(0.5), q(0.99), int(np.nonzero(h)[0].max())
        em, e50, e99, emx = summary(he)
        cm, c50, c99, cmx = summary(hc)
        return {"evaluations": int(n), "solver_iter_mean": float(a[1] / n) if n else 0.0,
                "ncon_mean": cm, "ncon_p50": c50, "ncon_p99": c99, "ncon_max": cmx,
                "nefc_mean": em, "nefc_p50": e50, "nefc_p99": e99, "nefc_max": emx,
                "rows_gt64_share": float(he[65:].sum() / he.sum()) if he.sum() else 0.0,
                "nefc_hist": he.astype(np.int64), "ncon_hist": hc.astype(np.int64)}

    def fused_frame(self):
        """(id, bytes) of the fused frame the batch's launches currently run on: 1 default, 2 wide (mjb_fused_frame)."""
        fid = int(self.lib.mjb_fused_frame(self.ptr))
        return fid, int(self.lib.mjb_frame_bytes(self.cm.ptr, fid))

    def noise_mode(self):
        """How the last fused launch got its ctrl-noise normals (mjb_noise_mode)."""
        return ("in-kernel", "same-stream", "side-stream")[int(self.lib.mjb_noise_mode(self.ptr))]

    def set_lane_env(self, mode):
        """-1 automatic, 0 never, 1 whenever eligible: the lane = env form of the unconstrained fused step (mjb_set_lane_env)."""
        _check(self.lib.mjb_set_lane_env(self.ptr, int(mode)), "mjb_set_lane_env")

    def lane_env_info(self):
        """(compiled-in topology index or -1, the last fused launch ran the lane = env kernel)."""
        used = C.c_int(0)
        topo = int(self.lib.mjb_lane_env_info(self.ptr, C.byref(used)))
        return topo, bool(used.value)

    def set_sensors_every_step(self, on=True):
        """The lane = env kernel evaluates the sensor stages at every step of a fused launch instead of its last one (mjb_set_sensors_every_step)."""
        _check(self.lib.mjb_set_sensors_every_step(self.ptr, 1 if on else 0), "mjb_set_sensors_every_step")

    def set_split_step(self, mode):
        """-1 automatic, 0 never, 1 whenever eligible: the split step of plain-PGS models -- smooth stages one env per lane, constraint stages one env
        per wavefront (mjb_set_split_step)."""
        _check(self.lib.mjb_set_split_step(self.ptr, int(mode)), "mjb_set_split_step")

    def split_step_info(self):
        """(topology index or -1, the last fused launch ran as a split step, its env slices)."""
        used, slices = C.c_int(0), C.c_int(0)
        topo = int(self.lib.mjb_split_step_info(self.ptr, C.byref(used), C.byref(slices)))
        return topo, bool(used.value), int(slices.value)

    def set_lane_env_form(self, form):
        """Process-wide: -1 by batch size, 0 one wavefront per 64 envs, 1 two (position / velocity halves), 2 two, pipelined (mjb_lane_env_set_form)."""
        return int(self.lib.mjb_lane_env_set_form(int(form)))

    def lane_env_last_form(self):
        return int(self.lib.mjb_lane_env_last_form())

    def lane_env_error(self):
        """Why the hiprtc build of this process's last lane = env topology was not available ('' if none failed)."""
        return self.lib.mjb_lane_env_error().decode()

    def time_steps(self, nsteps, nlaunch):
        ms = C.c_double(0)
        _check(self.lib.mjb_time_steps(self.ptr, int(nsteps), int(nlaunch), C.byref(ms)), "mjb_time_steps")
        return ms.value

    # ---- data access (env-major numpy arrays)
    def get(self, name, lo=0, hi=None):
        hi = self.nenv if hi is None else hi
        fid = Field.ids[name]
        n = self.cm.field_size(name)
        if Field.kinds[name] == "DI":
            out = np.zeros((hi - lo, n), dtype=np.int32)
            _check(self.lib.mjb_get_int(self.ptr, fid, lo, hi, out.ctypes.data_as(C.POINTER(C.c_int))), "mjb_get_int")
        else:
            out = np.zeros((hi - lo, n), dtype=np.float64)
            _check(self.lib.mjb_get(self.ptr, fid, lo, hi, out.ctypes.data_as(C.POINTER(C.c_double))), "mjb_get")
        return out

    def set(self, name, value, lo=0, hi=None):
        hi = self.nenv if hi is None else hi
        n = self.cm.field_size(name)
        if n == 0:  # (a world without this field -- empty_world has no dof: nothing to write)
            return
        v = np.ascontiguousarray(np.broadcast_to(np.asarray(value, dtype=np.float64).reshape(-1, n) if np.ndim(value) > 1
                                                 else np.asarray(value, dtype=np.float64), (hi - lo, n)))
        _check(self.lib.mjb_set(self.ptr, Field.ids[name], lo, hi, v.ctypes.data_as(C.POINTER(C.c_double))), "mjb_set")

    def device_ptr(self, name):
        p = self.lib.mjb_device_ptr(self.ptr, Field.ids[name])
        if not p:
            raise EngineError(self.lib.mjb_last_error().decode())
        return p

    @property
    def stream(self):
        return self.lib.mjb_get_stream(self.ptr)

    def close(self):
        if getattr(self, "ptr", None):
            self.lib.mjb_free_batch(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
