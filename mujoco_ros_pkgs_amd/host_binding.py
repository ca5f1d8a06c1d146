"""ctypes face of the host runtime (libmjr_host.so, include/mjr_host.h): the ROS-free batched mirror of the
reference's ``MujocoEnv`` + ``MujocoPlugin`` layer.  ``HostEnv`` methods map 1:1 onto the reference members
named in mjr_host.h."""
from __future__ import annotations

import ctypes as C
import json
import os

import numpy as np

from . import binding

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host", "libmjr_host.so")

OBJ_BODY, OBJ_JOINT, OBJ_GEOM, OBJ_SITE, OBJ_ACTUATOR, OBJ_SENSOR = 1, 3, 5, 6, 18, 19


class Names(C.Structure):
    _fields_ = [(k, C.POINTER(C.c_char_p)) for k in ("body", "joint", "geom", "site", "sensor", "actuator")]


class SensorRecord(C.Structure):
    """mjr_sensor_record (include/mjr_host.h)."""
    _fields_ = [("name", C.c_char * 64), ("frame_id", C.c_char * 64), ("kind", C.c_int), ("env", C.c_int),
                ("has_truth", C.c_int), ("stamp", C.c_double), ("value", C.c_float * 4), ("truth", C.c_float * 4)]


SENSOR_KINDS = ("scalar", "vector3", "point", "quaternion")

BACKEND_FACTORY = C.CFUNCTYPE(C.c_void_p, C.POINTER(binding.ModelDesc), C.c_int, C.c_int, C.c_void_p)

_lib = None


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    binding.load_library()  # libmjb.so first (RTLD_GLOBAL)
    if not os.path.exists(LIB_PATH):
        raise OSError(f"{LIB_PATH} not found; run __graft_entry__.build()")
    L = C.CDLL(LIB_PATH)
    vp, ci, cs = C.c_void_p, C.c_int, C.c_char_p
    sig = {
        "mjr_last_error": (cs, []),
        "mjr_make_mjb_backend": (vp, [C.POINTER(binding.ModelDesc), ci, ci, vp]),
        "mjr_env_create": (vp, [cs, cs]),
        "mjr_env_destroy": (None, [vp]),
        "mjr_env_set_param": (ci, [vp, cs, cs]),
        "mjr_env_delete_param": (ci, [vp, cs]),
        "mjr_env_queue_model": (ci, [vp, C.POINTER(binding.ModelDesc), C.POINTER(Names), ci, ci, vp, vp]),
        "mjr_env_queue_model_devices": (ci, [vp, C.POINTER(binding.ModelDesc), C.POINTER(Names), ci, C.POINTER(ci), ci, vp, vp]),
        "mjr_env_start": (ci, [vp]),
        "mjr_env_shutdown": (ci, [vp]),
        "mjr_env_operational_status": (ci, [vp]),
        "mjr_env_pending_steps": (ci, [vp]),
        "mjr_env_is_physics_running": (ci, [vp]),
        "mjr_env_is_event_running": (ci, [vp]),
        "mjr_env_model_valid": (ci, [vp]),
        "mjr_env_load_error": (cs, [vp]),
        "mjr_env_step": (ci, [vp, ci, ci]),
        "mjr_env_toggle_paused": (ci, [vp, ci, cs]),
        "mjr_env_step_goal": (ci, [vp, ci, C.POINTER(ci)]),
        "mjr_env_reset_request": (ci, [vp]),
        "mjr_env_set_pause": (ci, [vp, ci, cs]),
        "mjr_env_get_setting": (ci, [vp, cs]),
        "mjr_env_set_setting": (ci, [vp, cs, ci]),
        "mjr_env_set_ctrl_noise": (ci, [vp, C.c_double, C.c_double]),
        "mjr_env_sim_time": (C.c_double, [vp]),
        "mjr_env_register_collision_function": (C.c_int, [vp, C.c_int, C.c_int, C.c_int]),
        "mjr_env_data_time": (C.c_double, [vp]),
        "mjr_env_step_count": (C.c_ulonglong, [vp]),
        "mjr_env_nenv": (ci, [vp]),
        "mjr_env_name2id": (ci, [vp, ci, cs]),
        "mjr_env_get_field": (ci, [vp, ci, ci, C.POINTER(C.c_double)]),
        "mjr_env_set_field": (ci, [vp, ci, ci, C.POINTER(C.c_double)]),
        "mjr_env_num_plugins": (ci, [vp]),
        "mjr_env_num_cb_ready_plugins": (ci, [vp]),
        "mjr_env_test_plugin_flag": (ci, [vp, ci, cs, ci]),
        "mjr_env_notify_geom_changed": (ci, [vp, ci]),
        "mjr_env_set_callback_envs": (ci, [vp, ci]),
        "mjr_sensors_num_records": (ci, [vp, ci, ci]),
        "mjr_sensors_get_record": (ci, [vp, ci, ci, ci, C.POINTER(SensorRecord)]),
        "mjr_sensors_register_noise": (ci, [vp, ci, cs, ci, C.POINTER(C.c_double), C.POINTER(C.c_double), cs]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    L._mjr_symbols = sorted(sig)
    _lib = L
    return L


class HostEnv:
    """mujoco_ros::MujocoEnv over N env instances."""

    def __init__(self, params=None, admin_hash=""):
        self.L = load_library()
        self.ptr = self.L.mjr_env_create(admin_hash.encode(), json.dumps(params or {}).encode())
        if not self.ptr:
            raise RuntimeError(self.L.mjr_last_error().decode())
        self._keep = []
        self.model = None

    # ---- model
    def queue_model(self, model, nenv=1, device=0, backend_factory=None, devices=None):
        """``devices``: shard the batch over these devices (contiguous env blocks, one backend each)."""
        desc, keep = binding.make_desc(model)
        names = Names()
        for kind in ("body", "joint", "geom", "site", "sensor", "actuator"):
            lst = model["names"].get(kind, [])
            arr = (C.c_char_p * max(1, len(lst)))(*[s.encode() for s in lst])
            keep.append(arr)
            setattr(names, kind, C.cast(arr, C.POINTER(C.c_char_p)))
        self._keep += [desc, keep, names]
        self.model = model
        fac = C.cast(backend_factory, C.c_void_p) if backend_factory is not None else None
        if devices is not None:
            dv = (C.c_int * len(devices))(*[int(x) for x in devices])
            rc = self.L.mjr_env_queue_model_devices(self.ptr, C.byref(desc), C.byref(names), nenv, dv, len(devices), fac, None)
        else:
            rc = self.L.mjr_env_queue_model(self.ptr, C.byref(desc), C.byref(names), nenv, device, fac, None)
        if rc != 0:
            raise RuntimeError("queue_model failed")

    def start(self):
        self.L.mjr_env_start(self.ptr)

    def shutdown(self):
        if self.ptr:
            self.L.mjr_env_shutdown(self.ptr)

    def close(self):
        if self.ptr:
            self.L.mjr_env_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_param(self, key, value):
        assert self.L.mjr_env_set_param(self.ptr, key.encode(), json.dumps(value).encode()) == 0

    def delete_param(self, key):
        self.L.mjr_env_delete_param(self.ptr, key.encode())

    # ---- status / requests
    operational_status = property(lambda self: self.L.mjr_env_operational_status(self.ptr))
    pending_steps = property(lambda self: self.L.mjr_env_pending_steps(self.ptr))
    physics_running = property(lambda self: self.L.mjr_env_is_physics_running(self.ptr))
    event_running = property(lambda self: self.L.mjr_env_is_event_running(self.ptr))
    model_valid = property(lambda self: bool(self.L.mjr_env_model_valid(self.ptr)))
    load_error = property(lambda self: self.L.mjr_env_load_error(self.ptr).decode())
    sim_time = property(lambda self: self.L.mjr_env_sim_time(self.ptr))
    data_time = property(lambda self: self.L.mjr_env_data_time(self.ptr))
    step_count = property(lambda self: self.L.mjr_env_step_count(self.ptr))
    nenv = property(lambda self: self.L.mjr_env_nenv(self.ptr))

    def step(self, n=1, blocking=True):
        return bool(self.L.mjr_env_step(self.ptr, n, int(blocking)))

    def toggle_paused(self, paused, admin_hash=""):
        return bool(self.L.mjr_env_toggle_paused(self.ptr, int(paused), admin_hash.encode()))

    def set_pause(self, paused, admin_hash=""):
        return bool(self.L.mjr_env_set_pause(self.ptr, int(paused), admin_hash.encode()))

    def step_goal(self, n):
        pre = C.c_int(0)
        ok = self.L.mjr_env_step_goal(self.ptr, n, C.byref(pre))
        return bool(ok), bool(pre.value)

    def reset_request(self):
        self.L.mjr_env_reset_request(self.ptr)

    def setting(self, name):
        return self.L.mjr_env_get_setting(self.ptr, name.encode())

    def set_setting(self, name, value):
        assert self.L.mjr_env_set_setting(self.ptr, name.encode(), int(value)) == 0

    def set_ctrl_noise(self, std, rate):
        self.L.mjr_env_set_ctrl_noise(self.ptr, std, rate)

    def name2id(self, objtype, name):
        return self.L.mjr_env_name2id(self.ptr, objtype, name.encode())

    def get_field(self, name, env=0):
        n = binding.Field.dim(self.model, name)
        out = np.zeros(max(n, 1))
        rc = self.L.mjr_env_get_field(self.ptr, binding.Field.ids[name], env, out.ctypes.data_as(C.POINTER(C.c_double)))
        if rc != 0:
            raise RuntimeError(f"get_field({name}) failed")
        return out[:n]

    def set_field(self, name, value, env=0):
        v = np.ascontiguousarray(value, dtype=np.float64)
        rc = self.L.mjr_env_set_field(self.ptr, binding.Field.ids[name], env, v.ctypes.data_as(C.POINTER(C.c_double)))
        if rc != 0:
            raise RuntimeError(f"set_field({name}) failed")

    # ---- plugins
    num_plugins = property(lambda self: self.L.mjr_env_num_plugins(self.ptr))
    num_cb_ready_plugins = property(lambda self: self.L.mjr_env_num_cb_ready_plugins(self.ptr))

    def plugin_flag(self, i, name, clear=False):
        return self.L.mjr_env_test_plugin_flag(self.ptr, i, name.encode(), int(clear))

    def notify_geom_changed(self, geom_id):
        self.L.mjr_env_notify_geom_changed(self.ptr, geom_id)

    def register_collision_function(self, geom_type1, geom_type2, func):
        """MujocoEnv::registerCollisionFunction: 0 first registration, 1 duplicate (the reference warns), -1 refused."""
        return self.L.mjr_env_register_collision_function(self.ptr, int(geom_type1), int(geom_type2), int(func))

    def set_callback_envs(self, n):
        self.L.mjr_env_set_callback_envs(self.ptr, n)

    # ---- sensors plugin (mujoco_ros_sensors/MujocoRosSensorsPlugin)
    def sensor_records(self, plugin=0, env=0):
        """The messages the plugin would have published in its last lastStageCallback: {name: dict}."""
        n = self.L.mjr_sensors_num_records(self.ptr, plugin, env)
        if n < 0:
            raise RuntimeError("plugin %d is not a sensors plugin" % plugin)
        out = {}
        for k in range(n):
            r = SensorRecord()
            if self.L.mjr_sensors_get_record(self.ptr, plugin, env, k, C.byref(r)) != 0:
                raise RuntimeError("mjr_sensors_get_record failed")
            dim = {0: 1, 1: 3, 2: 3, 3: 4}[r.kind]
            out[r.name.decode()] = dict(frame_id=r.frame_id.decode(), kind=SENSOR_KINDS[r.kind], stamp=r.stamp, env=r.env,
                                        value=np.array(r.value[:dim], dtype=np.float32),
                                        truth=np.array(r.truth[:dim], dtype=np.float32) if r.has_truth else None)
        return out

    def register_noise_model(self, sensor_name, set_flag, mean, std, admin_hash="", plugin=0):
        """RegisterSensorNoiseModels for one model; returns the service's `success`."""
        mu = np.zeros(3)
        sg = np.zeros(3)
        mu[:len(mean)] = mean
        sg[:len(std)] = std
        pd = C.POINTER(C.c_double)
        rc = self.L.mjr_sensors_register_noise(self.ptr, plugin, sensor_name.encode(), int(set_flag), mu.ctypes.data_as(pd),
                                               sg.ctypes.data_as(pd), admin_hash.encode())
        if rc < 0:
            raise RuntimeError("plugin %d is not a sensors plugin" % plugin)
        return bool(rc)
