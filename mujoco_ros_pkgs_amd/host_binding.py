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
if os.environ.get("MJR_HOST_LIBRARY"):  # another build of the same runtime (libmjr_host_tsan.so / _asan.so: tools/run_sanitizers.sh)
    LIB_PATH = os.path.abspath(os.environ["MJR_HOST_LIBRARY"])

OBJ_BODY, OBJ_JOINT, OBJ_GEOM, OBJ_SITE, OBJ_ACTUATOR, OBJ_SENSOR = 1, 3, 5, 6, 18, 19


NAME_KINDS = ("body", "joint", "geom", "site", "sensor", "actuator", "equality", "tendon")


class Names(C.Structure):
    _fields_ = [(k, C.POINTER(C.c_char_p)) for k in NAME_KINDS]


class BodyStateC(C.Structure):
    """mjr_body_state"""
    _fields_ = [("name", C.c_char * 64), ("mass", C.c_double), ("pose", C.c_double * 7), ("pose_frame", C.c_char * 64),
                ("twist", C.c_double * 6), ("twist_frame", C.c_char * 64)]


class GeomPropertiesC(C.Structure):
    """mjr_geom_properties"""
    _fields_ = [("name", C.c_char * 64), ("type", C.c_int), ("body_mass", C.c_double), ("friction", C.c_double * 3),
                ("size", C.c_double * 3)]


class EqParametersC(C.Structure):
    """mjr_eq_parameters"""
    _fields_ = [("name", C.c_char * 64), ("element1", C.c_char * 64), ("element2", C.c_char * 64), ("type", C.c_int),
                ("active", C.c_int), ("anchor", C.c_double * 3), ("relpose", C.c_double * 7), ("torquescale", C.c_double),
                ("polycoef", C.c_double * 5), ("dmin", C.c_double), ("dmax", C.c_double), ("width", C.c_double),
                ("midpoint", C.c_double), ("power", C.c_double), ("timeconst", C.c_double), ("dampratio", C.c_double)]


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
    binding.check_desc_size(L, "mjr_model_desc_size", LIB_PATH)
    vp, ci, cs = C.c_void_p, C.c_int, C.c_char_p
    sig = {
        "mjr_last_error": (cs, []),
        "mjr_model_desc_size": (ci, []),
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
        "mjr_env_set_body_state": (ci, [vp, C.POINTER(BodyStateC), ci, ci, ci, ci, cs, ci, ci, C.c_char_p, ci]),
        "mjr_env_get_body_state": (ci, [vp, cs, cs, ci, C.POINTER(BodyStateC), C.c_char_p, ci]),
        "mjr_env_set_geom_properties": (ci, [vp, C.POINTER(GeomPropertiesC), ci, ci, ci, ci, cs, ci, ci, C.c_char_p, ci]),
        "mjr_env_get_geom_properties": (ci, [vp, cs, cs, ci, C.POINTER(GeomPropertiesC), C.c_char_p, ci]),
        "mjr_env_set_gravity": (ci, [vp, C.POINTER(C.c_double), cs, ci, ci, C.c_char_p, ci]),
        "mjr_env_get_gravity": (ci, [vp, cs, ci, C.POINTER(C.c_double), C.c_char_p, ci]),
        "mjr_env_set_eq_parameters": (ci, [vp, C.POINTER(EqParametersC), ci, cs, ci, ci, C.c_char_p, ci]),
        "mjr_env_get_eq_parameters": (ci, [vp, C.POINTER(C.c_char_p), ci, cs, ci, C.POINTER(EqParametersC), C.POINTER(ci), C.c_char_p, ci]),
        "mjr_env_reload": (ci, [vp, C.POINTER(binding.ModelDesc), C.POINTER(Names), ci, ci, vp, vp, C.c_char_p, ci]),
        "mjr_env_loading_request_state": (ci, [vp, C.c_char_p, ci]),
        "mjr_env_load_initial_joint_states": (ci, [vp]),
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
        desc, names, keep = self._desc_and_names(model)
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

    @staticmethod
    def _desc_and_names(model):
        desc, keep = binding.make_desc(model)
        names = Names()
        for kind in NAME_KINDS:
            lst = [s or "" for s in model["names"].get(kind, [])]
            arr = (C.c_char_p * max(1, len(lst)))(*[s.encode() for s in lst])
            keep.append(arr)
            setattr(names, kind, C.cast(arr, C.POINTER(C.c_char_p)))
        return desc, names, keep

    def start(self):
        self.L.mjr_env_start(self.ptr)

    # ---- model / body-state services (host/services.cpp <- callbacks.cpp:177-201, 210-592, 641-897); each returns
    # (success, status_message[, payload]).  env_hi < 0 = every env
    _MSG = 1024

    def set_body_state(self, name, pose=None, twist=None, mass=None, reset_qpos=False, pose_frame="", twist_frame="",
                       set_pose=None, set_twist=None, admin_hash="", env_lo=0, env_hi=-1):
        st = BodyStateC()
        st.name = name.encode()
        st.pose_frame, st.twist_frame = pose_frame.encode(), twist_frame.encode()
        for k, v in enumerate(pose if pose is not None else [0, 0, 0, 0, 0, 0, 0]):
            st.pose[k] = v
        for k, v in enumerate(twist if twist is not None else [0] * 6):
            st.twist[k] = v
        st.mass = 0.0 if mass is None else mass
        msg = C.create_string_buffer(self._MSG)
        ok = self.L.mjr_env_set_body_state(self.ptr, C.byref(st), int(pose is not None if set_pose is None else set_pose),
                                           int(twist is not None if set_twist is None else set_twist), int(mass is not None),
                                           int(reset_qpos), admin_hash.encode(), env_lo, env_hi, msg, self._MSG)
        return bool(ok == 1), msg.value.decode()

    def get_body_state(self, name, admin_hash="", env=0):
        st = BodyStateC()
        msg = C.create_string_buffer(self._MSG)
        ok = self.L.mjr_env_get_body_state(self.ptr, name.encode(), admin_hash.encode(), env, C.byref(st), msg, self._MSG)
        return bool(ok == 1), msg.value.decode(), dict(name=st.name.decode(), mass=st.mass, pose=np.array(st.pose[:]),
                                                      twist=np.array(st.twist[:]), pose_frame=st.pose_frame.decode(),
                                                      twist_frame=st.twist_frame.decode())

    def set_geom_properties(self, name, type=None, body_mass=None, friction=None, size=None, admin_hash="", env_lo=0, env_hi=-1):
        p = GeomPropertiesC()
        p.name = name.encode()
        p.type = 0 if type is None else int(type)
        p.body_mass = 0.0 if body_mass is None else body_mass
        for k in range(3):
            p.friction[k] = 0.0 if friction is None else friction[k]
            p.size[k] = 0.0 if size is None else size[k]
        msg = C.create_string_buffer(self._MSG)
        ok = self.L.mjr_env_set_geom_properties(self.ptr, C.byref(p), int(type is not None), int(body_mass is not None),
                                                int(friction is not None), int(size is not None), admin_hash.encode(), env_lo, env_hi,
                                                msg, self._MSG)
        return bool(ok == 1), msg.value.decode()

    def get_geom_properties(self, name, admin_hash="", env=0):
        p = GeomPropertiesC()
        msg = C.create_string_buffer(self._MSG)
        ok = self.L.mjr_env_get_geom_properties(self.ptr, name.encode(), admin_hash.encode(), env, C.byref(p), msg, self._MSG)
        return bool(ok == 1), msg.value.decode(), dict(name=p.name.decode(), type=p.type, body_mass=p.body_mass,
                                                      friction=np.array(p.friction[:]), size=np.array(p.size[:]))

    def set_gravity(self, gravity, admin_hash="", env_lo=0, env_hi=-1):
        g = (C.c_double * 3)(*[float(x) for x in gravity])
        msg = C.create_string_buffer(self._MSG)
        ok = self.L.mjr_env_set_gravity(self.ptr, g, admin_hash.encode(), env_lo, env_hi, msg, self._MSG)
        return bool(ok == 1), msg.value.decode()

    def get_gravity(self, admin_hash="", env=0):
        g = (C.c_double * 3)()
        msg = C.create_string_buffer(self._MSG)
        ok = self.L.mjr_env_get_gravity(self.ptr, admin_hash.encode(), env, g, msg, self._MSG)
        return bool(ok == 1), msg.value.decode(), np.array(g[:])

    @staticmethod
    def _eq_to_c(d):
        p = EqParametersC()
        p.name, p.element1, p.element2 = d["name"].encode(), d.get("element1", "").encode(), d.get("element2", "").encode()
        p.type, p.active = int(d.get("type", 0)), int(bool(d.get("active", False)))
        for key, n in (("anchor", 3), ("relpose", 7), ("polycoef", 5)):
            for k, v in enumerate(d.get(key, [0] * n)):
                getattr(p, key)[k] = v
        p.torquescale = d.get("torquescale", 0.0)
        for key in ("dmin", "dmax", "width", "midpoint", "power", "timeconst", "dampratio"):
            setattr(p, key, float(d.get(key, 0.0)))
        return p

    def set_eq_parameters(self, params, admin_hash="", env_lo=0, env_hi=-1):
        arr = (EqParametersC * max(1, len(params)))(*[self._eq_to_c(d) for d in params])
        msg = C.create_string_buffer(self._MSG)
        ok = self.L.mjr_env_set_eq_parameters(self.ptr, arr, len(params), admin_hash.encode(), env_lo, env_hi, msg, self._MSG)
        return bool(ok == 1), msg.value.decode()

    def get_eq_parameters(self, names, admin_hash="", env=0):
        na = (C.c_char_p * max(1, len(names)))(*[n.encode() for n in names])
        out = (EqParametersC * max(1, len(names)))()
        nout = C.c_int(0)
        msg = C.create_string_buffer(self._MSG)
        ok = self.L.mjr_env_get_eq_parameters(self.ptr, na, len(names), admin_hash.encode(), env, out, C.byref(nout), msg, self._MSG)
        res = []
        for k in range(nout.value):
            p = out[k]
            res.append(dict(name=p.name.decode(), element1=p.element1.decode(), element2=p.element2.decode(), type=p.type,
                            active=bool(p.active), anchor=np.array(p.anchor[:]), relpose=np.array(p.relpose[:]),
                            torquescale=p.torquescale, polycoef=np.array(p.polycoef[:]), dmin=p.dmin, dmax=p.dmax, width=p.width,
                            midpoint=p.midpoint, power=p.power, timeconst=p.timeconst, dampratio=p.dampratio))
        return bool(ok == 1), msg.value.decode(), res

    def reload(self, model, nenv=1, device=0, backend_factory=None):
        """reloadCB: blocks until the load request has been served; (success, status_message = load_error_)."""
        msg = C.create_string_buffer(self._MSG)
        fac = C.cast(backend_factory, C.c_void_p) if backend_factory is not None else None
        if model is None:
            ok = self.L.mjr_env_reload(self.ptr, None, None, nenv, device, fac, None, msg, self._MSG)
        else:
            desc, names, keep = self._desc_and_names(model)
            self._keep += [desc, keep, names]
            ok = self.L.mjr_env_reload(self.ptr, C.byref(desc), C.byref(names), nenv, device, fac, None, msg, self._MSG)
            if ok == 1:
                self.model = model
        return bool(ok == 1), msg.value.decode()

    def loading_request_state(self):
        buf = C.create_string_buffer(64)
        v = self.L.mjr_env_loading_request_state(self.ptr, buf, 64)
        return v, buf.value.decode()

    def load_initial_joint_states(self):
        return self.L.mjr_env_load_initial_joint_states(self.ptr) == 1

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
