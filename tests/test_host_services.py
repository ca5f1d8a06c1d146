"""The reference's model- and body-state service tests replayed on the batched host runtime (host/services.cpp):
/root/reference mujoco_ros/test/ros_interface_test.cpp:146-184 (reload), :426-539 (SetBodyStateCallback), :541-589
(GetBodyStateCallback), :591-695 (SetGeomPropertiesCallback), :697-745 (GetGeomPropertiesCallback), :954-1067 (SetEqConstraint),
:1069-1095 (GetEqConstraint) -- same requests, same expected `success`, same values afterwards.  The per-env dimension the
reference does not have is exercised on top: a request addressed to an env range leaves the other envs alone.

Backends: "oracle" (CPU harness: handler logic, gates, mirrors) and "hip" (the product: the overrides reach the device and
change the physics; `-m gpu`)."""
import os

import numpy as np
import pytest

from mujoco_ros_pkgs_amd import mjcf
from test_host_env import GOLDEN, factory, host, oracle_factory, pendulum, start, wait  # noqa: F401  (fixtures)

GEOM_BOX, GEOM_CYLINDER, GEOM_ELLIPSOID, GEOM_CAPSULE, GEOM_SPHERE = 6, 5, 4, 3, 2


def equality_world():
    return mjcf.compile_xml_file(os.path.join(GOLDEN, "equality_world.xml"))


def test_set_body_state_callback(host, factory):
    """ros_interface_test.cpp:426-539"""
    m = pendulum()
    env = start(host, factory, m, {"unpause": False})
    assert env.setting("run") == 0 and abs(env.get_field("time")[0]) < 1e-6
    id_free = m["names"]["joint"].index("ball_freejoint")
    qa, da = int(m["jnt_qposadr"][id_free]), int(m["jnt_dofadr"][id_free])
    # invalid body name
    ok, msg = env.set_body_state("unknown")
    assert not ok and "Could not find model (not body nor geom) with name unknown" in msg
    # resolve body; resolve body from child geom
    assert env.set_body_state("middle_link")[0]
    assert env.set_body_state("EE")[0]
    # position change errors: not a freejoint / no joint / unknown frame_id
    ok, msg = env.set_body_state("EE", pose=[0] * 7)
    assert not ok and "has no joint of type 'freetype'" in msg
    ok, msg = env.set_body_state("immovable", pose=[0] * 7)
    assert not ok and "Body has no joints, cannot move body!" in msg
    ok, msg = env.set_body_state("ball", pose=[0] * 7, pose_frame="unknown")
    assert not ok and "Could not transform frame 'unknown' to frame world" in msg
    # twist change errors: other frame_id than world
    ok, msg = env.set_body_state("ball", twist=[0] * 6, twist_frame="not-world", pose_frame="unknown")
    assert not ok and "Transforming twists from other frames is not supported" in msg
    # new twist and pose (quaternion given as x .707, z .707 -> normalised)
    ok, msg = env.set_body_state("ball", pose=[2, 2, 2, 0.0, 0.707, 0.0, 0.707], twist=[0.1, 0.1, -0.1, 0.1, 0, 0], pose_frame="world",
                                 twist_frame="world")
    assert ok, msg
    q, v = env.get_field("qpos"), env.get_field("qvel")
    assert np.allclose(q[qa:qa + 7], [2.0, 2.0, 2.0, 0.0, 0.707, 0.0, 0.707], atol=9e-4) and np.allclose(q[qa:qa + 3], 2.0, atol=0)
    assert abs(np.linalg.norm(q[qa + 3:qa + 7]) - 1) < 1e-12  # mju_normalize4
    assert np.array_equal(v[da:da + 6], [0.1, 0.1, -0.1, 0.1, 0.0, 0.0])
    # new mass
    body_ball = m["names"]["body"].index("body_ball")
    mass = float(np.float32(0.299))
    assert m["body_mass"][body_ball] != mass
    assert env.set_body_state("ball", mass=mass)[0]
    ok, _, st = env.get_body_state("body_ball")
    assert ok and st["mass"] == mass
    # reset
    ok, msg = env.set_body_state("ball", reset_qpos=True)
    assert ok, msg
    q, v = env.get_field("qpos"), env.get_field("qvel")
    assert np.array_equal(q[qa:qa + 7], [1.0, 0.0, 0.06, 1.0, 0.0, 0.0, 0.0]) and np.array_equal(v[da:da + 6], np.zeros(6))
    env.shutdown()


def test_get_body_state_callback(host, factory):
    """ros_interface_test.cpp:541-589, plus the TODO it leaves open: bodies without a free joint report the Cartesian pose"""
    m = pendulum()
    env = start(host, factory, m, {"unpause": False})
    mass = float(np.float32(0.299))
    pose, twist = [1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.0], [0.1, 0.1, -0.1, 0.1, 0.1, -0.1]
    assert env.set_body_state("body_ball", pose=pose, twist=twist, mass=mass)[0]
    ok, msg, _ = env.get_body_state("unknown")
    assert not ok and "Could not find model" in msg
    ok, msg, st = env.get_body_state("body_ball")
    assert ok and st["mass"] == mass and st["name"] == "body_ball"
    assert np.array_equal(st["pose"], pose) and np.array_equal(st["twist"], twist) and st["pose_frame"] == "world"
    # non-free-joint body: xpos / xquat / cvel of the body (not settable, but readable)
    ok, msg, st = env.get_body_state("immovable")
    assert ok and abs(np.linalg.norm(st["pose"][3:]) - 1) < 1e-9
    bid = m["names"]["body"].index("immovable")
    assert np.allclose(st["pose"][:3], np.asarray(m["body_pos"]).reshape(-1, 3)[bid], atol=1e-12) and np.all(st["twist"] == 0)
    env.shutdown()


def test_set_geom_properties_callback(host, factory):
    """ros_interface_test.cpp:591-695"""
    m = pendulum()
    env = start(host, factory, m, {"unpause": False})
    ball = m["names"]["geom"].index("ball")
    body = m["names"]["body"].index("body_ball")
    ok, msg = env.set_geom_properties("unknown")
    assert not ok and "Could not find model (mujoco geom) with name unknown" in msg
    assert env.set_geom_properties("ball")[0]
    mass = float(np.float32(0.299))
    assert m["body_mass"][body] != mass
    assert env.set_geom_properties("ball", body_mass=mass)[0]
    assert env.get_geom_properties("ball")[2]["body_mass"] == mass
    fr = np.asarray(m["geom_friction"]).reshape(-1, 3)[ball]
    assert np.all(fr != 0)
    assert env.set_geom_properties("ball", friction=[0, 0, 0])[0]
    assert np.all(env.get_geom_properties("ball")[2]["friction"] == 0)
    assert m["geom_type"][ball] != GEOM_BOX
    for t in (GEOM_BOX, GEOM_CYLINDER, GEOM_ELLIPSOID, GEOM_CAPSULE, GEOM_SPHERE):
        ok, msg = env.set_geom_properties("ball", type=t)
        assert ok, msg
        assert env.get_geom_properties("ball")[2]["type"] == t
        if t in (GEOM_CYLINDER, GEOM_ELLIPSOID):
            assert "no pair function" in msg  # stated, not silent: these types do not collide in the engine
    sz = float(np.float32(0.01))
    assert np.all(np.asarray(m["geom_size"]).reshape(-1, 3)[ball] != 0.01)
    assert env.set_geom_properties("ball", size=[sz, sz, sz])[0]
    assert np.allclose(env.get_geom_properties("ball")[2]["size"], 0.01, atol=9e-4)
    assert env.step(3)  # the changed model still steps
    assert np.all(np.isfinite(env.get_field("qpos")))
    env.shutdown()


def test_get_geom_properties_callback(host, factory):
    """ros_interface_test.cpp:697-745"""
    env = start(host, factory, pendulum(), {"unpause": False})
    f32 = lambda x: float(np.float32(x))  # noqa: E731
    req = dict(type=GEOM_BOX, body_mass=f32(0.299), size=[f32(0.01)] * 3, friction=[1.0, 1.0, 1.0])
    assert env.set_geom_properties("ball", **req)[0]
    ok, msg, _ = env.get_geom_properties("unknown")
    assert not ok
    ok, msg, p = env.get_geom_properties("ball")
    assert ok and p["name"] == "ball" and p["type"] == req["type"] and p["body_mass"] == req["body_mass"]
    assert np.array_equal(p["size"], req["size"]) and np.array_equal(p["friction"], req["friction"])
    env.shutdown()


def test_gravity_services(host, factory):
    """callbacks.cpp:462-506"""
    m = pendulum()
    env = start(host, factory, m, {"unpause": False}, nenv=3)
    ok, _, g = env.get_gravity()
    assert ok and np.array_equal(g, np.asarray(m["gravity"]))
    assert env.set_gravity([0, 0, -1.62], env_lo=1, env_hi=2)[0]
    assert np.array_equal(env.get_gravity(env=1)[2], [0, 0, -1.62]) and np.array_equal(env.get_gravity(env=0)[2], np.asarray(m["gravity"]))
    assert env.set_gravity([0, 0, -3.7])[0]
    assert all(np.array_equal(env.get_gravity(env=e)[2], [0, 0, -3.7]) for e in range(3))
    env.shutdown()


def _eq_requests():
    quat = np.array([-0.634, -0.002, -0.733, 0.244])
    quat /= np.linalg.norm(quat)
    sol = dict(active=False, dampratio=0.8, timeconst=0.2, dmin=0.7, dmax=0.9, width=0.001, midpoint=0.5, power=3.0)
    connect = dict(name="connect_eq", type=0, element1="immovable", element2="", anchor=[5.0, 3.0, 7.0], **sol)
    weld = dict(name="weld_eq", type=1, element1="", anchor=[5.0, 3.0, 7.0], relpose=[1.2, 1.3, 1.4] + list(quat), **sol)
    joint = dict(name="joint_eq", type=2, element1="", polycoef=[0.1, 1.0, 0.2, 0.3, 0.4], **sol)
    tendon = dict(name="tendon_eq", type=3, element1="", polycoef=[0.1, 1.0, 0.2, 0.3, 0.4], **sol)
    return [connect, weld, joint, tendon]


def _eq_matches(got, want):
    for k in ("active", "dampratio", "timeconst", "dmin", "dmax", "width", "midpoint", "power"):
        if got[k] != want[k]:
            return False
    if want["type"] in (0, 1) and not np.array_equal(got["anchor"], want["anchor"]):
        return False
    if want["type"] == 1 and not np.array_equal(got["relpose"], want["relpose"]):
        return False
    if want["type"] in (2, 3) and not np.array_equal(got["polycoef"], want["polycoef"]):
        return False
    return got["type"] == want["type"]


def test_set_and_get_eq_constraint(host, factory):
    """ros_interface_test.cpp:954-1067 (SetEqConstraint) and :1069-1095 (GetEqConstraint)"""
    m = equality_world()
    env = start(host, factory, m, {"unpause": False}, nenv=2)
    names = ["weld_eq", "tendon_eq", "joint_eq", "connect_eq"]
    ok, msg, cur = env.get_eq_parameters(names)
    assert ok and len(cur) == 4, msg
    # GetEqConstraint: the values are the model's
    for p in cur:
        q = m["names"]["equality"].index(p["name"])
        assert p["type"] == m["eq_type"][q] and p["active"] == bool(m["eq_active"][q])
        assert np.array_equal([p["timeconst"], p["dampratio"]], np.asarray(m["eq_solref"]).reshape(-1, 2)[q])
        assert np.array_equal([p["dmin"], p["dmax"], p["width"], p["midpoint"], p["power"]], np.asarray(m["eq_solimp"]).reshape(-1, 5)[q])
    joint = [p for p in cur if p["name"] == "joint_eq"][0]
    assert np.array_equal(joint["polycoef"], [0.5, 0.25, 0.76, 0.66, 1.0])            # :771-776
    assert (joint["element1"], joint["element2"]) == ("joint_eq_element1", "joint_eq_element2")  # :777-780
    reqs = _eq_requests()
    by_name = {p["name"]: p for p in cur}
    assert not any(_eq_matches(by_name[r["name"]], r) for r in reqs)  # "Verify values differ"
    ok, msg = env.set_eq_parameters(reqs, env_lo=0, env_hi=1)
    assert ok, msg
    ok, msg, got = env.get_eq_parameters([r["name"] for r in reqs], env=0)
    assert ok and all(_eq_matches(g, r) for g, r in zip(got, reqs))
    ok, msg, other = env.get_eq_parameters([r["name"] for r in reqs], env=1)  # the env outside the range keeps the model's values
    assert ok and not any(_eq_matches(g, r) for g, r in zip(other, reqs))
    # unknown names: partial and total failure messages (:762-778, :881-895)
    ok, msg = env.set_eq_parameters([reqs[0], dict(reqs[1], name="nope")])
    assert not ok and msg == "Not all constraints could be set"
    ok, msg = env.set_eq_parameters([dict(reqs[1], name="nope")])
    assert not ok and msg == "Could not set any constraints"
    ok, msg, res = env.get_eq_parameters(["weld_eq", "nope"])
    assert not ok and msg == "Not all constraints could be fetched" and len(res) == 1
    ok, msg, res = env.get_eq_parameters(["nope"])
    assert not ok and msg == "Could not fetch any constraints" and res == []
    assert env.step(2) and np.all(np.isfinite(env.get_field("qpos", env=0)))
    env.shutdown()


def test_eval_mode_gates_every_service(host, factory):
    """callbacks.cpp:213-223 and its copies: in eval mode a wrong admin hash is refused with the service's own message"""
    env = start(host, factory, pendulum(), {"unpause": False, "eval_mode": True}, admin_hash="right")
    assert env.set_body_state("ball", mass=1.0, admin_hash="wrong") == (False, "Hash mismatch, no permission to set body state!")
    assert env.get_body_state("ball", admin_hash="")[:2] == (False, "Hash mismatch, no permission to get body state!")
    assert env.set_geom_properties("ball", body_mass=1.0, admin_hash="wrong") == (False, "Hash mismatch, no permission to set geom properties!")
    assert env.get_geom_properties("ball", admin_hash="wrong")[:2] == (False, "Hash mismatch, no permission to get geom properties!")
    assert env.set_gravity([0, 0, 0], admin_hash="wrong") == (False, "Hash mismatch, no permission to set gravity!")
    assert env.get_gravity(admin_hash="wrong")[:2] == (False, "Hash mismatch, no permission to get gravity!")
    assert env.set_eq_parameters([], admin_hash="wrong")[0] is False
    assert env.get_eq_parameters([], admin_hash="wrong")[0] is False
    assert env.set_body_state("ball", mass=1.0, admin_hash="right")[0]
    assert env.get_body_state("ball", admin_hash="right")[2]["mass"] == 1.0
    env.shutdown()


def test_reload_and_loading_request_state(host, factory):
    """ros_interface_test.cpp:146-184 (ReloadSameModel / ReloadNewModel) + the get_loading_request_state and
    load_initial_joint_states services (callbacks.cpp:66-87)"""
    m = pendulum()
    env = start(host, factory, m, {"unpause": False})
    assert env.loading_request_state() == (0, "Sim ready")
    ok, msg = env.reload(m, backend_factory=factory)
    assert ok and msg == "" and env.operational_status == 0
    empty = mjcf.compile_xml_file(os.path.join(GOLDEN, "empty_world.xml"), disable=("contact",))
    ok, msg = env.reload(empty, backend_factory=factory)
    assert ok and env.operational_status == 0 and env.model["nq"] == empty["nq"]
    assert env.name2id(1, "body_ball") == -1  # the new model is the one in place
    ok, msg = env.reload(None)
    assert not ok and msg
    # load_initial_joint_states re-applies the configured joint map
    ok, _ = env.reload(m, backend_factory=factory)
    assert ok
    env.set_param("initial_joint_positions/joint_map", {"joint1": "-1.57"})
    assert env.load_initial_joint_states()
    j1 = m["names"]["joint"].index("joint1")
    assert env.get_field("qpos")[m["jnt_qposadr"][j1]] == -1.57
    env.shutdown()


@pytest.mark.gpu
def test_mass_and_gravity_overrides_change_the_physics_per_env(host):
    """hip backend: the services' edits reach the device as per-env overrides -- a heavier / lighter ball under different gravity
    falls differently in the env that was addressed and identically in the one that was not."""
    m = pendulum()
    env = start(host, None, m, {"unpause": False}, nenv=3)
    id_free = m["names"]["joint"].index("ball_freejoint")
    qa = int(m["jnt_qposadr"][id_free])
    for e in range(3):
        assert env.set_body_state("ball", pose=[1, 0, 1.0, 1, 0, 0, 0], twist=[0] * 6, env_lo=e, env_hi=e + 1)[0]
    assert env.set_gravity([0, 0, -1.0], env_lo=1, env_hi=2)[0]
    assert env.set_body_state("ball", mass=5.0, env_lo=2, env_hi=3)[0]
    assert env.step(50)
    z = [env.get_field("qpos", env=e)[qa + 2] for e in range(3)]
    dt = m["timestep"][0]
    t = 50 * dt
    assert abs((1.0 - z[0]) - 0.5 * 9.81 * t * (t + dt)) < 1e-9      # semi-implicit Euler free fall
    assert abs((1.0 - z[1]) - 0.5 * 1.0 * t * (t + dt)) < 1e-9       # the env with the moon-like gravity
    assert abs(z[2] - z[0]) < 1e-12                                   # mass does not change free fall
    env.shutdown()
