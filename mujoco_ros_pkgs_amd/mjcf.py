"""MJCF-subset model compiler: XML -> the mjModel-shaped flat arrays of ``include/mjb_model_fields.def``.

The reference never compiles models itself: it calls MuJoCo's ``mj_loadXML``
(/root/reference mujoco_ros/src/mujoco_env.cpp:836-843) and hands the resulting ``mjModel`` to the
step.  MuJoCo is an un-vendored dependency that is absent here, so this module
restates the part of MuJoCo 2.3.7's model compiler (user_model.cc / user_objects.cc /
engine_setconst.c, [UPSTREAM]) that the hot path's inputs need, for the element subset the shipped
worlds and the BASELINE configs use: bodies, inertial, free/ball/hinge/slide joints, plane / sphere
/ capsule / box geoms (incl. ``fromto``), sites, motor / position / velocity / general actuators with
joint transmission, the sensor subset of ``mjb.h`` and ``<contact><exclude>``.  Anything else raises
``MjcfError`` -- nothing is silently dropped except purely visual elements (asset, visual, light,
camera, material attributes).

Output: ``Model`` -- a dict-like of numpy arrays named exactly as in ``mjModel`` plus name tables.
"""
from __future__ import annotations

import math
import os
import xml.etree.ElementTree as ET

import numpy as np

mjMINVAL = 1e-15

JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = 0, 1, 2, 3
GEOM_PLANE, GEOM_SPHERE, GEOM_CAPSULE, GEOM_BOX = 0, 2, 3, 6
GEOM_TYPES = {"plane": GEOM_PLANE, "sphere": GEOM_SPHERE, "capsule": GEOM_CAPSULE, "box": GEOM_BOX}
JNT_TYPES = {"free": JNT_FREE, "ball": JNT_BALL, "slide": JNT_SLIDE, "hinge": JNT_HINGE}
OBJ = {"body": 1, "xbody": 2, "joint": 3, "geom": 5, "site": 6, "tendon": 17, "actuator": 18}
SENSORS = {  # name -> (mjtSensor, dim, needstage, default objtype)
    "touch": (0, 1, 3, "site"),
    "accelerometer": (1, 3, 3, "site"),
    "velocimeter": (2, 3, 2, "site"),
    "gyro": (3, 3, 2, "site"),
    "force": (4, 3, 3, "site"),
    "torque": (5, 3, 3, "site"),
    "magnetometer": (6, 3, 1, "site"),
    "rangefinder": (7, 1, 1, "site"),
    "tendonpos": (10, 1, 1, "tendon"),
    "tendonvel": (11, 1, 2, "tendon"),
    "jointpos": (8, 1, 1, "joint"),
    "jointvel": (9, 1, 2, "joint"),
    "actuatorpos": (12, 1, 1, "actuator"),
    "actuatorvel": (13, 1, 2, "actuator"),
    "actuatorfrc": (14, 1, 3, "actuator"),
    "ballquat": (15, 4, 1, "joint"),
    "ballangvel": (16, 3, 2, "joint"),
    "jointlimitpos": (17, 1, 1, "joint"),
    "jointlimitvel": (18, 1, 2, "joint"),
    "jointlimitfrc": (19, 1, 3, "joint"),
    "tendonlimitpos": (20, 1, 1, "tendon"),
    "tendonlimitvel": (21, 1, 2, "tendon"),
    "tendonlimitfrc": (22, 1, 3, "tendon"),
    "framepos": (23, 3, 1, None),
    "framequat": (24, 4, 1, None),
    "framexaxis": (25, 3, 1, None),
    "frameyaxis": (26, 3, 1, None),
    "framezaxis": (27, 3, 1, None),
    "framelinvel": (28, 3, 2, None),
    "frameangvel": (29, 3, 2, None),
    "framelinacc": (30, 3, 3, None),
    "frameangacc": (31, 3, 3, None),
    "subtreecom": (32, 3, 1, "body"),
    "subtreelinvel": (33, 3, 2, "body"),
    "subtreeangmom": (34, 3, 2, "body"),
    "jointactuatorfrc": (38, 1, 3, "joint"),  # (engine-side value: see include/mjb.h MJB_SENS_JOINTACTFRC)
    "clock": (35, 1, 1, None),
}
DISABLE_BITS = {
    "constraint": 1 << 0, "equality": 1 << 1, "frictionloss": 1 << 2, "limit": 1 << 3,
    "contact": 1 << 4, "passive": 1 << 5, "gravity": 1 << 6, "clampctrl": 1 << 7,
    "warmstart": 1 << 8, "filterparent": 1 << 9, "actuation": 1 << 10, "refsafe": 1 << 11,
    "sensor": 1 << 12, "eulerdamp": 1 << 14,
}
IGNORED_TOP = {"asset", "visual", "statistic", "size", "custom", "keyframe", "extension"}


class MjcfError(ValueError):
    pass


# --------------------------------------------------------------------------- small math helpers
def _floats(s, n=None, what=""):
    v = np.array([float(x) for x in s.split()], dtype=np.float64)
    if n is not None and v.size != n:
        raise MjcfError(f"{what}: expected {n} numbers, got '{s}'")
    return v


def quat_mul(a, b):
    return np.array([
        a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
        a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
        a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
        a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


def quat2mat(q):
    q = np.asarray(q, dtype=np.float64)
    q00, q11, q22, q33 = q[0] * q[0], q[1] * q[1], q[2] * q[2], q[3] * q[3]
    q01, q02, q03 = q[0] * q[1], q[0] * q[2], q[0] * q[3]
    q12, q13, q23 = q[1] * q[2], q[1] * q[3], q[2] * q[3]
    return np.array([[q00 + q11 - q22 - q33, 2 * (q12 - q03), 2 * (q13 + q02)],
                     [2 * (q12 + q03), q00 - q11 + q22 - q33, 2 * (q23 - q01)],
                     [2 * (q13 - q02), 2 * (q23 + q01), q00 - q11 - q22 + q33]])


def mat2quat(R):
    """Rotation matrix -> unit quaternion (w>=0)."""
    R = np.asarray(R, dtype=np.float64)
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    if tr > 0:
        s = math.sqrt(tr + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = math.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, 0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s])
    elif R[1, 1] > R[2, 2]:
        s = math.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = np.array([(R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s])
    else:
        s = math.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = np.array([(R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s])
    if q[0] < 0:
        q = -q
    return q / np.linalg.norm(q)


def z2quat(vec):
    """Quaternion rotating the z axis onto ``vec`` (MuJoCo's mjuu_z2quat, used for ``fromto``)."""
    v = np.asarray(vec, dtype=np.float64)
    v = v / np.linalg.norm(v)
    axis = np.cross([0.0, 0.0, 1.0], v)
    s = np.linalg.norm(axis)
    if s < 1e-10:
        # parallel or anti-parallel
        return np.array([1.0, 0, 0, 0]) if v[2] > 0 else np.array([0.0, 1.0, 0, 0])
    axis = axis / s
    ang = math.atan2(s, v[2])
    return np.concatenate([[math.cos(ang / 2)], axis * math.sin(ang / 2)])


def euler2quat(e, seq="xyz"):
    q = np.array([1.0, 0, 0, 0])
    for i, ch in enumerate(seq):
        ax = "xyz".index(ch.lower())
        r = np.zeros(4)
        r[0] = math.cos(e[i] / 2)
        r[1 + ax] = math.sin(e[i] / 2)
        # lower case: intrinsic (rotating frame) -> post-multiply; upper case: extrinsic
        q = quat_mul(q, r) if ch.islower() else quat_mul(r, q)
    return q


# --------------------------------------------------------------------------- defaults handling
class _Defaults:
    """MJCF <default> classes: per element-tag attribute dicts with inheritance."""

    def __init__(self):
        self.classes = {"main": {}}
        self.parent = {"main": None}

    def load(self, node, cls="main"):
        if cls not in self.classes:
            self.classes[cls] = {}
        for ch in node:
            if ch.tag == "default":
                name = ch.get("class")
                if name is None:
                    raise MjcfError("nested <default> needs a class")
                self.parent[name] = cls
                self.load(ch, name)
            else:
                self.classes[cls].setdefault(ch.tag, {}).update(ch.attrib)

    def resolve(self, tag, cls):
        chain = []
        c = cls if cls in self.classes else "main"
        while c is not None:
            chain.append(c)
            c = self.parent.get(c)
        out = {}
        for c in reversed(chain):
            out.update(self.classes[c].get(tag, {}))
        return out


class Model(dict):
    """mjModel-shaped dict of numpy arrays (+ ``names`` tables). Attribute access allowed."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def name2id(self, kind, name):
        try:
            return self["names"][kind].index(name)
        except ValueError:
            return -1


# --------------------------------------------------------------------------- the compiler
class _Compiler:
    def __init__(self, root, nconmax=None, nefcmax=None, disable=(), override=None, skip_unsupported_pairs=False):
        self.root = root
        self.extra_disable = tuple(disable)
        self.override = dict(override or {})
        self.skip_unsupported_pairs = skip_unsupported_pairs
        self.equalities = []
        self.tendons = []
        self.skipped_pairs = []
        self.notes = []  # compile notes (capacity requests clamped, ...): model["compile_notes"]
        self.nconmax_req = nconmax
        self.nefcmax_req = nefcmax
        self.angle_scale = math.pi / 180.0  # MJCF default: degrees
        self.eulerseq = "xyz"
        self.inertiafromgeom = "auto"
        self.autolimits = False
        self.defaults = _Defaults()
        self.bodies = []  # dicts
        self.joints = []
        self.geoms = []
        self.sites = []
        self.actuators = []
        self.sensors = []
        self.excludes = []
        self.pairs = []      # <contact><pair>: merged attribute dicts
        self.materials = {}   # <asset><material name rgba>: only the colour's alpha matters here (mj_ray skips invisible geoms)
        self.opt = dict(magnetic=np.array([0.0, -0.5, 0.0]), timestep=0.002, gravity=np.array([0, 0, -9.81]), tolerance=1e-8, impratio=1.0,
                        integrator=0, cone=0, solver=2, iterations=100, disableflags=0, enableflags=0)

    # ---- attribute helpers
    def _merged(self, node, childclass):
        cls = node.get("class", childclass or "main")
        # <freejoint> takes no defaults (MJCF reference: "does not use defaults")
        a = {} if node.tag == "freejoint" else dict(self.defaults.resolve(node.tag, cls))
        a.update(node.attrib)
        return a

    def _orientation(self, a, what):
        given = [k for k in ("quat", "euler", "axisangle", "xyaxes", "zaxis") if k in a]
        if len(given) > 1:
            raise MjcfError(f"{what}: multiple orientation specifiers {given}")
        if not given:
            return np.array([1.0, 0, 0, 0])
        k = given[0]
        if k == "quat":
            q = _floats(a["quat"], 4, what)
            n = np.linalg.norm(q)
            if n < mjMINVAL:
                raise MjcfError(f"{what}: zero quaternion")
            return q / n
        if k == "euler":
            return euler2quat(_floats(a["euler"], 3, what) * self.angle_scale, self.eulerseq)
        if k == "axisangle":
            v = _floats(a["axisangle"], 4, what)
            ax = v[:3] / np.linalg.norm(v[:3])
            ang = v[3] * self.angle_scale
            return np.concatenate([[math.cos(ang / 2)], ax * math.sin(ang / 2)])
        if k == "zaxis":
            return z2quat(_floats(a["zaxis"], 3, what))
        v = _floats(a["xyaxes"], 6, what)
        x = v[:3] / np.linalg.norm(v[:3])
        y = v[3:] - x * np.dot(x, v[3:])
        y = y / np.linalg.norm(y)
        return mat2quat(np.stack([x, y, np.cross(x, y)], axis=1))

    # ---- top level
    def compile(self):
        root = self.root
        if root.tag != "mujoco":
            raise MjcfError("root element must be <mujoco>")
        for node in root:
            if node.tag == "compiler":
                self._compiler(node)
        for node in root:
            if node.tag == "default":
                self.defaults.load(node)
        for node in root:
            t = node.tag
            if t == "size":
                # <size nconmax= njmax=>: MuJoCo's per-mjData capacities = the engine's per-env nconmax / nefcmax (explicit
                # arguments of the compile call win); -1 (MuJoCo's "arena") leaves the engine's worst-case sizing on
                if self.nconmax_req is None and int(node.get("nconmax", "-1")) >= 0:
                    self.nconmax_req = int(node.get("nconmax"))
                if self.nefcmax_req is None and int(node.get("njmax", "-1")) >= 0:
                    self.nefcmax_req = int(node.get("njmax"))
                continue
            if t == "asset":
                for mat in node:
                    if mat.tag == "material" and mat.get("name"):
                        self.materials[mat.get("name")] = _floats(mat.get("rgba", "1 1 1 1"), 4, "material rgba")
                continue
            if t in ("compiler", "default") or t in IGNORED_TOP:
                continue
            if t == "option":
                self._option(node)
            elif t == "worldbody":
                self._body(node, parent=-1, childclass=None, is_world=True)
            elif t == "actuator":
                for a in node:
                    self._actuator(a)
            elif t == "sensor":
                for s in node:
                    self._sensor(s)
            elif t == "equality":
                for c in node:
                    self._equality(c)
            elif t == "tendon":
                for c in node:
                    self._tendon(c)
            elif t == "contact":
                for c in node:
                    if c.tag == "exclude":
                        self.excludes.append((c.get("body1"), c.get("body2")))
                    elif c.tag == "pair":
                        self.pairs.append(self._merged(c, None))
                    else:
                        raise MjcfError(f"<contact><{c.tag}> is not supported")
            else:
                raise MjcfError(f"unsupported top-level element <{t}>")
        if not self.bodies:
            self._body(ET.Element("worldbody"), parent=-1, childclass=None, is_world=True)
        for k in self.extra_disable:
            self.opt["disableflags"] |= DISABLE_BITS[k]
        for k, v in self.override.items():  # e.g. {"cone": "pyramidal", "solver": "PGS"}
            if k == "cone":
                self.opt["cone"] = {"pyramidal": 0, "elliptic": 1}[v]
            elif k == "solver":
                self.opt["solver"] = {"PGS": 0, "CG": 1, "Newton": 2}[v]
            elif k == "integrator":
                if v not in ("Euler", "RK4"):
                    raise MjcfError("only the Euler and RK4 integrators are implemented")
                self.opt["integrator"] = {"Euler": 0, "RK4": 1}[v]
            elif k in ("timestep", "tolerance", "impratio", "iterations"):
                self.opt[k] = v
            else:
                raise MjcfError(f"cannot override option '{k}'")
        m = self._finalize()
        m["skipped_collision_pairs"] = list(self.skipped_pairs)
        m["compile_notes"] = list(self.notes)
        return m

    def _compiler(self, node):
        ang = node.get("angle", "degree")
        if ang not in ("degree", "radian"):
            raise MjcfError("compiler angle must be degree or radian")
        self.angle_scale = 1.0 if ang == "radian" else math.pi / 180.0
        self.eulerseq = node.get("eulerseq", "xyz")
        self.inertiafromgeom = node.get("inertiafromgeom", "auto")
        self.autolimits = node.get("autolimits", "false") == "true"

    def _option(self, node):
        o = self.opt
        a = node.attrib
        for k in ("timestep", "tolerance", "impratio"):
            if k in a:
                o[k] = float(a[k])
        if "gravity" in a:
            o["gravity"] = _floats(a["gravity"], 3, "option gravity")
        if "magnetic" in a:
            o["magnetic"] = _floats(a["magnetic"], 3, "option magnetic")
        if "iterations" in a:
            o["iterations"] = int(a["iterations"])
        if "integrator" in a:
            if a["integrator"] not in ("Euler", "RK4", "implicitfast"):
                raise MjcfError("integrator: Euler, RK4 and implicitfast are implemented (implicit is refused)")
            o["integrator"] = {"Euler": 0, "RK4": 1, "implicitfast": 3}[a["integrator"]]
        if "cone" in a:
            o["cone"] = {"pyramidal": 0, "elliptic": 1}[a["cone"]]
        if "solver" in a:
            o["solver"] = {"PGS": 0, "CG": 1, "Newton": 2}[a["solver"]]
        # attributes that would change the physics and are not implemented: refuse them instead of dropping them
        for k in ("noslip_iterations", "density", "viscosity"):
            if k in a and float(a[k]) != 0:
                raise MjcfError(f"option {k} != 0 is not supported ({'the noslip post-solver' if k.startswith('noslip') else 'fluid forces'})")
        if "collision" in a:
            if a["collision"] not in ("all", "dynamic", "predefined"):
                raise MjcfError("option collision must be all, predefined or dynamic")
            o["collision"] = a["collision"]   # (mjCOL_ALL / PAIR / DYNAMIC: which of the <contact><pair> list and the filtered geom pairs are candidates)
        for fl in node:
            if fl.tag != "flag":
                raise MjcfError(f"<option><{fl.tag}> not supported")
            for k, v in fl.attrib.items():
                if k in DISABLE_BITS:
                    if v == "disable":
                        o["disableflags"] |= DISABLE_BITS[k]
                elif k == "energy":
                    if v == "enable":
                        o["enableflags"] |= 1 << 1  # mjENBL_ENERGY
                elif v == "enable" and k in ("override", "fwdinv", "sensornoise", "multiccd"):
                    raise MjcfError(f"enable flag {k} not supported")

    # ---- kinematic tree
    def _body(self, node, parent, childclass, is_world=False):
        bid = len(self.bodies)
        if is_world:
            b = dict(name="world", parent=0, pos=np.zeros(3), quat=np.array([1.0, 0, 0, 0]), inertial=None,
                     joints=[], geoms=[], sites=[])
        else:
            a = node.attrib
            b = dict(name=a.get("name", f"body{bid}"), parent=parent,
                     pos=_floats(a.get("pos", "0 0 0"), 3, "body pos"),
                     quat=self._orientation(a, "body"), inertial=None, joints=[], geoms=[], sites=[])
            if float(a.get("gravcomp", 0)) != 0:
                raise MjcfError(f"body '{b['name']}': gravcomp is not supported")
            b["mocap"] = a.get("mocap", "false") == "true"
            if b["mocap"] and parent != 0:
                raise MjcfError(f"mocap body '{b['name']}' must be a child of the world")
            childclass = a.get("childclass", childclass)
        self.bodies.append(b)
        for ch in node:
            t = ch.tag
            if t == "body":
                continue
            if t == "inertial":
                ia = ch.attrib
                I = dict(pos=_floats(ia["pos"], 3, "inertial pos"), quat=self._orientation(ia, "inertial"),
                         mass=float(ia["mass"]))
                if "diaginertia" in ia:
                    I["inertia"] = _floats(ia["diaginertia"], 3, "diaginertia")
                elif "fullinertia" in ia:
                    f = _floats(ia["fullinertia"], 6, "fullinertia")
                    Im = np.array([[f[0], f[3], f[4]], [f[3], f[1], f[5]], [f[4], f[5], f[2]]])
                    I["inertia"], I["quat"] = _principal(Im)
                else:
                    raise MjcfError("inertial needs diaginertia or fullinertia")
                b["inertial"] = I
            elif t in ("joint", "freejoint"):
                if is_world:
                    raise MjcfError("joint in worldbody")
                b["joints"].append(self._joint(ch, bid, childclass))
            elif t == "geom":
                b["geoms"].append(self._geom(ch, bid, childclass))
            elif t == "site":
                b["sites"].append(self._site(ch, bid, childclass))
            elif t in ("light", "camera"):
                pass
            else:
                raise MjcfError(f"unsupported element <{t}> in body '{b['name']}'")
        for ch in node:
            if ch.tag == "body":
                self._body(ch, bid, childclass)

    def _joint(self, node, bid, childclass):
        a = self._merged(node, childclass)
        typ = JNT_FREE if node.tag == "freejoint" else JNT_TYPES[a.get("type", "hinge")]
        j = dict(name=a.get("name", f"joint{len(self.joints)}_{bid}"), type=typ, body=bid)
        j["pos"] = _floats(a.get("pos", "0 0 0"), 3, "joint pos")
        ax = _floats(a.get("axis", "0 0 1"), 3, "joint axis")
        if typ in (JNT_HINGE, JNT_SLIDE):
            n = np.linalg.norm(ax)
            if n < mjMINVAL:
                raise MjcfError("zero joint axis")
            ax = ax / n
        else:
            ax = np.array([0.0, 0, 1.0])
        if typ == JNT_FREE:
            j["pos"] = np.zeros(3)
        j["axis"] = ax
        scale = self.angle_scale if typ in (JNT_HINGE, JNT_BALL) else 1.0
        rng = _floats(a.get("range", "0 0"), 2, "joint range") * scale
        lim = a.get("limited", "auto")
        if lim == "auto":
            limited = self.autolimits and "range" in a
        else:
            limited = lim == "true"
        if limited and typ == JNT_FREE:
            raise MjcfError("free joint cannot be limited")
        j.update(range=rng, limited=int(limited), stiffness=float(a.get("stiffness", 0)),
                 damping=float(a.get("damping", 0)), armature=float(a.get("armature", 0)),
                 margin=float(a.get("margin", 0)),  # (not an angle to the compiler: mjCJoint::Compile converts range / ref / springref only)
                 ref=float(a.get("ref", 0)) * (scale if typ == JNT_HINGE else 1.0),
                 springref=float(a.get("springref", 0)) * (scale if typ == JNT_HINGE else 1.0),
                 solref=_floats(a.get("solreflimit", "0.02 1"), 2, "solreflimit"),
                 solimp=_solimp(a.get("solimplimit")))
        j.update(frictionloss=float(a.get("frictionloss", 0)),
                 solref_fri=_floats(a.get("solreffriction", "0.02 1"), 2, "solreffriction"),
                 solimp_fri=_solimp(a.get("solimpfriction")))
        if j["frictionloss"] < 0:
            raise MjcfError("joint frictionloss must be >= 0")
        if "springdamper" in a and np.any(_floats(a["springdamper"], 2, "springdamper") != 0):
            raise MjcfError("joint springdamper (stiffness / damping from a time constant, needs the compiler's mass properties) is not supported: give stiffness and damping")
        self.joints.append(j)
        return j

    def _geom(self, node, bid, childclass):
        a = self._merged(node, childclass)
        tname = a.get("type", "sphere")
        if tname not in GEOM_TYPES:
            raise MjcfError(f"geom type '{tname}' is not supported (plane/sphere/capsule/box only)")
        typ = GEOM_TYPES[tname]
        g = dict(name=a.get("name", ""), type=typ, body=bid)
        size = _floats(a.get("size", "0 0 0")) if "size" in a else np.zeros(0)
        pos = _floats(a.get("pos", "0 0 0"), 3, "geom pos")
        quat = self._orientation(a, "geom")
        if "fromto" in a:
            if typ != GEOM_CAPSULE and typ != GEOM_BOX:
                raise MjcfError("fromto only supported for capsule/box")
            ft = _floats(a["fromto"], 6, "fromto")
            p0, p1 = ft[:3], ft[3:]
            pos = 0.5 * (p0 + p1)
            half = 0.5 * np.linalg.norm(p1 - p0)
            quat = z2quat(p1 - p0)
            if typ == GEOM_CAPSULE:
                size = np.array([size[0], half, 0.0])
            else:
                size = np.array([size[0], size[0], half])
        s3 = np.zeros(3)
        s3[:min(3, size.size)] = size[:3]
        need = {GEOM_PLANE: 0, GEOM_SPHERE: 1, GEOM_CAPSULE: 2, GEOM_BOX: 3}[typ]
        if typ != GEOM_PLANE and (size.size < need and "fromto" not in a):
            raise MjcfError(f"geom '{g['name']}' of type {tname} needs {need} size values")
        if typ != GEOM_PLANE and np.any(s3[:need] <= 0):
            raise MjcfError(f"geom '{g['name']}': sizes must be positive")
        if typ == GEOM_PLANE and bid != 0 and False:
            pass
        g.update(size=s3, pos=pos, quat=quat,
                 contype=int(a.get("contype", 1)), conaffinity=int(a.get("conaffinity", 1)),
                 condim=int(a.get("condim", 3)), priority=int(a.get("priority", 0)),
                 friction=_pad_friction(a.get("friction")), solmix=float(a.get("solmix", 1)),
                 solref=_floats(a.get("solref", "0.02 1"), 2, "geom solref"), solimp=_solimp(a.get("solimp")),
                 margin=float(a.get("margin", 0)), gap=float(a.get("gap", 0)),
                 density=float(a.get("density", 1000)), mass=(float(a["mass"]) if "mass" in a else None))
        # effective colour: the geom's own rgba when it differs from the default, else its material's (mjCGeom: a material overrides the default rgba)
        rgba = _floats(a.get("rgba", "0.5 0.5 0.5 1"), 4, "geom rgba")
        if "rgba" not in a and a.get("material") in self.materials:
            rgba = self.materials[a.get("material")]
        g["rgba"] = rgba
        if g["condim"] not in (1, 3, 4, 6):
            raise MjcfError("condim must be 1, 3, 4 or 6")
        self.geoms.append(g)
        return g

    def _site(self, node, bid, childclass):
        a = self._merged(node, childclass)
        styp = a.get("type", "sphere")
        if styp not in ("sphere", "box"):
            raise MjcfError(f"site type '{styp}' is not supported (sphere / box)")
        size = np.full(3, 0.005)
        sz = _floats(a.get("size", "0.005"))
        size[:sz.size] = sz
        s = dict(name=a.get("name", f"site{len(self.sites)}"), body=bid, type=GEOM_TYPES[styp], size=size,
                 pos=_floats(a.get("pos", "0 0 0"), 3, "site pos"), quat=self._orientation(a, "site"))
        self.sites.append(s)
        return s

    # ---- actuators / sensors
    def _equality(self, node):
        """<equality><connect|weld|joint>: stored raw; body / joint ids and eq_data are resolved in _finalize (they need
        the kinematics at qpos0).  Tendon and distance equalities are not implemented."""
        if node.tag not in ("connect", "weld", "joint", "tendon"):
            raise MjcfError(f"<equality><{node.tag}> is not supported (connect / weld / joint / tendon only)")
        a = self.defaults.resolve("equality", node.get("class", "main"))
        a.update(node.attrib)
        self.equalities.append((node.tag, a))

    def _tendon(self, node):
        """<tendon><fixed>: length = sum coef_i * qpos[joint_i] (hinge / slide joints).  Spatial tendons are refused."""
        if node.tag != "fixed":
            raise MjcfError(f"<tendon><{node.tag}> is not supported (fixed tendons only)")
        a = self.defaults.resolve("tendon", node.get("class", "main"))
        a.update(node.attrib)
        wraps = []
        for w in node:
            if w.tag != "joint":
                raise MjcfError("fixed tendons take <joint> entries only")
            wraps.append((w.get("joint"), float(w.get("coef", 1))))
        if float(a.get("frictionloss", 0)) < 0:
            raise MjcfError("tendon frictionloss must be >= 0")
        self.tendons.append((a, wraps))

    def _actuator(self, node):
        a = self._merged(node, None)
        t = node.tag
        if "jointinparent" in a and "joint" not in a:
            a = dict(a, joint=a["jointinparent"])   # (identical to joint= for the scalar joints actuators are supported on)
        if ("joint" in a) == ("tendon" in a):
            raise MjcfError(f"actuator <{t}> needs exactly one of joint= / tendon= (joint and fixed-tendon transmissions are supported)")
        act = dict(name=a.get("name", f"actuator{len(self.actuators)}"), joint=a.get("joint"), tendon=a.get("tendon"))
        gear = np.zeros(6)
        gv = _floats(a.get("gear", "1"))
        gear[:gv.size] = gv
        gain = np.zeros(3)
        bias = np.zeros(3)
        gaintype, biastype = 0, 0
        dyntype, dynprm = 0, np.zeros(3)  # mjtDyn: none 0, integrator 1, filter 2 (muscle / user refused)
        if t == "motor":
            gain[0] = 1.0
        elif t == "position":
            kp = float(a.get("kp", 1))
            gain[0] = kp
            bias[1] = -kp
            biastype = 1
            if "kv" in a:
                bias[2] = -float(a["kv"])
        elif t == "velocity":
            kv = float(a.get("kv", 1))
            gain[0] = kv
            bias[2] = -kv
            biastype = 1
        elif t == "intvelocity":
            # integrated-velocity servo (MuJoCo 2.3.0+): the activation is the position set point, act_dot = ctrl; actrange is mandatory
            kp = float(a.get("kp", 1))
            gain[0] = kp
            bias[1] = -kp
            biastype = 1
            dyntype = 1
            if "actrange" not in a:
                raise MjcfError("<intvelocity> needs actrange")
            a = dict(a, actlimited="true")
        elif t == "damper":
            # active damper: force = -kv * velocity * ctrl, ctrl >= 0 (ctrlrange mandatory, lower bound >= 0)
            kv = float(a.get("kv", 1))
            if kv < 0:
                raise MjcfError("<damper> kv must be >= 0")
            gaintype = 1
            gain[2] = -kv
            if "ctrlrange" not in a or _floats(a["ctrlrange"], 2, "ctrlrange")[0] < 0:
                raise MjcfError("<damper> needs ctrlrange with a lower bound >= 0")
            a = dict(a, ctrllimited="true")
        elif t == "cylinder":
            # pneumatic / hydraulic cylinder: first-order filter on ctrl (timeconst), gain = area, bias
            dyntype = 2
            dynprm[0] = float(a.get("timeconst", 1))
            area = float(a.get("area", 1))
            if "diameter" in a:
                area = np.pi / 4 * float(a["diameter"]) ** 2
            gain[0] = area
            bp = _floats(a.get("bias", "0 0 0"))
            bias[:min(3, bp.size)] = bp[:3]
            biastype = 1
        elif t == "general":
            gaintype = {"fixed": 0, "affine": 1}[a.get("gaintype", "fixed")]
            biastype = {"none": 0, "affine": 1}[a.get("biastype", "none")]
            gp = _floats(a.get("gainprm", "1"))
            bp = _floats(a.get("biasprm", "0"))
            gain[:min(3, gp.size)] = gp[:3]
            bias[:min(3, bp.size)] = bp[:3]
            dt = a.get("dyntype", "none")
            if dt not in ("none", "integrator", "filter"):
                raise MjcfError(f"actuator dyntype '{dt}' is not supported (none / integrator / filter)")
            dyntype = {"none": 0, "integrator": 1, "filter": 2}[dt]
            dp = _floats(a.get("dynprm", "1"))
            dynprm[:min(3, dp.size)] = dp[:3]
        else:
            raise MjcfError(f"actuator type <{t}> not supported")
        cr = _floats(a.get("ctrlrange", "0 0"), 2, "ctrlrange")
        fr = _floats(a.get("forcerange", "0 0"), 2, "forcerange")
        ar = _floats(a.get("actrange", "0 0"), 2, "actrange")

        def lim(key, rng_key):
            v = a.get(key, "auto")
            if v == "auto":
                return int(self.autolimits and rng_key in a)
            return int(v == "true")

        act.update(gear=gear, gainprm=gain, biasprm=bias, gaintype=gaintype, biastype=biastype, ctrlrange=cr,
                   forcerange=fr, ctrllimited=lim("ctrllimited", "ctrlrange"),
                   forcelimited=lim("forcelimited", "forcerange"), dyntype=dyntype, dynprm=dynprm, actrange=ar,
                   actlimited=lim("actlimited", "actrange") if dyntype else 0)
        if act["actlimited"] and not ar[0] < ar[1]:
            raise MjcfError(f"actuator '{act['name']}': actlimited needs actrange[0] < actrange[1]")
        self.actuators.append(act)

    def _sensor(self, node):
        t = node.tag
        if t not in SENSORS:
            raise MjcfError(f"sensor <{t}> not supported")
        typ, dim, stage, objkind = SENSORS[t]
        a = node.attrib
        s = dict(name=a.get("name", f"sensor{len(self.sensors)}"), type=typ, dim=dim, stage=stage,
                 cutoff=float(a.get("cutoff", 0)), reftype=None, refname=None)
        if t == "clock":
            s.update(objtype=None, objname=None)
        elif objkind is None:
            s.update(objtype=a["objtype"], objname=a["objname"], reftype=a.get("reftype"), refname=a.get("refname"))
        else:
            key = {"site": "site", "joint": "joint", "actuator": "actuator", "body": "body", "tendon": "tendon"}[objkind]
            s.update(objtype=objkind, objname=a[key])
        self.sensors.append(s)

    # ---- finalisation: flat arrays
    def _finalize(self):
        B, J, Gm, S = self.bodies, self.joints, self.geoms, self.sites
        nbody = len(B)
        m = Model()
        names = dict(body=[b["name"] for b in B], joint=[j["name"] for j in J], geom=[g["name"] for g in Gm],
                     site=[s["name"] for s in S], actuator=[a["name"] for a in self.actuators],
                     sensor=[s["name"] for s in self.sensors],
                     tendon=[a.get("name", f"tendon{i}") for i, (a, _) in enumerate(self.tendons)])
        for kind in ("body", "joint", "site", "actuator"):
            nn = [n for n in names[kind] if n]
            if len(set(nn)) != len(nn):
                raise MjcfError(f"duplicate {kind} names")
        m["names"] = names

        # joints are stored in body order already (depth-first creation == MuJoCo id order)
        # re-index joints grouped by body in body order
        J_sorted = []
        for bi, b in enumerate(B):
            for j in b["joints"]:
                J_sorted.append(j)
        J = J_sorted
        names["joint"] = [j["name"] for j in J]
        G_sorted, S_sorted = [], []
        for b in B:
            G_sorted += b["geoms"]
            S_sorted += b["sites"]
        Gm, S = G_sorted, S_sorted
        names["geom"] = [g["name"] for g in Gm]
        names["site"] = [s["name"] for s in S]
        njnt, ngeom, nsite = len(J), len(Gm), len(S)

        qn = {JNT_FREE: 7, JNT_BALL: 4, JNT_SLIDE: 1, JNT_HINGE: 1}
        vn = {JNT_FREE: 6, JNT_BALL: 3, JNT_SLIDE: 1, JNT_HINGE: 1}
        nq = sum(qn[j["type"]] for j in J)
        nv = sum(vn[j["type"]] for j in J)

        I, D = np.int32, np.float64
        m["body_parentid"] = np.array([b["parent"] for b in B], I)
        mocapid, nmocap = [], 0
        for b in B:
            if b.get("mocap"):
                if b["joints"]:
                    raise MjcfError(f"mocap body '{b['name']}' cannot have joints")
                mocapid.append(nmocap)
                nmocap += 1
            else:
                mocapid.append(-1)
        m["body_mocapid"] = np.array(mocapid, I)
        m["nmocap"] = nmocap
        m["body_pos"] = np.array([b["pos"] for b in B], D).reshape(nbody, 3)
        m["body_quat"] = np.array([b["quat"] for b in B], D).reshape(nbody, 4)
        body_jntnum = np.zeros(nbody, I)
        body_jntadr = -np.ones(nbody, I)
        body_dofnum = np.zeros(nbody, I)
        body_dofadr = -np.ones(nbody, I)
        jnt_type = np.zeros(njnt, I)
        jnt_qposadr = np.zeros(njnt, I)
        jnt_dofadr = np.zeros(njnt, I)
        jnt_bodyid = np.zeros(njnt, I)
        dof_bodyid = np.zeros(nv, I)
        dof_jntid = np.zeros(nv, I)
        dof_parentid = -np.ones(nv, I)
        dof_armature = np.zeros(nv, D)
        dof_damping = np.zeros(nv, D)
        dof_frictionloss = np.zeros(nv, D)
        dof_solref = np.zeros((nv, 2), D)
        dof_solimp = np.zeros((nv, 5), D)
        qpos0 = np.zeros(nq, D)
        qpos_spring = np.zeros(nq, D)
        body_id = {id(b): i for i, b in enumerate(B)}
        last_dof_of_body = -np.ones(nbody, I)  # last dof on the path root->body (inclusive)
        ji = qa = da = 0
        for bi, b in enumerate(B):
            if bi > 0:
                last_dof_of_body[bi] = last_dof_of_body[b["parent"]]
            if b["joints"]:
                body_jntadr[bi] = ji
                body_dofadr[bi] = da
                if any(j["type"] == JNT_FREE for j in b["joints"]) and (len(b["joints"]) > 1 or b["parent"] != 0):
                    raise MjcfError("free joint must be the only joint of a top-level body")
            for j in b["joints"]:
                t = j["type"]
                jnt_type[ji] = t
                jnt_qposadr[ji] = qa
                jnt_dofadr[ji] = da
                jnt_bodyid[ji] = bi
                j["id"] = ji
                if t == JNT_FREE:
                    qpos0[qa:qa + 3] = b["pos"]
                    qpos0[qa + 3:qa + 7] = b["quat"]
                elif t == JNT_BALL:
                    qpos0[qa:qa + 4] = [1, 0, 0, 0]
                else:
                    qpos0[qa] = j["ref"]
                qpos_spring[qa:qa + qn[t]] = qpos0[qa:qa + qn[t]]
                if t in (JNT_HINGE, JNT_SLIDE):
                    qpos_spring[qa] = j["springref"]
                for k in range(vn[t]):
                    dof_bodyid[da] = bi
                    dof_jntid[da] = ji
                    dof_parentid[da] = last_dof_of_body[bi]
                    last_dof_of_body[bi] = da
                    dof_armature[da] = j["armature"]
                    dof_damping[da] = j["damping"]
                    dof_frictionloss[da] = j["frictionloss"]
                    dof_solref[da] = j["solref_fri"]
                    dof_solimp[da] = j["solimp_fri"]
                    da += 1
                qa += qn[t]
                ji += 1
            body_jntnum[bi] = len(b["joints"])
            body_dofnum[bi] = sum(vn[j["type"]] for j in b["joints"])
        dof_Madr = np.zeros(nv, I)
        nM = 0
        for i in range(nv):
            dof_Madr[i] = nM
            k = i
            while k >= 0:
                nM += 1
                k = dof_parentid[k]
        m.update(nq=nq, nv=nv, nbody=nbody, njnt=njnt, ngeom=ngeom, nsite=nsite, nM=nM, na=0)
        m.update(body_jntnum=body_jntnum, body_jntadr=body_jntadr, body_dofnum=body_dofnum, body_dofadr=body_dofadr,
                 jnt_type=jnt_type, jnt_qposadr=jnt_qposadr, jnt_dofadr=jnt_dofadr, jnt_bodyid=jnt_bodyid,
                 dof_bodyid=dof_bodyid, dof_jntid=dof_jntid, dof_parentid=dof_parentid, dof_Madr=dof_Madr,
                 dof_armature=dof_armature, dof_damping=dof_damping, dof_frictionloss=dof_frictionloss, dof_solref=dof_solref,
                 dof_solimp=dof_solimp, qpos0=qpos0, qpos_spring=qpos_spring)
        m["jnt_pos"] = np.array([j["pos"] for j in J], D).reshape(njnt, 3)
        m["jnt_axis"] = np.array([j["axis"] for j in J], D).reshape(njnt, 3)
        m["jnt_limited"] = np.array([j["limited"] for j in J], I)
        m["jnt_stiffness"] = np.array([j["stiffness"] for j in J], D)
        m["jnt_range"] = np.array([j["range"] for j in J], D).reshape(njnt, 2)
        m["jnt_margin"] = np.array([j["margin"] for j in J], D)
        m["jnt_solref"] = np.array([j["solref"] for j in J], D).reshape(njnt, 2)
        m["jnt_solimp"] = np.array([j["solimp"] for j in J], D).reshape(njnt, 5)

        # root / weld ids
        rootid = np.zeros(nbody, I)
        weldid = np.zeros(nbody, I)
        for bi in range(1, nbody):
            p = B[bi]["parent"]
            rootid[bi] = bi if p == 0 else rootid[p]
            weldid[bi] = bi if body_jntnum[bi] > 0 else weldid[p]
        m["body_rootid"] = rootid
        m["body_weldid"] = weldid

        # geoms
        for gi, g in enumerate(Gm):
            g["id"] = gi
        m["geom_type"] = np.array([g["type"] for g in Gm], I)
        m["geom_contype"] = np.array([g["contype"] for g in Gm], I)
        m["geom_conaffinity"] = np.array([g["conaffinity"] for g in Gm], I)
        m["geom_condim"] = np.array([g["condim"] for g in Gm], I)
        m["geom_bodyid"] = np.array([g["body"] for g in Gm], I)
        m["geom_priority"] = np.array([g["priority"] for g in Gm], I)
        m["geom_size"] = np.array([g["size"] for g in Gm], D).reshape(ngeom, 3)
        m["geom_pos"] = np.array([g["pos"] for g in Gm], D).reshape(ngeom, 3)
        m["geom_quat"] = np.array([g["quat"] for g in Gm], D).reshape(ngeom, 4)
        m["geom_friction"] = np.array([g["friction"] for g in Gm], D).reshape(ngeom, 3)
        m["geom_solmix"] = np.array([g["solmix"] for g in Gm], D)
        m["geom_solref"] = np.array([g["solref"] for g in Gm], D).reshape(ngeom, 2)
        m["geom_solimp"] = np.array([g["solimp"] for g in Gm], D).reshape(ngeom, 5)
        m["geom_margin"] = np.array([g["margin"] for g in Gm], D)
        m["geom_rgba"] = np.array([g["rgba"] for g in Gm], D).reshape(ngeom, 4)
        m["geom_gap"] = np.array([g["gap"] for g in Gm], D)
        rb = np.zeros(ngeom, D)
        for gi, g in enumerate(Gm):
            s = g["size"]
            rb[gi] = {GEOM_PLANE: 0.0, GEOM_SPHERE: s[0], GEOM_CAPSULE: s[0] + s[1],
                      GEOM_BOX: float(np.linalg.norm(s))}[g["type"]]
        m["geom_rbound"] = rb
        m["geom_sameframe"] = np.array(
            [int(np.all(g["pos"] == 0) and np.all(g["quat"] == [1, 0, 0, 0])) for g in Gm], I)

        # body inertial properties
        body_mass = np.zeros(nbody, D)
        body_inertia = np.zeros((nbody, 3), D)
        body_ipos = np.zeros((nbody, 3), D)
        body_iquat = np.tile(np.array([1.0, 0, 0, 0]), (nbody, 1))
        for bi, b in enumerate(B):
            use_geoms = self.inertiafromgeom == "true" or (self.inertiafromgeom == "auto" and b["inertial"] is None)
            if b["inertial"] is not None and not use_geoms:
                Iin = b["inertial"]
                body_mass[bi] = Iin["mass"]
                body_inertia[bi] = Iin["inertia"]
                body_ipos[bi] = Iin["pos"]
                body_iquat[bi] = Iin["quat"]
            elif b["geoms"] and bi > 0 and self.inertiafromgeom != "false":
                body_mass[bi], body_ipos[bi], body_iquat[bi], body_inertia[bi] = _inertia_from_geoms(b["geoms"])
        m.update(body_mass=body_mass, body_inertia=body_inertia, body_ipos=body_ipos, body_iquat=body_iquat)
        m["body_sameframe"] = np.array(
            [int(np.all(body_ipos[i] == 0) and np.all(body_iquat[i] == [1, 0, 0, 0])) for i in range(nbody)], I)
        sub = body_mass.copy()
        for bi in range(nbody - 1, 0, -1):
            sub[B[bi]["parent"]] += sub[bi]
        m["body_subtreemass"] = sub
        for bi in range(1, nbody):
            if body_dofnum[bi] > 0 and sub[bi] < mjMINVAL:
                raise MjcfError(f"mass and inertia of moving body '{B[bi]['name']}' must be larger than mjMINVAL")

        # sites
        m["site_bodyid"] = np.array([s["body"] for s in S], I)
        m["site_pos"] = np.array([s["pos"] for s in S], D).reshape(nsite, 3)
        m["site_quat"] = np.array([s["quat"] for s in S], D).reshape(nsite, 4)
        m["site_type"] = np.array([s["type"] for s in S], I)
        m["site_size"] = np.array([s["size"] for s in S], D).reshape(nsite, 3)
        m["site_sameframe"] = np.array(
            [int(np.all(s["pos"] == 0) and np.all(s["quat"] == [1, 0, 0, 0])) for s in S], I)

        # actuators
        A = self.actuators
        nu = len(A)
        m["nu"] = nu
        trnid = -np.ones((nu, 2), I)
        trntype = np.zeros(nu, I)
        tendon_names = [ta.get("name", f"tendon{k}") for k, (ta, _) in enumerate(self.tendons)]
        for i, a in enumerate(A):
            if a["tendon"] is not None:   # mjTRN_TENDON: length = gear * ten_length, moment = gear * the tendon's coefficients
                if a["tendon"] not in tendon_names:
                    raise MjcfError(f"actuator '{a['name']}': unknown tendon '{a['tendon']}'")
                trnid[i, 0] = tendon_names.index(a["tendon"])
                trntype[i] = 3
                continue
            jid = m.name2id("joint", a["joint"])
            if jid < 0:
                raise MjcfError(f"actuator '{a['name']}': unknown joint '{a['joint']}'")
            if jnt_type[jid] not in (JNT_HINGE, JNT_SLIDE):
                raise MjcfError("actuators are only supported on hinge/slide joints")
            trnid[i, 0] = jid
        m["actuator_trnid"] = trnid
        m["actuator_trntype"] = trntype
        m["actuator_dyntype"] = np.array([a["dyntype"] for a in A], I)
        # one activation variable per stateful actuator, in actuator order (mjModel.actuator_actadr; -1: stateless)
        actadr, na = [], 0
        for a in A:
            actadr.append(na if a["dyntype"] else -1)
            na += 1 if a["dyntype"] else 0
        m["na"] = na
        m["actuator_actadr"] = np.array(actadr, I)
        m["actuator_actlimited"] = np.array([a["actlimited"] for a in A], I)
        m["actuator_dynprm"] = np.array([a["dynprm"] for a in A], D).reshape(nu, 3)
        m["actuator_actrange"] = np.array([a["actrange"] for a in A], D).reshape(nu, 2)
        m["actuator_gaintype"] = np.array([a["gaintype"] for a in A], I)
        m["actuator_biastype"] = np.array([a["biastype"] for a in A], I)
        m["actuator_ctrllimited"] = np.array([a["ctrllimited"] for a in A], I)
        m["actuator_forcelimited"] = np.array([a["forcelimited"] for a in A], I)
        m["actuator_gainprm"] = np.array([a["gainprm"] for a in A], D).reshape(nu, 3)
        m["actuator_biasprm"] = np.array([a["biasprm"] for a in A], D).reshape(nu, 3)
        m["actuator_ctrlrange"] = np.array([a["ctrlrange"] for a in A], D).reshape(nu, 2)
        m["actuator_forcerange"] = np.array([a["forcerange"] for a in A], D).reshape(nu, 2)
        m["actuator_gear"] = np.array([a["gear"] for a in A], D).reshape(nu, 6)

        # sensors
        SS = self.sensors
        ns = len(SS)
        adr = 0
        s_type, s_stage, s_objtype, s_objid = (np.zeros(ns, I) for _ in range(4))
        s_reftype, s_refid, s_dim, s_adr = np.zeros(ns, I), -np.ones(ns, I), np.zeros(ns, I), np.zeros(ns, I)
        for i, s in enumerate(SS):
            s_type[i], s_dim[i], s_stage[i], s_adr[i] = s["type"], s["dim"], s["stage"], adr
            adr += s["dim"]
            if s["objtype"] is None:
                s_objtype[i], s_objid[i] = 0, -1
            else:
                if s["objtype"] not in OBJ:
                    raise MjcfError(f"sensor objtype '{s['objtype']}' not supported")
                s_objtype[i] = OBJ[s["objtype"]]
                kind = "body" if s["objtype"] == "xbody" else s["objtype"]
                s_objid[i] = m.name2id(kind, s["objname"])
                if s_objid[i] < 0:
                    raise MjcfError(f"sensor '{s['name']}': unknown {kind} '{s['objname']}'")
            if s["reftype"] is not None:
                s_reftype[i] = OBJ[s["reftype"]]
                kind = "body" if s["reftype"] == "xbody" else s["reftype"]
                s_refid[i] = m.name2id(kind, s["refname"])
                if s_refid[i] < 0:
                    raise MjcfError(f"sensor '{s['name']}': unknown ref {kind} '{s['refname']}'")
        m.update(nsensor=ns, nsensordata=adr, sensor_type=s_type, sensor_needstage=s_stage, sensor_objtype=s_objtype,
                 sensor_objid=s_objid, sensor_reftype=s_reftype, sensor_refid=s_refid, sensor_dim=s_dim,
                 sensor_adr=s_adr, sensor_cutoff=np.array([s["cutoff"] for s in SS], D))

        # options
        o = self.opt
        m.update(magnetic=np.asarray(o["magnetic"], D), timestep=np.array([o["timestep"]], D), gravity=np.asarray(o["gravity"], D),
                 tolerance=np.array([o["tolerance"]], D), impratio=np.array([o["impratio"]], D),
                 integrator=o["integrator"], cone=o["cone"], solver=o["solver"], iterations=o["iterations"],
                 disableflags=o["disableflags"], enableflags=o["enableflags"])

        # fixed tendons, then equality constraints (mjCEquality::Compile, [UPSTREAM] user_objects.cc): ids + eq_data at qpos0
        self._compile_tendons(m)
        if o["integrator"] == 3:
            # implicitfast: the engine takes a constant, diagonal velocity derivative (include/mjb.h, MJB_INT_IMPLICITFAST)
            if np.any(np.asarray(m["tendon_damping"]) != 0):
                raise MjcfError("integrator implicitfast with tendon damping is not supported")
            if nu and np.any((m["actuator_trntype"] == 3) & (m["actuator_biastype"] == 1) & (m["actuator_biasprm"][:, 2] != 0)):
                raise MjcfError("integrator implicitfast with a velocity-dependent actuator on a tendon is not supported")
            if nu and np.any((m["actuator_gaintype"] == 1) & (m["actuator_gainprm"][:, 2] != 0)):
                raise MjcfError("integrator implicitfast with a velocity term in an affine actuator gain (<damper>) is not supported")
        self._compile_equalities(m)

        # static collision candidates (restates the body/geom filters of MuJoCo's mj_collision)
        pairs = self._collision_pairs(m)
        m["collpair_geom"] = np.array([(a, b) for a, b, _ in pairs], I).reshape(len(pairs), 2)
        m["ncollpair"] = len(pairs)
        expl = np.zeros(len(pairs), I)
        pcond = -np.ones(len(pairs), I)
        pprm = np.full((len(pairs), 14), np.nan)
        for k, (_, _, pa) in enumerate(pairs):
            if pa is None:
                continue
            expl[k] = 1
            if "condim" in pa:
                pcond[k] = int(pa["condim"])
                if pcond[k] not in (1, 3, 4, 6):
                    raise MjcfError("pair condim must be 1, 3, 4 or 6")
            if "friction" in pa:   # up to five numbers; the missing ones take MuJoCo's pair defaults
                fv = _floats(pa["friction"])
                full = np.array([1.0, 1.0, 0.005, 0.0001, 0.0001])
                full[:min(5, fv.size)] = fv[:5]
                if fv.size == 1:
                    full[1] = fv[0]
                pprm[k, 0:5] = full
            if "solref" in pa:
                pprm[k, 5:7] = _floats(pa["solref"], 2, "pair solref")
            if "solimp" in pa:
                pprm[k, 7:12] = _solimp(pa["solimp"])
            if "margin" in pa:
                pprm[k, 12] = float(pa["margin"])
            if "gap" in pa:
                pprm[k, 13] = float(pa["gap"])
        m["collpair_explicit"], m["collpair_condim"], m["collpair_param"] = expl, pcond, pprm
        pairs = [(a, b) for a, b, _ in pairs]
        # capacities: worst case contacts per pair type
        maxcon = 0
        for g1, g2 in pairs:
            t1, t2 = m["geom_type"][g1], m["geom_type"][g2]
            maxcon += _max_contacts(t1, t2)
        # (a request above the worst case can never be reached: a stock MuJoCo file's <size nconmax="100" njmax="500"/> must neither
        #  inflate the LDS frame nor trip the solvers' row caps -- the request only ever LOWERS the capacity.  ADVICE r02.)
        nconmax = maxcon if self.nconmax_req is None else min(int(self.nconmax_req), maxcon)
        nlimit = 0
        for j in range(njnt):
            if m["jnt_limited"][j]:
                r = m["jnt_range"][j]
                if jnt_type[j] == JNT_BALL:
                    # a ball joint's limit is on its rotation angle, against max(range): one row (mj_instantiateLimit)
                    if max(r[0], r[1]) <= 0:
                        raise MjcfError("a limited ball joint needs a positive range[1] (its maximum rotation angle)")
                    nlimit += 1
                    continue
                nlimit += 1 if (r[1] - r[0]) > 2 * m["jnt_margin"][j] else 2
        for t in range(m["ntendon"]):
            if m["tendon_limited"][t]:
                r = m["tendon_range"][t]
                nlimit += 1 if (r[1] - r[0]) > 2 * m["tendon_margin"][t] else 2
        if o["disableflags"] & DISABLE_BITS["limit"]:
            nlimit = 0
        rows_per_con = 0
        if pairs:
            maxdim = max(int(pcond[k]) if pcond[k] > 0 else int(max(m["geom_condim"][g1], m["geom_condim"][g2])) for k, (g1, g2) in enumerate(pairs))
            rows_per_con = 1 if maxdim == 1 else (2 * (maxdim - 1) if o["cone"] == 0 else maxdim)
        neqrow = 0
        if not (o["disableflags"] & DISABLE_BITS["equality"]):
            neqrow = int(sum({0: 3, 1: 6, 2: 1, 3: 1}[int(t)] for t in m["eq_type"]))
        nfric = 0 if (o["disableflags"] & DISABLE_BITS["frictionloss"]) else int(
            np.count_nonzero(m["dof_frictionloss"] > 0) + np.count_nonzero(m["tendon_frictionloss"] > 0))
        worst_rows = neqrow + nfric + nlimit + rows_per_con * nconmax
        nefcmax = worst_rows if self.nefcmax_req is None else min(int(self.nefcmax_req), worst_rows)
        if self.nefcmax_req is not None and int(self.nefcmax_req) > worst_rows:
            self.notes.append(f"<size njmax={int(self.nefcmax_req)}> exceeds the model's worst case ({worst_rows} rows): capacity set to {worst_rows}")
        if self.nconmax_req is not None and int(self.nconmax_req) > maxcon:
            self.notes.append(f"<size nconmax={int(self.nconmax_req)}> exceeds the model's worst case ({maxcon} contacts): capacity set to {maxcon}")
        if o["disableflags"] & (DISABLE_BITS["constraint"]):
            nconmax, nefcmax = 0, 0
        m["nconmax"], m["nefcmax"] = int(nconmax), int(nefcmax)

        # mj_setConst: inverse weights at qpos0 (independent numpy dynamics, see refdyn.py)
        from . import refdyn
        dof_inv, body_inv, meaninertia = refdyn.invweight0(m)
        m["dof_invweight0"] = dof_inv
        m["body_invweight0"] = body_inv
        m["meaninertia"] = np.array([meaninertia], D)
        return m

    def _compile_tendons(self, m):
        from . import refdyn
        I, D = np.int32, np.float64
        T = self.tendons
        nt = len(T)
        adr, num, objid, prm, names = [], [], [], [], []
        lim, rng, margin = np.zeros(nt, I), np.zeros((nt, 2)), np.zeros(nt)
        solref, solimp = np.zeros((nt, 2)), np.zeros((nt, 5))
        stiff, damp, lspring = np.zeros(nt), np.zeros(nt), np.zeros(nt)
        floss, solref_f, solimp_f = np.zeros(nt), np.zeros((nt, 2)), np.zeros((nt, 5))
        for i, (a, wraps) in enumerate(T):
            names.append(a.get("name", f"tendon{i}"))
            adr.append(len(objid))
            num.append(len(wraps))
            for jn, coef in wraps:
                j = m.name2id("joint", jn or "")
                if j < 0 or m["jnt_type"][j] < JNT_SLIDE:
                    raise MjcfError(f"tendon '{names[-1]}': joint '{jn}' must be an existing hinge / slide joint")
                objid.append(j)
                prm.append(coef)
            r = _floats(a.get("range", "0 0"), 2, "tendon range")
            limited = a.get("limited", "auto")
            lim[i] = 1 if (limited == "true" or (limited == "auto" and "range" in a and self.autolimits)) else 0
            rng[i], margin[i] = r, float(a.get("margin", 0))
            solref[i] = _floats(a.get("solreflimit", "0.02 1"), 2, "tendon solreflimit")
            solimp[i] = _solimp(a.get("solimplimit"))
            stiff[i], damp[i] = float(a.get("stiffness", 0)), float(a.get("damping", 0))
            lspring[i] = float(a.get("springlength", -1))
            floss[i] = float(a.get("frictionloss", 0))
            solref_f[i] = _floats(a.get("solreffriction", "0.02 1"), 2, "tendon solreffriction")
            solimp_f[i] = _solimp(a.get("solimpfriction"))
        nv = m["nv"]
        J = np.zeros((nt, nv))
        q0 = np.asarray(m["qpos0"], D)
        len0 = np.zeros(nt)
        for i in range(nt):
            for w in range(adr[i], adr[i] + num[i]):
                j = objid[w]
                J[i, m["jnt_dofadr"][j]] += prm[w]
                len0[i] += prm[w] * q0[m["jnt_qposadr"][j]]
        inv0 = np.zeros(nt)
        if nt and nv:
            Minv = np.linalg.inv(refdyn.mass_matrix(m, q0))
            inv0 = np.einsum("ti,ij,tj->t", J, Minv, J)
        lspring = np.where(lspring < 0, len0, lspring)   # springlength = -1: the length at qpos0
        m.update(ntendon=nt, nwrap=len(objid), tendon_adr=np.array(adr, I), tendon_num=np.array(num, I),
                 tendon_limited=lim, wrap_objid=np.array(objid, I), wrap_prm=np.array(prm, D),
                 tendon_range=rng.reshape(nt, 2), tendon_margin=margin, tendon_solref_lim=solref.reshape(nt, 2),
                 tendon_solimp_lim=solimp.reshape(nt, 5), tendon_length0=len0, tendon_invweight0=inv0,
                 tendon_stiffness=stiff, tendon_damping=damp, tendon_lengthspring=lspring, tendon_frictionloss=floss,
                 tendon_solref_fri=solref_f.reshape(nt, 2), tendon_solimp_fri=solimp_f.reshape(nt, 5))
        m["names"]["tendon"] = names

    def _compile_equalities(self, m):
        from . import refdyn
        I, D = np.int32, np.float64
        E = self.equalities
        n = len(E)
        eq_type, o1, o2, act = np.zeros(n, I), np.zeros(n, I), np.zeros(n, I), np.ones(n, I)
        solref, solimp, data = np.zeros((n, 2)), np.zeros((n, 5)), np.zeros((n, 11))
        names = []
        kin = refdyn.kinematics(m, np.asarray(m["qpos0"], D)) if n else None
        for i, (tag, a) in enumerate(E):
            names.append(a.get("name", ""))
            act[i] = 0 if a.get("active", "true") == "false" else 1
            solref[i] = _floats(a.get("solref", "0.02 1"), 2, "equality solref")
            solimp[i] = _solimp(a.get("solimp"))
            if tag in ("connect", "weld"):
                b1 = m.name2id("body", a.get("body1", ""))
                b2 = m.name2id("body", a["body2"]) if "body2" in a else 0
                if b1 < 0 or b2 < 0:
                    raise MjcfError(f"equality '{names[-1]}': unknown body")
                o1[i], o2[i] = b1, b2
                x1, R1, q1 = kin["xpos"][b1], kin["xmat"][b1].reshape(3, 3), kin["xquat"][b1]
                x2, R2, q2 = kin["xpos"][b2], kin["xmat"][b2].reshape(3, 3), kin["xquat"][b2]
                anchor = _floats(a.get("anchor", "0 0 0"), 3, "equality anchor")
                if tag == "connect":
                    eq_type[i] = 0
                    data[i, 0:3] = anchor                                  # in body1
                    data[i, 3:6] = R2.T @ (x1 + R1 @ anchor - x2)           # the same world point in body2 at qpos0
                else:
                    eq_type[i] = 1
                    rel = _floats(a.get("relpose", "0 1 0 0 0 0 0"), 7, "weld relpose")
                    if np.all(rel[3:] == 0):                               # unspecified: the reference pose of qpos0
                        qc = np.array([q1[0], -q1[1], -q1[2], -q1[3]])
                        relpos, relquat = R1.T @ (x2 - x1), quat_mul(qc, q2)
                    else:
                        relpos, relquat = rel[:3], rel[3:] / np.linalg.norm(rel[3:])
                    data[i, 0:3] = anchor                                  # in body2
                    data[i, 3:6] = relpos + quat2mat(relquat) @ anchor      # the same point in body1
                    data[i, 6:10] = relquat
                    data[i, 10] = float(a.get("torquescale", 1))
            elif tag == "tendon":
                eq_type[i] = 3
                t1 = m["names"]["tendon"].index(a["tendon1"]) if a.get("tendon1") in m["names"]["tendon"] else -1
                t2 = (m["names"]["tendon"].index(a["tendon2"]) if a.get("tendon2") in m["names"]["tendon"] else -2) \
                    if "tendon2" in a else -1
                if t1 < 0 or t2 == -2:
                    raise MjcfError(f"equality '{names[-1]}': unknown tendon")
                o1[i], o2[i] = t1, t2
                pc = _floats(a.get("polycoef", "0 1 0 0 0"))
                data[i, :pc.size] = pc
            else:
                eq_type[i] = 2
                j1 = m.name2id("joint", a.get("joint1", ""))
                j2 = m.name2id("joint", a["joint2"]) if "joint2" in a else -1
                if j1 < 0 or ("joint2" in a and j2 < 0):
                    raise MjcfError(f"equality '{names[-1]}': unknown joint")
                for j in (j1, j2):
                    if j >= 0 and m["jnt_type"][j] < JNT_SLIDE:
                        raise MjcfError("joint equality needs hinge / slide joints")
                o1[i], o2[i] = j1, j2
                pc = _floats(a.get("polycoef", "0 1 0 0 0"))
                data[i, :pc.size] = pc
        m.update(neq=n, eq_type=eq_type, eq_obj1id=o1, eq_obj2id=o2, eq_active=act, eq_solref=solref.reshape(n, 2),
                 eq_solimp=solimp.reshape(n, 5), eq_data=data.reshape(n, 11))
        m["names"]["equality"] = names

    def _collision_pairs(self, m):
        if m["disableflags"] & (DISABLE_BITS["contact"] | DISABLE_BITS["constraint"]):
            return []
        B = self.bodies
        nbody = len(B)
        weld = m["body_weldid"]
        parent = m["body_parentid"]
        excl = set()
        for b1, b2 in self.excludes:
            i1, i2 = m.name2id("body", b1), m.name2id("body", b2)
            if i1 < 0 or i2 < 0:
                raise MjcfError("contact exclude: unknown body")
            excl.add((min(i1, i2), max(i1, i2)))
        filterparent = not (m["disableflags"] & DISABLE_BITS["filterparent"])
        geoms_of = [[g["id"] for g in b["geoms"]] for b in B]
        # <contact><pair>: taken as named -- no contype / conaffinity, parent-child or <exclude> filter (mj_collision tests the predefined
        # pairs ahead of the dynamic ones of the same body pair and skips the dynamic twin of a predefined geom pair)
        explicit = {}
        colmode = self.opt.get("collision", "all")
        for pa in (self.pairs if colmode != "dynamic" else []):
            if "geom1" not in pa or "geom2" not in pa:
                raise MjcfError("<contact><pair> needs geom1 and geom2")
            i1, i2 = m.name2id("geom", pa["geom1"]), m.name2id("geom", pa["geom2"])
            if i1 < 0 or i2 < 0:
                raise MjcfError(f"<contact><pair>: unknown geom '{pa['geom1'] if i1 < 0 else pa['geom2']}'")
            bb1, bb2 = int(m["geom_bodyid"][i1]), int(m["geom_bodyid"][i2])
            if weld[bb1] == weld[bb2]:
                raise MjcfError("<contact><pair>: the two geoms are on the same (welded) body")
            t1, t2 = m["geom_type"][i1], m["geom_type"][i2]
            a, b = (i1, i2) if t1 <= t2 else (i2, i1)
            if _max_contacts(m["geom_type"][a], m["geom_type"][b]) == 0:
                raise MjcfError(f"<contact><pair>: collision between geom types {m['geom_type'][a]} and {m['geom_type'][b]} is not implemented")
            explicit.setdefault((min(bb1, bb2), max(bb1, bb2)), []).append((a, b, pa))
        pairs = []
        for b1 in range(nbody):
            for b2 in range(b1 + 1, nbody):
                mine = explicit.get((b1, b2), [])
                pairs.extend(mine)
                taken = {(a, b) for a, b, _ in mine}
                if colmode == "predefined" or not geoms_of[b1] or not geoms_of[b2]:
                    continue
                if (b1, b2) in excl:
                    continue
                w1, w2 = weld[b1], weld[b2]
                if w1 == w2:
                    continue
                if filterparent and w1 != 0 and w2 != 0:
                    wp1, wp2 = weld[parent[w1]], weld[parent[w2]]
                    if w1 == wp2 or w2 == wp1:
                        continue
                for g1 in geoms_of[b1]:
                    for g2 in geoms_of[b2]:
                        ct1, ca1 = m["geom_contype"][g1], m["geom_conaffinity"][g1]
                        ct2, ca2 = m["geom_contype"][g2], m["geom_conaffinity"][g2]
                        if not ((ct1 & ca2) or (ct2 & ca1)):
                            continue
                        t1, t2 = m["geom_type"][g1], m["geom_type"][g2]
                        a, b = (g1, g2) if t1 <= t2 else (g2, g1)
                        ta, tb = m["geom_type"][a], m["geom_type"][b]
                        if ta == GEOM_PLANE and tb == GEOM_PLANE:
                            continue
                        if _max_contacts(ta, tb) == 0 and self.skip_unsupported_pairs:
                            self.skipped_pairs.append((m["names"]["geom"][a] or str(a), m["names"]["geom"][b] or str(b)))
                            continue
                        if _max_contacts(ta, tb) == 0:
                            raise MjcfError(
                                f"collision between geom types {ta} and {tb} is not implemented; "
                                "filter the pair with contype/conaffinity or <exclude>")
                        if (a, b) in taken:
                            continue
                        pairs.append((a, b, None))
        return pairs


def _max_contacts(t1, t2):
    if t1 > t2:
        t1, t2 = t2, t1
    table = {
        (GEOM_PLANE, GEOM_SPHERE): 1, (GEOM_PLANE, GEOM_CAPSULE): 2, (GEOM_PLANE, GEOM_BOX): 4,
        (GEOM_SPHERE, GEOM_SPHERE): 1, (GEOM_SPHERE, GEOM_CAPSULE): 1, (GEOM_SPHERE, GEOM_BOX): 1,
        (GEOM_CAPSULE, GEOM_CAPSULE): 2, (GEOM_CAPSULE, GEOM_BOX): 2, (GEOM_BOX, GEOM_BOX): 8,
    }
    return table.get((int(t1), int(t2)), 0)


def _solimp(s):
    v = np.array([0.9, 0.95, 0.001, 0.5, 2.0])
    if s is not None:
        x = _floats(s)
        v[:x.size] = x
    return v


def _pad_friction(s):
    v = np.array([1.0, 0.005, 0.0001])
    if s is not None:
        x = _floats(s)
        v[:x.size] = x
    return v


def _principal(Im):
    """Principal axes of a symmetric inertia matrix: (eigenvalues descending, quaternion of the frame)."""
    w, V = np.linalg.eigh(Im)
    order = np.argsort(-w)
    w, V = w[order], V[:, order]
    if np.linalg.det(V) < 0:
        V[:, 2] = -V[:, 2]
    return w, mat2quat(V)


def _geom_mass_inertia(g):
    s = g["size"]
    t = g["type"]
    if t == GEOM_SPHERE:
        vol = 4.0 / 3.0 * math.pi * s[0] ** 3
    elif t == GEOM_CAPSULE:
        vol = math.pi * s[0] ** 2 * (2 * s[1]) + 4.0 / 3.0 * math.pi * s[0] ** 3
    elif t == GEOM_BOX:
        vol = 8.0 * s[0] * s[1] * s[2]
    else:
        return 0.0, np.zeros(3)
    mass = g["mass"] if g["mass"] is not None else g["density"] * vol
    if t == GEOM_SPHERE:
        I = np.full(3, 0.4 * mass * s[0] ** 2)
    elif t == GEOM_BOX:
        I = mass / 3.0 * np.array([s[1] ** 2 + s[2] ** 2, s[0] ** 2 + s[2] ** 2, s[0] ** 2 + s[1] ** 2])
    else:
        r, h = s[0], 2 * s[1]
        ms = mass * 4 * r / (4 * r + 3 * h)  # two hemispheres
        mc = mass - ms
        Ixy = mc * (3 * r * r + h * h) / 12.0 + 0.4 * ms * r * r + ms * h * (3 * r + 2 * h) / 8.0
        I = np.array([Ixy, Ixy, mc * r * r / 2.0 + 0.4 * ms * r * r])
    return mass, I


def _inertia_from_geoms(geoms):
    parts = []
    for g in geoms:
        mass, I = _geom_mass_inertia(g)
        if mass > 0:
            parts.append((mass, I, g["pos"], g["quat"]))
    if not parts:
        return 0.0, np.zeros(3), np.array([1.0, 0, 0, 0]), np.zeros(3)
    if len(parts) == 1:
        mass, I, pos, quat = parts[0]
        return mass, pos.copy(), quat.copy(), I
    M = sum(p[0] for p in parts)
    com = sum(p[0] * p[2] for p in parts) / M
    Im = np.zeros((3, 3))
    for mass, I, pos, quat in parts:
        R = quat2mat(quat)
        d = pos - com
        Im += R @ np.diag(I) @ R.T + mass * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
    w, q = _principal(Im)
    return M, com, q, w


# --------------------------------------------------------------------------- public entry points
def with_body_mass(model, body_mass, body_inertia=None):
    """A copy of a compiled model with new body masses (and optionally principal inertias) and everything mj_setConst
    derives from them: body_subtreemass, dof_invweight0, body_invweight0, tendon_invweight0, meaninertia
    (the reference: setBodyState sets body_mass then calls mj_setConst, mujoco_ros/src/callbacks.cpp:251-256)."""
    from . import refdyn
    m = Model(dict(model))
    nbody, nv = int(m["nbody"]), int(m["nv"])
    mass = np.asarray(body_mass, dtype=np.float64).reshape(nbody).copy()
    m["body_mass"] = mass
    if body_inertia is not None:
        m["body_inertia"] = np.asarray(body_inertia, dtype=np.float64).reshape(nbody, 3).copy()
    sub = mass.copy()
    parent = np.asarray(m["body_parentid"])
    for b in range(nbody - 1, 0, -1):
        sub[parent[b]] += sub[b]
    m["body_subtreemass"] = sub
    dof_inv, body_inv, mean = refdyn.invweight0(m)
    m["dof_invweight0"], m["body_invweight0"], m["meaninertia"] = dof_inv, body_inv, np.array([mean])
    nt = int(m["ntendon"])
    if nt and nv:
        J = np.zeros((nt, nv))
        for t in range(nt):
            for w in range(int(m["tendon_adr"][t]), int(m["tendon_adr"][t]) + int(m["tendon_num"][t])):
                J[t, m["jnt_dofadr"][m["wrap_objid"][w]]] += m["wrap_prm"][w]
        Minv = np.linalg.inv(refdyn.mass_matrix(m, np.asarray(m["qpos0"], dtype=np.float64)))
        m["tendon_invweight0"] = np.einsum("ti,ij,tj->t", J, Minv, J)
    return m


def mass_params(model):
    """The packed per-env block of mjb_set_env_mass_params for a (possibly re-massed) compiled model."""
    return np.concatenate([np.asarray(model[k], dtype=np.float64).reshape(-1) for k in (
        "body_mass", "body_subtreemass", "body_inertia", "dof_invweight0", "body_invweight0", "tendon_invweight0", "meaninertia")])


def compile_xml_string(xml, nconmax=None, nefcmax=None, disable=(), override=None, skip_unsupported_pairs=False):
    """``disable``: extra mjtDisableBit names (e.g. ("contact",)) OR-ed into opt.disableflags.
    ``override``: option overrides, e.g. {"cone": "pyramidal", "solver": "PGS"}.
    ``skip_unsupported_pairs``: drop geom pairs whose narrow phase is not implemented (capsule-box, box-box)
    instead of raising; the dropped pairs are listed in ``model["skipped_collision_pairs"]``."""
    return _Compiler(ET.fromstring(xml), nconmax, nefcmax, disable, override, skip_unsupported_pairs).compile()


def compile_xml_file(path, **kw):
    with open(path, "r") as f:
        return compile_xml_string(f.read(), **kw)


ASSET_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")


def load_asset(name, **kw):
    """Compile one of the models shipped with the package (``assets/<name>.xml``)."""
    return compile_xml_file(os.path.join(ASSET_DIR, name + ".xml"), **kw)
