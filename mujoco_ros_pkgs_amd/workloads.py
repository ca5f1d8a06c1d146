"""Synthetic initial states of the BASELINE workloads (SURVEY.md §8d "Synthetic inputs"), shared by bench.py and
the tests so that both run the same thing."""
import numpy as np


HAND_GRASP = {"WRJ2": 0.0, "WRJ1": 0.0, "LFJ5": 0.2, "THJ5": 0.7, "THJ4": 1.1, "THJ3": 0.0, "THJ2": 0.4, "THJ1": 0.6}


def hand_grasp_angle(joint_name):
    """Half-closed grasp of the Shadow-Hand-like model: fingers folded over the cube lying in the palm."""
    if joint_name in HAND_GRASP:
        return HAND_GRASP[joint_name]
    if joint_name.endswith("J4"):
        return 0.0
    return {"J3": 1.1, "J2": 1.3, "J1": 0.5}[joint_name[-2:]]


# The power grasp of shadow_hand_grasp.xml (tools/gen_hand_model.py grasp): servo targets of the actuated joints and the sprung J1's
# rest angle.  J3's target lies beyond its stop (1.57): the proximal phalanges are driven into the joint limit, i.e. the limit rows
# are active and the servo keeps pressing.
HAND_POWER_GRASP = {"WRJ2": 0.0, "WRJ1": 0.0, "FFJ4": -0.18, "MFJ4": -0.06, "RFJ4": 0.06, "LFJ4": 0.18, "LFJ5": 0.57,
                    "THJ5": 0.9, "THJ4": 0.73, "THJ3": 0.04, "THJ2": 0.7, "THJ1": 1.25}


def hand_power_grasp_angle(joint_name):
    if joint_name in HAND_POWER_GRASP:
        return HAND_POWER_GRASP[joint_name]
    return {"J3": 1.65, "J2": 1.24, "J1": 0.84}[joint_name[-2:]]


def hand_power_grasp_states(model, nenv, seed=0):
    """Config 5 (power grasp): the joints at the servo targets (J3 just short of its stop) + U(-0.05, 0.05), the cube on the palm
    between the thenar pad and the fingers, 2 mm above contact, yaw U(-0.3, 0.3).  The wrist starts FLEXED (WRJ1 0.45 rad) and its servo
    swings the palm level within the first 0.1 s: that flick seats the cube in the closed fingers -- started level, the fingers shove the
    cube back against the thenar pad and half of the contacts are lost within a second (measured on the oracle: mean ncon 24 -> 24 over
    8 s with the flick, 17 -> 10 without)."""
    rng = np.random.default_rng(seed)
    qpos, qvel = hand_grasp_states(model, nenv, seed)
    names = model["names"]["joint"]
    rngj = np.asarray(model["jnt_range"], dtype=np.float64).reshape(-1, 2)
    for j, name in enumerate(names):
        if model["jnt_type"][j] != 3:
            continue
        a = int(model["jnt_qposadr"][j])
        t = min(hand_power_grasp_angle(name), 1.5) if name.endswith("J3") else (0.45 if name == "WRJ1" else hand_power_grasp_angle(name))
        qpos[:, a] = np.clip(t + rng.uniform(-0.05, 0.05, nenv), rngj[j, 0] + 1e-3, rngj[j, 1] - 1e-3)
    ca = int(model["jnt_qposadr"][names.index("cube_joint")])
    qpos[:, ca + 0] = 0.067 + rng.uniform(-0.002, 0.002, nenv)
    qpos[:, ca + 2] = 0.1 + 0.012 + 0.025 + 0.002
    return qpos, qvel


def hand_grasp_states(model, nenv, seed=0):
    """Config 5: half-closed grasp + U(-0.05, 0.05) per joint, cube 5 cm in the palm with a random yaw."""
    rng = np.random.default_rng(seed)
    nq, nv = model["nq"], model["nv"]
    qpos = np.tile(np.asarray(model["qpos0"], dtype=np.float64), (nenv, 1))
    qvel = np.zeros((nenv, nv))
    names = model["names"]["joint"]
    for j, name in enumerate(names):
        if model["jnt_type"][j] != 3:
            continue
        a = int(model["jnt_qposadr"][j])
        g = hand_grasp_angle(name)
        rng_j = np.asarray(model["jnt_range"], dtype=np.float64).reshape(-1, 2)[j]
        lo, hi = float(rng_j[0]), float(rng_j[1])
        qpos[:, a] = np.clip(g + rng.uniform(-0.05, 0.05, nenv), lo + 1e-3, hi - 1e-3)
    ca = int(model["jnt_qposadr"][names.index("cube_joint")])
    qpos[:, ca + 0] = 0.07 + rng.uniform(-0.005, 0.005, nenv)
    qpos[:, ca + 1] = rng.uniform(-0.005, 0.005, nenv)
    qpos[:, ca + 2] = 0.139 + rng.uniform(0.0, 0.004, nenv)
    yaw = rng.uniform(-0.3, 0.3, nenv)
    qpos[:, ca + 3] = np.cos(yaw / 2)
    qpos[:, ca + 4:ca + 6] = 0
    qpos[:, ca + 6] = np.sin(yaw / 2)
    return qpos, qvel
