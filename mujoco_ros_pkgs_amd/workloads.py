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
