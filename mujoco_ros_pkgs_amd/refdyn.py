"""Independent numpy rigid-body dynamics (Jacobian / Lagrangian formulation).

Two uses, neither on the product's step path:

1. ``invweight0`` restates the part of MuJoCo's ``mj_setConst`` ([UPSTREAM] engine_setconst.c, called
   by the reference at /root/reference mujoco_ros/src/callbacks.cpp:254,582 and implicitly by
   ``mj_loadXML``) that fills ``dof_invweight0`` / ``body_invweight0`` -- model-compile-time
   constants the constraint stage reads.
2. The tests use ``mass_matrix`` / ``bias_lagrange`` as a *differently derived* check of the C
   oracle's CRB and RNE (SURVEY.md §8c substitute oracle (ii)): M from body Jacobians, bias from
   finite-differenced Lagrangian terms.  Nothing here shares code or algorithm with oracle/ or csrc/.
"""
from __future__ import annotations

import numpy as np

JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = 0, 1, 2, 3


def _quat_mul(a, b):
    return np.array([
        a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
        a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
        a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
        a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


def _quat2mat(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _axis_angle_quat(axis, angle):
    return np.concatenate([[np.cos(angle / 2)], np.asarray(axis) * np.sin(angle / 2)])


def kinematics(m, qpos):
    """World poses of bodies, inertial frames, joint anchors/axes for generalized position qpos."""
    nb, nj = m["nbody"], m["njnt"]
    xpos = np.zeros((nb, 3))
    xquat = np.tile(np.array([1.0, 0, 0, 0]), (nb, 1))
    xanchor = np.zeros((nj, 3))
    xaxis = np.zeros((nj, 3))
    for b in range(1, nb):
        p = m["body_parentid"][b]
        ja, jn = m["body_jntadr"][b], m["body_jntnum"][b]
        if jn == 1 and m["jnt_type"][ja] == JNT_FREE:
            qa = m["jnt_qposadr"][ja]
            pos = qpos[qa:qa + 3].copy()
            q = qpos[qa + 3:qa + 7] / np.linalg.norm(qpos[qa + 3:qa + 7])
            xanchor[ja] = pos
            xaxis[ja] = m["jnt_axis"][ja]
        else:
            Rp = _quat2mat(xquat[p])
            pos = xpos[p] + Rp @ m["body_pos"][b]
            q = _quat_mul(xquat[p], m["body_quat"][b])
            for j in range(ja, ja + jn):
                qa = m["jnt_qposadr"][j]
                R = _quat2mat(q)
                xaxis[j] = R @ m["jnt_axis"][j]
                xanchor[j] = pos + R @ m["jnt_pos"][j]
                t = m["jnt_type"][j]
                if t == JNT_SLIDE:
                    pos = pos + xaxis[j] * (qpos[qa] - m["qpos0"][qa])
                else:
                    if t == JNT_BALL:
                        ql = qpos[qa:qa + 4] / np.linalg.norm(qpos[qa:qa + 4])
                    else:
                        ql = _axis_angle_quat(m["jnt_axis"][j], qpos[qa] - m["qpos0"][qa])
                    q = _quat_mul(q, ql)
                    pos = xanchor[j] - _quat2mat(q) @ m["jnt_pos"][j]
        xpos[b] = pos
        xquat[b] = q / np.linalg.norm(q)
    xmat = np.array([_quat2mat(q) for q in xquat])
    xipos = np.array([xpos[b] + xmat[b] @ m["body_ipos"][b] for b in range(nb)])
    ximat = np.array([xmat[b] @ _quat2mat(m["body_iquat"][b]) for b in range(nb)])
    return dict(xpos=xpos, xquat=xquat, xmat=xmat, xipos=xipos, ximat=ximat, xanchor=xanchor, xaxis=xaxis)


def jac_point(m, kin, body, point):
    """Translational (at ``point``, fixed to ``body``) and rotational Jacobians, 3 x nv each."""
    nv = m["nv"]
    jp = np.zeros((3, nv))
    jr = np.zeros((3, nv))
    b = body
    while b > 0:
        ja, jn = m["body_jntadr"][b], m["body_jntnum"][b]
        for j in range(ja, ja + jn):
            d = m["jnt_dofadr"][j]
            t = m["jnt_type"][j]
            if t == JNT_HINGE:
                ax = kin["xaxis"][j]
                jr[:, d] = ax
                jp[:, d] = np.cross(ax, point - kin["xanchor"][j])
            elif t == JNT_SLIDE:
                jp[:, d] = kin["xaxis"][j]
            elif t == JNT_BALL:
                for k in range(3):
                    ax = kin["xmat"][b][:, k]
                    jr[:, d + k] = ax
                    jp[:, d + k] = np.cross(ax, point - kin["xanchor"][j])
            else:  # free
                for k in range(3):
                    jp[k, d + k] = 1.0
                    ax = kin["xmat"][b][:, k]
                    jr[:, d + 3 + k] = ax
                    jp[:, d + 3 + k] = np.cross(ax, point - kin["xpos"][b])
        b = m["body_parentid"][b]
    return jp, jr


def mass_matrix(m, qpos, kin=None):
    """Dense joint-space inertia  M = sum_b m_b Jp'Jp + Jr' (R I R') Jr  + diag(armature)."""
    kin = kin or kinematics(m, qpos)
    nv = m["nv"]
    M = np.diag(np.asarray(m["dof_armature"], dtype=np.float64)).copy() if nv else np.zeros((0, 0))
    for b in range(1, m["nbody"]):
        if m["body_mass"][b] == 0 and not np.any(m["body_inertia"][b]):
            continue
        jp, jr = jac_point(m, kin, b, kin["xipos"][b])
        Iw = kin["ximat"][b] @ np.diag(m["body_inertia"][b]) @ kin["ximat"][b].T
        M += m["body_mass"][b] * jp.T @ jp + jr.T @ Iw @ jr
    return M


def potential(m, qpos):
    kin = kinematics(m, qpos)
    g = np.asarray(m["gravity"], dtype=np.float64)
    return -sum(m["body_mass"][b] * np.dot(g, kin["xipos"][b]) for b in range(1, m["nbody"]))


def bias_lagrange(m, qpos, qvel, eps=1e-6):
    """Coriolis + centrifugal + gravity generalized force c(q,v) such that  M a + c = tau,
    from the Lagrangian (central finite differences of M(q) and V(q)).  Hinge/slide joints only
    (qpos is then a vector space and d/dt qpos = qvel)."""
    if np.any(np.asarray(m["jnt_type"]) < JNT_SLIDE):
        raise ValueError("bias_lagrange supports hinge/slide models only")
    n = m["nv"]
    dM = np.zeros((n, n, n))  # dM[k] = dM/dq_k
    dV = np.zeros(n)
    for k in range(n):
        e = np.zeros(n)
        e[k] = eps
        dM[k] = (mass_matrix(m, qpos + e) - mass_matrix(m, qpos - e)) / (2 * eps)
        dV[k] = (potential(m, qpos + e) - potential(m, qpos - e)) / (2 * eps)
    Mdot = np.tensordot(qvel, dM, axes=(0, 0))
    c = Mdot @ qvel - 0.5 * np.array([qvel @ dM[k] @ qvel for k in range(n)]) + dV
    return c


def invweight0(m):
    """dof_invweight0[nv], body_invweight0[nbody,2], stat.meaninertia at qpos0 (restates mj_setConst's set0)."""
    nv, nb = m["nv"], m["nbody"]
    dof_inv = np.zeros(nv)
    body_inv = np.zeros((nb, 2))
    if nv == 0:
        return dof_inv, body_inv, 1.0
    kin = kinematics(m, np.asarray(m["qpos0"], dtype=np.float64))
    M = mass_matrix(m, m["qpos0"], kin)
    Minv = np.linalg.inv(M)
    for b in range(1, nb):
        if m["body_weldid"][b] == 0:
            continue
        jp, jr = jac_point(m, kin, b, kin["xipos"][b])
        A = np.vstack([jp, jr]) @ Minv @ np.vstack([jp, jr]).T
        body_inv[b, 0] = (A[0, 0] + A[1, 1] + A[2, 2]) / 3
        body_inv[b, 1] = (A[3, 3] + A[4, 4] + A[5, 5]) / 3
    for j in range(m["njnt"]):
        d = m["jnt_dofadr"][j]
        t = m["jnt_type"][j]
        if t in (JNT_HINGE, JNT_SLIDE):
            dof_inv[d] = Minv[d, d]
        elif t == JNT_BALL:
            dof_inv[d:d + 3] = np.trace(Minv[d:d + 3, d:d + 3]) / 3
        else:
            dof_inv[d:d + 3] = np.trace(Minv[d:d + 3, d:d + 3]) / 3
            dof_inv[d + 3:d + 6] = np.trace(Minv[d + 3:d + 6, d + 3:d + 6]) / 3
    return dof_inv, body_inv, float(np.mean(np.diag(M)))
