"""What pins the oracle's constraint path while MuJoCo itself is out of reach (oracle/mjo.h: "parity unpinned"): solutions of the
SAME convex problems by independent routes that share no code with oracle/mjo_constraint.c --

  * pyramidal / limit rows: the primal  min_a 1/2 (a - a0)' M (a - a0) + sum_i 1/2 D_i min(J_i a - aref_i, 0)^2  solved by a numpy
    active-set iteration, and its dual  min_{f >= 0} 1/2 f'(J M^-1 J' + R) f + f'(J a0 - aref)  by scipy's bounded L-BFGS-B;
  * elliptic cones: the dual over the friction cones  f_n >= 0, sum (f_j / mu_j)^2 <= f_n^2  by accelerated projected gradient with the closed-form
    second-order-cone projection -- none of the oracle's
    primal cone-zone formulas are used;
  * the Jacobian rows: efc_vel = J qvel against the finite-differenced relative velocity of the two bodies' material points at the
    contact, projected on the contact frame -- kinematics only (which refdyn.py pins independently).
"""
import os

import numpy as np
import pytest
from scipy import optimize

from mujoco_ros_pkgs_amd import mjcf
from test_gpu_contact import scenario_states


def dense_M(model, qM):
    nv = model["nv"]
    M = np.zeros((nv, nv))
    for i in range(nv):
        adr, j = model["dof_Madr"][i], i
        while j >= 0:
            M[i, j] = M[j, i] = qM[adr]
            adr += 1
            j = model["dof_parentid"][j]
    return M


def forward(po, model, qpos, qvel, ctrl=None):
    d = po.OracleData(model)
    d.qpos[:] = qpos
    d.qvel[:] = qvel
    if ctrl is not None:
        d.ctrl[:] = ctrl
    d.forward()
    return d


def problem(model, d):
    nv, n = model["nv"], int(d.nefc[0])
    M = dense_M(model, d.qM)
    J = d.efc_J[:n * nv].reshape(n, nv).copy()
    return M, J, d.efc_D[:n].copy(), d.efc_R[:n].copy(), d.efc_aref[:n].copy(), d.qacc_smooth.copy(), d.efc_type[:n].copy()


@pytest.mark.parametrize("solver", ["Newton", "PGS"])
def test_pyramidal_rows_against_independent_primal_and_dual(oracle_built, solver):
    path = os.path.join(mjcf.ASSET_DIR, "franka_table.xml")
    model = mjcf.compile_xml_file(path, override={"solver": solver}, nconmax=48, nefcmax=201 if solver == "Newton" else 128)
    qpos, qvel = scenario_states(model, 10, seed=8)
    checked = 0
    for e in range(10):
        d = forward(oracle_built, model, qpos[e], qvel[e])
        M, J, D, R, aref, a0, types = problem(model, d)
        n = len(D)
        if n == 0:
            continue
        assert np.all(types >= 3), "scenario has limit / contact rows only (one-sided)"
        # primal by active-set iteration (exact for a piecewise-quadratic convex cost)
        a = a0.copy()
        for _ in range(200):
            act = (J @ a - aref) < 0
            H = M + J[act].T @ (D[act, None] * J[act])
            a_new = np.linalg.solve(H, M @ a0 + J[act].T @ (D[act] * aref[act]))
            if np.array_equal(act, (J @ a_new - aref) < 0):
                a = a_new
                break
            a = a_new
        f_primal = np.where(J @ a - aref < 0, -D * (J @ a - aref), 0.0)
        # dual by bounded quasi-Newton
        A = J @ np.linalg.solve(M, J.T) + np.diag(R)
        b = J @ a0 - aref
        res = optimize.minimize(lambda f: (0.5 * f @ A @ f + f @ b, A @ f + b), np.maximum(f_primal, 0), jac=True, method="L-BFGS-B",
                                bounds=[(0, None)] * n, options=dict(maxiter=5000, ftol=1e-16, gtol=1e-12))
        a_dual = a0 + np.linalg.solve(M, J.T @ res.x)
        sc = 1 + np.abs(a).max()
        assert np.abs(a - a_dual).max() <= 1e-6 * sc, "the two independent routes disagree: the test is broken"
        # Newton converges quadratically; PGS stops at a cost improvement of 1e-8 (first order: ~1e-4 in the solution) or at its
        # 100-sweep cap, where it is only as good as a hundred Gauss-Seidel sweeps get
        capped = solver == "PGS" and int(d.solver_iter[0]) >= model["iterations"]
        tol = 1e-7 if solver == "Newton" else (5e-3 if capped else 1e-4)
        assert np.abs(d.qacc - a).max() <= tol * sc, f"env {e}: oracle {solver} qacc vs independent primal {np.abs(d.qacc - a).max():.2e}"
        fs = 1 + np.abs(f_primal).max()
        assert np.abs(d.efc_force[:n] - f_primal).max() <= 10 * tol * fs
        checked += 1
    assert checked >= 8


def test_elliptic_cones_against_independent_dual(oracle_built):
    path = os.path.join(mjcf.ASSET_DIR, "franka_table.xml")
    model = mjcf.compile_xml_file(path, override={"solver": "Newton", "cone": "elliptic"}, nconmax=24, nefcmax=120)
    qpos, qvel = scenario_states(model, 12, seed=9)
    checked = 0
    for e in range(12):
        d = forward(oracle_built, model, qpos[e], qvel[e])
        M, J, D, R, aref, a0, types = problem(model, d)
        n, ncon = len(D), int(d.ncon[0])
        if n == 0 or n > 90:
            continue
        A = J @ np.linalg.solve(M, J.T) + np.diag(R)
        b = J @ a0 - aref
        # accelerated projected gradient on the dual in the scaled variables y_j = f_j / mu_j (normal: y_0 = f_0), where every
        # contact's cone is the standard second-order cone |y_t| <= y_n with its closed-form Euclidean projection
        scale = np.ones(n)
        blocks, scalar_pos = [], []
        for r in range(n):
            if types[r] in (3, 4, 5):
                scalar_pos.append(r)
        for c in range(ncon):
            adr, dim = int(d.contact_efc_address[c]), int(d.contact_dim[c])
            if adr < 0:
                continue
            if dim == 1:
                continue
            scale[adr + 1:adr + dim] = d.contact_friction[5 * c:5 * c + 5][:dim - 1]
            blocks.append((adr, dim))
        As = scale[:, None] * A * scale[None, :]
        bs = scale * b

        def project(y):
            y = y.copy()
            y[scalar_pos] = np.maximum(y[scalar_pos], 0)
            for adr, dim in blocks:
                yn, yt = y[adr], y[adr + 1:adr + dim]
                t = np.linalg.norm(yt)
                if t <= yn:
                    continue
                if t <= -yn:
                    y[adr:adr + dim] = 0
                else:
                    k = 0.5 * (yn + t)
                    y[adr] = k
                    y[adr + 1:adr + dim] = yt * (k / t)
            return y

        Lc = np.linalg.eigvalsh(As)[-1]
        y = project(np.array(d.efc_force[:n]) / scale * 0.5)
        z, tk = y.copy(), 1.0
        for it in range(400000):
            y_new = project(z - (As @ z + bs) / Lc)
            if (y_new - y) @ (z - y_new) > 0:   # restart
                z, tk = y_new.copy(), 1.0
            else:
                tn = 0.5 * (1 + np.sqrt(1 + 4 * tk * tk))
                z = y_new + (tk - 1) / tn * (y_new - y)
                tk = tn
            done = np.abs(y_new - y).max() <= 1e-13 * (1 + np.abs(y_new).max())
            y = y_new
            if done and it > 100:
                break

        class res:  # noqa: N801
            x = y * scale
        a_dual = a0 + np.linalg.solve(M, J.T @ res.x)
        sc = 1 + np.abs(a_dual).max()
        assert np.abs(d.qacc - a_dual).max() <= 2e-5 * sc, f"env {e}: oracle Newton (elliptic) vs projected-gradient dual {np.abs(d.qacc - a_dual).max():.2e}"
        assert np.abs(d.efc_force[:n] - res.x).max() <= 2e-4 * (1 + np.abs(res.x).max())
        checked += 1
    assert checked >= 5


def _cone_grad(f, adr, dim, mu):
    g = np.zeros_like(f)
    g[adr] = 2 * f[adr]
    g[adr + 1:adr + dim] = -2 * f[adr + 1:adr + dim] / mu ** 2
    return g


def _body_point_velocity(po, model, qpos, qvel, body, point_world, eps=1e-6):
    """Velocity of the material point of `body` currently at `point_world`, by central differences of the kinematics along qvel."""
    def pose(sign):
        d = po.OracleData(model)
        d.qpos[:] = qpos
        d.qvel[:] = sign * eps * np.asarray(qvel) / model["timestep"][0]
        d.call("kinematics")
        # integrate the positions by one Euler step of velocity sign * eps * qvel / dt over dt = eps-equivalent displacement
        d.call("euler_positions_only") if False else None
        return d
    # displace with the oracle's own position integrator is NOT independent; use the exponential map here
    def displaced(sign):
        q = np.array(qpos, dtype=np.float64)
        for j in range(model["njnt"]):
            t, qa, da = model["jnt_type"][j], model["jnt_qposadr"][j], model["jnt_dofadr"][j]
            if t >= 2:
                q[qa] += sign * eps * qvel[da]
            else:
                if t == 0:
                    q[qa:qa + 3] += sign * eps * qvel[da:da + 3]
                    qa, da = qa + 3, da + 3
                w = sign * eps * np.asarray(qvel[da:da + 3])   # body-frame angular velocity
                ang = np.linalg.norm(w)
                dq = np.array([1.0, 0, 0, 0]) if ang == 0 else np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * w / ang])
                a, b = q[qa:qa + 4], dq
                q[qa:qa + 4] = [a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                                a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]]
        d = po.OracleData(model)
        d.qpos[:] = q
        d.call("kinematics")
        return d.xpos[3 * body:3 * body + 3].copy(), d.xmat[9 * body:9 * body + 9].reshape(3, 3).copy()
    d0 = po.OracleData(model)
    d0.qpos[:] = qpos
    d0.call("kinematics")
    p0, R0 = d0.xpos[3 * body:3 * body + 3], d0.xmat[9 * body:9 * body + 9].reshape(3, 3)
    local = R0.T @ (np.asarray(point_world) - p0)
    pp, Rp = displaced(+1)
    pm, Rm = displaced(-1)
    return ((pp + Rp @ local) - (pm + Rm @ local)) / (2 * eps)


@pytest.mark.parametrize("cone", ["pyramidal", "elliptic"])
def test_contact_and_limit_jacobians_by_finite_differences(oracle_built, cone):
    path = os.path.join(mjcf.ASSET_DIR, "franka_table.xml")
    model = mjcf.compile_xml_file(path, override={"solver": "Newton", "cone": cone}, nconmax=24, nefcmax=120)
    qpos, qvel = scenario_states(model, 6, seed=10)
    seen = 0
    for e in range(6):
        d = forward(oracle_built, model, qpos[e], qvel[e])
        n, ncon, nv = int(d.nefc[0]), int(d.ncon[0]), model["nv"]
        vel = d.efc_vel[:n]
        for r in range(n):
            if d.efc_type[r] == 3:  # joint limit: d(dist)/dt = -+ qvel
                j = d.efc_id[r]
                da, qa = model["jnt_dofadr"][j], model["jnt_qposadr"][j]
                lo, hi = model["jnt_range"][j]
                side = -1.0 if (hi - qpos[e][qa]) < (qpos[e][qa] - lo) else 1.0
                assert abs(vel[r] - side * qvel[e][da]) < 1e-12
        for c in range(ncon):
            adr, dim = int(d.contact_efc_address[c]), int(d.contact_dim[c])
            if adr < 0:
                continue
            g1, g2 = d.contact_geom[2 * c], d.contact_geom[2 * c + 1]
            b1, b2 = model["geom_bodyid"][g1], model["geom_bodyid"][g2]
            pos = d.contact_pos[3 * c:3 * c + 3]
            frame = d.contact_frame[9 * c:9 * c + 9].reshape(3, 3)
            v1 = _body_point_velocity(oracle_built, model, qpos[e], qvel[e], b1, pos) if b1 else np.zeros(3)
            v2 = _body_point_velocity(oracle_built, model, qpos[e], qvel[e], b2, pos) if b2 else np.zeros(3)
            rel = frame @ (v2 - v1)   # relative velocity of body 2 w.r.t. body 1 in the contact frame (normal first)
            mu = d.contact_friction[5 * c]
            if cone == "elliptic":
                assert np.allclose(vel[adr:adr + 3], rel, rtol=0, atol=1e-6 * (1 + np.abs(rel).max()))
            else:
                want = [rel[0] + mu * rel[1], rel[0] - mu * rel[1], rel[0] + mu * rel[2], rel[0] - mu * rel[2]]
                assert np.allclose(vel[adr:adr + 4], want, rtol=0, atol=1e-6 * (1 + np.abs(rel).max()))
            seen += 1
    assert seen >= 10
