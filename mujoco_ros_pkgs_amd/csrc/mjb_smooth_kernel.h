// mjb_smooth_kernel.h -- the SMOOTH half of a constrained step in lane = env form (SURVEY.md §7 "measure a 64-envs-per-wave
// (lane = env) variant for the smooth-dynamics phases"; §8a rows A1 - A3, A8 - A9, A12).
//
// The constrained kernels (mjb_step.hip) give one env a whole wavefront: right for collision, the rows and the solver, wasteful for the
// smooth stages, where a 15-dof model keeps a handful of lanes busy (config 3: 65 k of a wave-step's 213 k cycles, 7.2 x of the executed
// fp64 slots idle or redundant).  Here ONE LANE owns an env for exactly those stages -- kinematics incl. geom / site frames, inertias,
// cdof, velocities, RNE, passive and actuator forces, composite inertias, qM, both L'DL factors (M, and M + h B for Euler's implicit
// damping), qacc_smooth -- and leaves what the constraint stages read in the env's HAND-OFF record in HBM (mjb_dev.h: hand-off order);
// mjb_cstep_kernel (mjb_step.hip) picks it up with one env per wavefront: collision, make_constraint, PGS, Euler.  One launch = one step of
// every env of its range; the host alternates the two kernels on streams of env slices (two by default), so this kernel's latency hides behind
// the other slice's constraint stages.  Opt-in (mjb_set_split_step): measurements and the reason in DESIGN.md §11, profiles/r06_split_step.txt.
//
// Same template idea as mjb_lane_env_kernel.h (a lane's "arrays" are registers: every index is a compile-time constant of the model's
// integer structure `T`, a SmTopo_* of csrc/smooth_topos.h), widened to what config 3 and the reference's own worlds need: free and ball
// joints (pendulum_world.xml:18-38), qpos / dof addresses that differ from the joint index, geoms.  Spatial quantities are taken about the
// origin of the tree's root body (any common point gives the same qM / qfrc_bias / contact Jacobians; the record's subtree_com entries hold
// that point).  Arithmetic otherwise follows oracle/mjo_smooth.c stage by stage; reciprocals are Newton-refined hardware seeds, so results
// agree with the generic kernels to rounding, not bit for bit.
#pragma once
#include "mjb_lane_env_kernel.h"

namespace mjb_sm {

using namespace mjb_le;

template <class T> struct Sq {
	static constexpr bool anc(int a, int i)  // dof a is dof i or one of its ancestors
	{
		for (int j = i; j >= 0; j = T::dof_parentid[j])
			if (j == a) return true;
		return false;
	}
	static constexpr bool needed(int b)  // the body moves: it, or an ancestor, carries a joint
	{
		for (int a = b; a > 0; a = T::body_parentid[a])
			if (T::body_jnt[a] >= 0) return true;
		return false;
	}
	static constexpr int ord(int b)
	{
		int n = 0;
		for (int a = 1; a < b; a++)
			if (needed(a)) n++;
		return n;
	}
	static constexpr int nneeded() { return ord(T::NBODY); }
	static constexpr int ent_i(int e)  // entry e of MuJoCo's sparse qM -> dof i, ancestor a
	{
		int i = 0;
		for (int d = 0; d < T::NV; d++)
			if (T::dof_Madr[d] <= e) i = d;
		return i;
	}
	static constexpr int ent_a(int e)
	{
		int a = ent_i(e);
		for (int k = T::dof_Madr[ent_i(e)]; k < e; k++) a = T::dof_parentid[a];
		return a;
	}
	static constexpr int jtype(int b) { return T::body_jnt[b] >= 0 ? T::jnt_type[T::body_jnt[b]] : -1; }
	static constexpr int ndof(int b) { return jtype(b) == MJB_JNT_FREE ? 6 : (jtype(b) == MJB_JNT_BALL ? 3 : (jtype(b) >= 0 ? 1 : 0)); }
	static constexpr int dofadr(int b) { return T::body_jnt[b] >= 0 ? T::jnt_dofadr[T::body_jnt[b]] : 0; }
	static constexpr int qadr(int b) { return T::body_jnt[b] >= 0 ? T::jnt_qposadr[T::body_jnt[b]] : 0; }
	// LDS pair slots of a lane: cdof (3 per dof), then per moving body its force (3) and its cinert (5)
	static constexpr int CD0 = 0, CF0 = 3 * T::NV;
	static constexpr int CI0 = CF0 + 3 * nneeded();
	static constexpr int nslots() { return CI0 + 5 * nneeded(); }
	static constexpr int bytes() { return nslots() * 64 * 16; }
};

// mju_normalize4 without a per-lane branch (a vanishing quaternion becomes the identity, as there)
DEVI void normalize4_full(double *q)
{
	const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
	const bool tiny = !(n2 >= MJB_MINVAL * MJB_MINVAL);
	const double r = frsq(tiny ? 1.0 : n2), n = n2 * r;
	const double s = (fabs(n - 1) > MJB_MINVAL) ? r : 1.0;
	q[0] = tiny ? 1.0 : q[0] * s;
	q[1] = tiny ? 0.0 : q[1] * s;
	q[2] = tiny ? 0.0 : q[2] * s;
	q[3] = tiny ? 0.0 : q[3] * s;
}

// One step's smooth half of one env per lane.  env: the lane's env (a lane without one -- !live -- recomputes a neighbour's and stores the same
// values); step: the env's step index (the ctrl-noise key); last: position / velocity sensors go out (wave-uniform); lp: the lane's pair slots
// (slot q at lp[64 * q]; the kernel keeps them in LDS: cdof, and per moving body its force and its cinert -- Sq<T>).
template <class T>
DEVI void smooth_lane_env_core(const KernelParams MJB_AS4 *__restrict__ P, const int env, const bool live, const unsigned int step, const bool last, Pair *const lp)
{
	constexpr int NB = T::NBODY, NV = T::NV, NQ = T::NQ, NU = T::NU, NG = T::NGEOM, NM = T::NM;
	using Q = Sq<T>;
	const DevModel MJB_AS4 &m = P->m;
	const DevState MJB_AS4 &s = P->s;
	const size_t ev = (size_t)env;
	double *const H = s.handoff + ev * (size_t)s.handoff_stride;
	const LeTapeHdr MJB_AS4 *th = reinterpret_cast<const LeTapeHdr MJB_AS4 *>(m.le_tape);
	const LeTapeBody MJB_AS4 *tb = reinterpret_cast<const LeTapeBody MJB_AS4 *>(th + 1);
	const LeTapeAct MJB_AS4 *const ta = reinterpret_cast<const LeTapeAct MJB_AS4 *>(tb + NB);

	// ---- state
	double qpos[NQ], qvel[NV > 0 ? NV : 1], cn[NU > 0 ? NU : 1], ctrl[NU > 0 ? NU : 1], qfa[NV > 0 ? NV : 1];
	sfor<NQ>([&](auto I) { qpos[I] = s.qpos[ev * NQ + I]; });
	sfor<NV>([&](auto I) { qvel[I] = s.qvel[ev * NV + I]; qfa[I] = s.qfrc_applied[ev * NV + I]; });
	double time = s.time[ev];
	const bool nz_on = P->nz.enabled != 0;
	sfor<NU>([&](auto I) { cn[I] = s.ctrlnoise[ev * NU + I]; ctrl[I] = s.ctrl[ev * NU + I]; });
	// H10: the reference's ctrl-noise injector (mujoco_env.cpp:469-481) runs ahead of mj_step
	if (nz_on) {
		const double rate = P->nz.rate, scale = P->nz.scale;
		const unsigned long long seed = P->nz.seed, genv = (unsigned long long)(P->nz.env_offset + env);
		double *zl = reinterpret_cast<double *>(lp);  // (one copy of the generator in the instruction stream; the pair slots are free at this point)
#pragma nounroll
		for (int i = 0; i < NU; i++) zl[128 * i] = philox_normal(seed, genv, step, (unsigned int)i);
		sfor<NU>([&](auto I) { cn[I] = rate * cn[I] + scale * zl[128 * I]; ctrl[I] = cn[I]; });
	}
	// ---- mj_checkPos / mj_checkVel (qpos first: its reset hides a bad qvel) -> mj_resetData
	{
		bool badp = false, badv = false;
		sfor<NQ>([&](auto I) { badp |= bad_val(qpos[I]); });
		sfor<NV>([&](auto I) { badv |= bad_val(qvel[I]); });
		const bool bad = badp || badv;
		if (__builtin_amdgcn_ballot_w64(bad)) {  // (wave-uniform; selects inside, no per-lane branch)
			atomicAdd(s.nwarn + MJB_WARN_BADQPOS, (badp && live) ? 1ull : 0ull);
			atomicAdd(s.nwarn + MJB_WARN_BADQVEL, (!badp && badv && live) ? 1ull : 0ull);
			sfor<NQ>([&](auto I) { const double q0 = pins(m.qpos0[I]); qpos[I] = bad ? q0 : qpos[I]; });
			sfor<NV>([&](auto I) {
				qvel[I] = bad ? 0.0 : qvel[I];
				qfa[I] = bad ? 0.0 : qfa[I];
				const double w = pinv(s.qacc_warmstart[ev * NV + I]);
				s.qacc_warmstart[ev * NV + I] = bad ? 0.0 : w;
				s.qfrc_applied[ev * NV + I] = qfa[I];
				s.qvel[ev * NV + I] = qvel[I];
			});
			sfor<NQ>([&](auto I) { s.qpos[ev * NQ + I] = qpos[I]; });
			sfor<NU>([&](auto I) { cn[I] = bad ? 0.0 : cn[I]; ctrl[I] = bad ? 0.0 : ctrl[I]; });
			time = bad ? 0.0 : time;
			s.time[ev] = time;
		}
	}
	sfor<NU>([&](auto I) { s.ctrlnoise[ev * NU + I] = cn[I]; s.ctrl[ev * NU + I] = ctrl[I]; });

	const bool sens_on = last && !(m.disableflags & MJB_DSBL_SENSOR);
	double *sd = s.sensordata + ev * T::NSENSORDATA;
	double grav[3];
	{
		const bool g_on = !(m.disableflags & MJB_DSBL_GRAVITY);
		for (int k = 0; k < 3; k++) grav[k] = g_on ? th->gravity[k] : 0.0;
	}
	const bool pas_on = !(m.disableflags & MJB_DSBL_PASSIVE);
	// mjData.energy (mjENBL_ENERGY) of the launch's last step: mj_energyPos gathered along the sweep, mj_energyVel from qM (oracle/mjo_smooth.c mjo_energy)
	const bool e_on = last && (m.enableflags & MJB_ENBL_ENERGY);
	const bool eg_on = e_on && !(m.disableflags & MJB_DSBL_GRAVITY);
	double pe = 0;
	double f[NV > 0 ? NV : 1];  // qfrc_passive + qfrc_applied + qfrc_actuator, then (- qfrc_bias) qfrc_smooth
	sfor<NV>([&](auto I) { f[I] = qfa[I] - (pas_on ? m.dof_damping[I] * qvel[I] : 0.0); });

	// world geoms (body 0)
	auto geoms_of = [&](auto Bq, const double *xp, const double *xq, const double *xm) {
		constexpr int b = Bq;
		sfor<NG>([&](auto G) {
			constexpr int g = G;
			if constexpr (T::geom_bodyid[g] == b) {
				double gp[3], gm[9];
				if constexpr (T::geom_sameframe[g]) {
					for (int k = 0; k < 3; k++) gp[k] = xp[k];
					for (int k = 0; k < 9; k++) gm[k] = xm[k];
				} else {
					double lp3[3], lq[4], v[3], q[4];
					ldc3(lp3, m.geom_pos + 3 * g);
					ldc4(lq, m.geom_quat + 4 * g);
					matvec3(v, xm, lp3);
					for (int k = 0; k < 3; k++) gp[k] = v[k] + xp[k];
					qmul(q, xq, lq);
					quat2mat_nocheck(gm, q);
				}
				for (int k = 0; k < 3; k++) H[T::H_GEOM_XPOS + 3 * g + k] = gp[k];
				for (int k = 0; k < 9; k++) H[T::H_GEOM_XMAT + 9 * g + k] = gm[k];
			}
		});
	};
	// position-stage sensors on a body's frames
	auto frame_sensors = [&](auto Bq, const double *xp, const double *xq, const double *xm, const double *xipos) {
		constexpr int b = Bq;
		if (sens_on) {
			sfor<T::NSENSOR>([&](auto S) {
				constexpr int i = S, type = T::sensor_type[i], ot = T::sensor_objtype[i], id = T::sensor_objid[i], adr = T::sensor_adr[i];
				if constexpr (type == MJB_SENS_FRAMEPOS || type == MJB_SENS_FRAMEQUAT) {
					constexpr int sb = ot == MJB_OBJ_SITE ? T::site_bodyid[id] : id;
					if constexpr (sb == b) {
						double o3[3], o4[4];
						if constexpr (ot == MJB_OBJ_SITE) {
							if constexpr (type == MJB_SENS_FRAMEPOS) {
								if constexpr (T::site_sameframe[id]) {
									for (int k = 0; k < 3; k++) o3[k] = xp[k];
								} else {
									double sp[3], v[3];
									ldc3(sp, m.site_pos + 3 * id);
									matvec3(v, xm, sp);
									for (int k = 0; k < 3; k++) o3[k] = v[k] + xp[k];
								}
							} else {
								double sq[4];
								ldc4(sq, m.site_quat + 4 * id);
								qmul(o4, xq, sq);
							}
						} else if constexpr (ot == MJB_OBJ_BODY) {
							if constexpr (type == MJB_SENS_FRAMEPOS) {
								for (int k = 0; k < 3; k++) o3[k] = xipos[k];
							} else {
								double iq[4];
								ldc4(iq, m.body_iquat + 4 * b);
								qmul(o4, xq, iq);
							}
						} else {  // xbody
							for (int k = 0; k < 3; k++) o3[k] = xp[k];
							for (int k = 0; k < 4; k++) o4[k] = xq[k];
						}
						if constexpr (type == MJB_SENS_FRAMEPOS) {
							const double cut = m.sensor_cutoff[i];
							for (int k = 0; k < 3; k++) sd[adr + k] = cut > 0 ? clampd(o3[k], -cut, cut) : o3[k];
						} else {
							for (int k = 0; k < 4; k++) sd[adr + k] = o4[k];
						}
					}
				}
			});
		}
	};
	{
		const double x0[3] = { 0, 0, 0 }, q0[4] = { 1, 0, 0, 0 }, m0[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
		geoms_of(IC<0>{}, x0, q0, m0);
		frame_sensors(IC<0>{}, x0, q0, m0, x0);
	}

	// ============ one sweep root -> leaf: A1 mj_kinematics (+ geoms, sites), comPos (cinert, cdof), A8 comVel, A9 RNE's forward pass ============
	double xpos[NB][3], xquat[NB][4], xmat[NB][9];
	double cvel[NB][6], cacc[NB][6];
	__builtin_amdgcn_sched_barrier(0);
	sfor<NB>([&](auto B) {
		constexpr int b = B;
		if constexpr (b > 0) {
			constexpr int p = T::body_parentid[b], r = T::body_rootid[b], jt = Q::jtype(b), da = Q::dofadr(b), qa = Q::qadr(b);
			const LeTapeBody MJB_AS4 &t = tb[b];
			double pos[3], quat[4];
			[[maybe_unused]] double xanch[3];
			[[maybe_unused]] bool offc = false;
			if constexpr (jt == MJB_JNT_FREE) {
				for (int k = 0; k < 3; k++) pos[k] = qpos[qa + k];
				for (int k = 0; k < 4; k++) quat[k] = qpos[qa + 3 + k];
				normalize4_full(quat);
				for (int k = 0; k < 4; k++) qpos[qa + 3 + k] = quat[k];  // (mj_kinematics normalises the quaternions IN qpos)
				for (int k = 0; k < 3; k++) xanch[k] = pos[k];
			} else {
				for (int k = 0; k < 3; k++) pos[k] = t.pos[k];
				for (int k = 0; k < 4; k++) quat[k] = t.quat[k];
				if constexpr (p != 0) {
					double v[3], q[4];
					matvec3(v, xmat[p], pos);
					for (int k = 0; k < 3; k++) pos[k] = v[k] + xpos[p][k];
					qmul(q, xquat[p], quat);
					for (int k = 0; k < 4; k++) quat[k] = q[k];
				}
				if constexpr (jt >= 0) {
					const double jp[3] = { t.jpos[0], t.jpos[1], t.jpos[2] };
					offc = jp[0] != 0 || jp[1] != 0 || jp[2] != 0;  // (wave-uniform)
					for (int k = 0; k < 3; k++) xanch[k] = pos[k];
					if (offc) {
						double M0[9], v[3];
						quat2mat_nocheck(M0, quat);
						matvec3(v, M0, jp);
						for (int k = 0; k < 3; k++) xanch[k] += v[k];
					}
					if constexpr (jt == MJB_JNT_HINGE) {
						double sn, cs, ql[4], q[4];
						sincos_nb((qpos[qa] - t.qpos0) * 0.5, &sn, &cs);
						ql[0] = cs; ql[1] = t.jaxis[0] * sn; ql[2] = t.jaxis[1] * sn; ql[3] = t.jaxis[2] * sn;
						qmul(q, quat, ql);
						for (int k = 0; k < 4; k++) quat[k] = q[k];
					} else if constexpr (jt == MJB_JNT_BALL) {
						double ql[4] = { qpos[qa], qpos[qa + 1], qpos[qa + 2], qpos[qa + 3] }, q[4];
						normalize4_full(ql);
						for (int k = 0; k < 4; k++) qpos[qa + k] = ql[k];
						qmul(q, quat, ql);
						for (int k = 0; k < 4; k++) quat[k] = q[k];
					}
				}
			}
			normalize4_full(quat);
			for (int k = 0; k < 4; k++) xquat[b][k] = quat[k];
			quat2mat_nocheck(xmat[b], quat);
			[[maybe_unused]] double xaxis[3];
			if constexpr (jt == MJB_JNT_HINGE || jt == MJB_JNT_SLIDE) {
				// the joint's world axis through the body's FINAL orientation (a hinge turns about it, a slide does not turn)
				const double ax[3] = { t.jaxis[0], t.jaxis[1], t.jaxis[2] };
				matvec3(xaxis, xmat[b], ax);
			}
			if constexpr (jt == MJB_JNT_SLIDE) {
				const double dq = qpos[qa] - t.qpos0;
				for (int k = 0; k < 3; k++) pos[k] += xaxis[k] * dq;
			} else if constexpr (jt == MJB_JNT_HINGE || jt == MJB_JNT_BALL) {
				if (offc) {  // correct for off-centre rotation
					const double jp[3] = { t.jpos[0], t.jpos[1], t.jpos[2] };
					double v[3];
					matvec3(v, xmat[b], jp);
					for (int k = 0; k < 3; k++) pos[k] = xanch[k] - v[k];
				}
			}
			for (int k = 0; k < 3; k++) xpos[b][k] = pos[k];
			if constexpr (r == b) for (int k = 0; k < 3; k++) H[T::H_SUBTREE_COM + 3 * b + k] = pos[k];  // the point the tree's spatial quantities are about
			geoms_of(B, xpos[b], xquat[b], xmat[b]);
			// inertial frame
			double xipos[3];
			if constexpr (T::body_sameframe[b]) {
				for (int k = 0; k < 3; k++) xipos[k] = pos[k];
			} else {
				double ip[3] = { t.ipos[0], t.ipos[1], t.ipos[2] }, v[3];
				matvec3(v, xmat[b], ip);
				for (int k = 0; k < 3; k++) xipos[k] = v[k] + pos[k];
			}
			frame_sensors(B, xpos[b], xquat[b], xmat[b], xipos);
			if (eg_on) pe -= t.mass * (grav[0] * xipos[0] + grav[1] * xipos[1] + grav[2] * xipos[2]);
			if constexpr (Q::needed(b)) {
				// cinert about the tree root's origin: world inertia X Ib X' (Ib = R(iquat) diag(inertia) R(iquat)' from the tape) + the offset's terms
				double ci[10];
				{
					double dif[3];
					for (int k = 0; k < 3; k++) dif[k] = xipos[k] - xpos[r][k];
					const double *X = xmat[b];
					const double mass = t.mass, ixx = t.ibody[0], iyy = t.ibody[1], izz = t.ibody[2], ixy = t.ibody[3], ixz = t.ibody[4], iyz = t.ibody[5];
					double Tm[9];
					for (int rr = 0; rr < 3; rr++) {
						Tm[3 * rr + 0] = X[3 * rr] * ixx + X[3 * rr + 1] * ixy + X[3 * rr + 2] * ixz;
						Tm[3 * rr + 1] = X[3 * rr] * ixy + X[3 * rr + 1] * iyy + X[3 * rr + 2] * iyz;
						Tm[3 * rr + 2] = X[3 * rr] * ixz + X[3 * rr + 1] * iyz + X[3 * rr + 2] * izz;
					}
					ci[0] = Tm[0] * X[0] + Tm[1] * X[1] + Tm[2] * X[2] + mass * (dif[1] * dif[1] + dif[2] * dif[2]);
					ci[1] = Tm[3] * X[3] + Tm[4] * X[4] + Tm[5] * X[5] + mass * (dif[0] * dif[0] + dif[2] * dif[2]);
					ci[2] = Tm[6] * X[6] + Tm[7] * X[7] + Tm[8] * X[8] + mass * (dif[0] * dif[0] + dif[1] * dif[1]);
					ci[3] = Tm[0] * X[3] + Tm[1] * X[4] + Tm[2] * X[5] - mass * dif[0] * dif[1];
					ci[4] = Tm[0] * X[6] + Tm[1] * X[7] + Tm[2] * X[8] - mass * dif[0] * dif[2];
					ci[5] = Tm[3] * X[6] + Tm[4] * X[7] + Tm[5] * X[8] - mass * dif[1] * dif[2];
					ci[6] = mass * dif[0];
					ci[7] = mass * dif[1];
					ci[8] = mass * dif[2];
					ci[9] = mass;
					constexpr int c0 = Q::CI0 + 5 * Q::ord(b);
					for (int k = 0; k < 5; k++) lp[64 * (c0 + k)] = Pair{ ci[2 * k], ci[2 * k + 1] };
				}
				// parent's velocity / acceleration (world, or a jointless chain down from it: at rest, -gravity)
				double cv[6], ca[6];
				if constexpr (p == 0 || !Q::needed(p)) {
					for (int k = 0; k < 6; k++) cv[k] = 0;
					ca[0] = ca[1] = ca[2] = 0;
					for (int k = 0; k < 3; k++) ca[3 + k] = -grav[k];
				} else {
					for (int k = 0; k < 6; k++) { cv[k] = cvel[p][k]; ca[k] = cacc[p][k]; }
				}
				// cdof of the body's dofs (mj_comPos), then mj_comVel's cdof_dot / cvel and mj_rne's cacc, dof group by dof group
				[[maybe_unused]] double off[3];
				if constexpr (jt >= 0) for (int k = 0; k < 3; k++) off[k] = xpos[r][k] - xanch[k];
				auto put_cdof = [&](auto D, const double *cd) {
					constexpr int d = D;
					lp[64 * (Q::CD0 + 3 * d)] = Pair{ cd[0], cd[1] };
					lp[64 * (Q::CD0 + 3 * d + 1)] = Pair{ cd[2], cd[3] };
					lp[64 * (Q::CD0 + 3 * d + 2)] = Pair{ cd[4], cd[5] };
					for (int k = 0; k < 6; k++) H[T::H_CDOF + 6 * d + k] = cd[k];
				};
				if constexpr (jt == MJB_JNT_FREE) {  // translations: cdof_dot = 0, velocity first
					sfor<3>([&](auto K) {
						constexpr int k = K;
						double cd[6] = { 0, 0, 0, 0, 0, 0 };
						cd[3 + k] = 1;
						put_cdof(IC<da + k>{}, cd);
						cv[3 + k] += qvel[da + k];
					});
				}
				if constexpr (jt == MJB_JNT_FREE || jt == MJB_JNT_BALL) {
					constexpr int d0 = da + (jt == MJB_JNT_FREE ? 3 : 0);
					double cd3[3][6], cdd[3][6];
					sfor<3>([&](auto K) {
						constexpr int k = K;
						const double axis[3] = { xmat[b][k], xmat[b][k + 3], xmat[b][k + 6] };
						for (int c = 0; c < 3; c++) cd3[k][c] = axis[c];
						cross3(cd3[k] + 3, axis, off);
						put_cdof(IC<d0 + k>{}, cd3[k]);
						cross_motion(cdd[k], cv, cd3[k]);  // (all three with the velocity before the group's own)
					});
					for (int k = 0; k < 3; k++)
						for (int c = 0; c < 6; c++) {
							ca[c] += cdd[k][c] * qvel[d0 + k];
							cv[c] += cd3[k][c] * qvel[d0 + k];
						}
				} else if constexpr (jt == MJB_JNT_HINGE || jt == MJB_JNT_SLIDE) {
					double cd[6], cdd[6];
					if constexpr (jt == MJB_JNT_SLIDE) {
						cd[0] = cd[1] = cd[2] = 0;
						for (int k = 0; k < 3; k++) cd[3 + k] = xaxis[k];
					} else {
						for (int k = 0; k < 3; k++) cd[k] = xaxis[k];
						cross3(cd + 3, xaxis, off);
					}
					put_cdof(IC<da>{}, cd);
					cross_motion(cdd, cv, cd);
					for (int c = 0; c < 6; c++) {
						ca[c] += cdd[c] * qvel[da];
						cv[c] += cd[c] * qvel[da];
					}
				}
				for (int k = 0; k < 6; k++) { cvel[b][k] = cv[k]; cacc[b][k] = ca[k]; }
				// cfrc_body = cinert * cacc + cvel x* (cinert * cvel): parked in LDS for the backward sweep
				double cf[6], t0[6], t1[6];
				mul_inert_vec(cf, ci, ca);
				mul_inert_vec(t0, ci, cv);
				cross_force(t1, cv, t0);
				constexpr int q0 = Q::CF0 + 3 * Q::ord(b);
				lp[64 * q0] = Pair{ cf[0] + t1[0], cf[1] + t1[1] };
				lp[64 * (q0 + 1)] = Pair{ cf[2] + t1[2], cf[3] + t1[3] };
				lp[64 * (q0 + 2)] = Pair{ cf[4] + t1[4], cf[5] + t1[5] };
			}
			// A8 mj_passive: the joint's spring (dampers were folded into f at the top)
			if constexpr (jt == MJB_JNT_HINGE || jt == MJB_JNT_SLIDE) {
				if (pas_on) f[da] -= t.stiffness * (qpos[qa] - t.spring);
				if (e_on && pas_on) pe += 0.5 * t.stiffness * (qpos[qa] - t.spring) * (qpos[qa] - t.spring);
			} else if constexpr (jt == MJB_JNT_FREE || jt == MJB_JNT_BALL) {
				const double kst = m.jnt_stiffness[T::body_jnt[b]];
				if (pas_on && kst != 0) {  // (wave-uniform)
					constexpr int qq = qa + (jt == MJB_JNT_FREE ? 3 : 0), dd = da + (jt == MJB_JNT_FREE ? 3 : 0);
					if constexpr (jt == MJB_JNT_FREE)
						for (int c = 0; c < 3; c++) {
							const double dq = qpos[qa + c] - m.qpos_spring[qa + c];
							f[da + c] -= kst * dq;
							if (e_on) pe += 0.5 * kst * dq * dq;
						}
					double qs[4], dif[3];
					ldc4(qs, m.qpos_spring + qq);
					quat_sub(dif, qpos + qq, qs);
					for (int c = 0; c < 3; c++) f[dd + c] -= kst * dif[c];
					if (e_on) pe += 0.5 * kst * (dif[0] * dif[0] + dif[1] * dif[1] + dif[2] * dif[2]);
				}
			}
			__builtin_amdgcn_sched_barrier(0);
		}
	});
	// (the normalised quaternions of free / ball joints back to mjData.qpos, as mj_kinematics leaves them)
	sfor<T::NJNT>([&](auto J) {
		constexpr int j = J;
		if constexpr (T::jnt_type[j] == MJB_JNT_FREE || T::jnt_type[j] == MJB_JNT_BALL) {
			constexpr int q0 = T::jnt_qposadr[j] + (T::jnt_type[j] == MJB_JNT_FREE ? 3 : 0);
			for (int k = 0; k < 4; k++) s.qpos[ev * NQ + q0 + k] = qpos[q0 + k];
		}
	});

	// ============ A12 mj_fwdActuation (joint transmission on hinge / slide joints) ============
	{
		const bool act_on = !(m.disableflags & MJB_DSBL_ACTUATION);
		const bool clamp_on = !(m.disableflags & MJB_DSBL_CLAMPCTRL);
		sfor<NU>([&](auto U) {
			constexpr int i = U, j = T::act_jnt[i], qa = T::jnt_qposadr[j], da = T::jnt_dofadr[j];
			double force = 0;
			const LeTapeAct MJB_AS4 &A = ta[i];
			const double gear = A.gear;
			if (act_on) {
				double c = ctrl[i];
				if constexpr (T::act_ctrllimited[i]) {
					if (clamp_on) c = clampd(c, A.ctrllo, A.ctrlhi);
				}
				const double len = qpos[qa] * gear, vel = qvel[da] * gear;
				double gain = A.gain[0], bs = 0;
				if constexpr (T::act_gaintype[i] == MJB_GAIN_AFFINE) gain = gain + A.gain[1] * len + A.gain[2] * vel;
				if constexpr (T::act_biastype[i] == MJB_BIAS_AFFINE) bs = A.bias[0] + A.bias[1] * len + A.bias[2] * vel;
				force = gain * c + bs;
				if constexpr (T::act_forcelimited[i]) force = clampd(force, A.forcelo, A.forcehi);
				f[da] += gear * force;
			}
			if (sens_on) {
				sfor<T::NSENSOR>([&](auto S) {
					constexpr int q = S;
					if constexpr (T::sensor_objid[q] == i && (T::sensor_type[q] == MJB_SENS_ACTUATORFRC || T::sensor_type[q] == MJB_SENS_ACTUATORPOS || T::sensor_type[q] == MJB_SENS_ACTUATORVEL)) {
						double v = T::sensor_type[q] == MJB_SENS_ACTUATORFRC ? force : (T::sensor_type[q] == MJB_SENS_ACTUATORPOS ? qpos[qa] * gear : qvel[da] * gear);
						const double cut = m.sensor_cutoff[q];
						sd[T::sensor_adr[q]] = cut > 0 ? clampd(v, -cut, cut) : v;
					}
				});
			}
		});
	}
	if (sens_on) {
		sfor<T::NSENSOR>([&](auto S) {
			constexpr int q = S, type = T::sensor_type[q];
			if constexpr (type == MJB_SENS_JOINTPOS || type == MJB_SENS_JOINTVEL || type == MJB_SENS_CLOCK) {
				constexpr int jj = T::sensor_objid[q] < 0 ? 0 : T::sensor_objid[q];
				const double v = type == MJB_SENS_CLOCK ? time : (type == MJB_SENS_JOINTPOS ? qpos[T::jnt_qposadr[jj]] : qvel[T::jnt_dofadr[jj]]);
				const double cut = m.sensor_cutoff[q];
				sd[T::sensor_adr[q]] = cut > 0 ? clampd(v, -cut, cut) : v;
			}
		});
	}
	__builtin_amdgcn_sched_barrier(0);

	// ================= A2 mj_crb + A9 RNE backward pass, one sweep leaf -> root =================
	double qM[NV > 0 ? NV : 1][NV > 0 ? NV : 1];  // [i][a], a = i or an ancestor of i (the other entries never exist)
	double csum[NB][6], crbs[NB][10];             // forces / composite inertias of a body's children, summed as the sweep passes them
	sfor<NB - 1>([&](auto Bi) {
		constexpr int b = NB - 1 - Bi;
		constexpr int p = T::body_parentid[b], nd = Q::ndof(b), da = Q::dofadr(b);
		if constexpr (Q::needed(b)) {
			double cf[6], cb[10];
			{
				constexpr int q0 = Q::CF0 + 3 * Q::ord(b), c0 = Q::CI0 + 5 * Q::ord(b);
				const Pair a0 = lp[64 * q0], a1 = lp[64 * (q0 + 1)], a2 = lp[64 * (q0 + 2)];
				cf[0] = a0.a; cf[1] = a0.b; cf[2] = a1.a; cf[3] = a1.b; cf[4] = a2.a; cf[5] = a2.b;
				for (int k = 0; k < 5; k++) {
					const Pair c = lp[64 * (c0 + k)];
					cb[2 * k] = c.a;
					cb[2 * k + 1] = c.b;
				}
			}
			constexpr bool has_child = [] { for (int c = b + 1; c < NB; c++) if (T::body_parentid[c] == b && Q::needed(c)) return true; return false; }();
			if constexpr (has_child) {
				for (int k = 0; k < 6; k++) cf[k] += csum[b][k];
				for (int k = 0; k < 10; k++) cb[k] += crbs[b][k];
			}
			sfor<nd>([&](auto K) {
				constexpr int d = da + K;
				double cd[6], buf[6];
				{
					const Pair a0 = lp[64 * (Q::CD0 + 3 * d)], a1 = lp[64 * (Q::CD0 + 3 * d + 1)], a2 = lp[64 * (Q::CD0 + 3 * d + 2)];
					cd[0] = a0.a; cd[1] = a0.b; cd[2] = a1.a; cd[3] = a1.b; cd[4] = a2.a; cd[5] = a2.b;
				}
				f[d] -= dot6r(cd, cf);
				mul_inert_vec(buf, cb, cd);
				sfor<NV>([&](auto A) {
					constexpr int a = A;
					if constexpr (Q::anc(a, d)) {
						if constexpr (a == d) qM[d][a] = m.dof_armature[d] + dot6r(cd, buf);
						else {
							const Pair a0 = lp[64 * (Q::CD0 + 3 * a)], a1 = lp[64 * (Q::CD0 + 3 * a + 1)], a2 = lp[64 * (Q::CD0 + 3 * a + 2)];
							qM[d][a] = a0.a * buf[0] + a0.b * buf[1] + a1.a * buf[2] + a1.b * buf[3] + a2.a * buf[4] + a2.b * buf[5];
						}
					}
				});
			});
			if constexpr (p > 0 && Q::needed(p)) {
				constexpr bool first = [] { for (int c = b + 1; c < NB; c++) if (T::body_parentid[c] == p && Q::needed(c)) return false; return true; }();
				if constexpr (first) {
					for (int k = 0; k < 6; k++) csum[p][k] = cf[k];
					for (int k = 0; k < 10; k++) crbs[p][k] = cb[k];
				} else {
					for (int k = 0; k < 6; k++) csum[p][k] += cf[k];
					for (int k = 0; k < 10; k++) crbs[p][k] += cb[k];
				}
			}
		}
		__builtin_amdgcn_sched_barrier(0);
	});

	if (e_on) {  // mj_energyVel: 0.5 qvel' M qvel
		double ke = 0;
		sfor<NV>([&](auto I) {
			sfor<NV>([&](auto A) {
				constexpr int i = I, a = A;
				if constexpr (Q::anc(a, i)) ke += (a == i ? 0.5 : 1.0) * qM[i][a] * qvel[i] * qvel[a];
			});
		});
		s.energy[2 * ev] = pe;
		s.energy[2 * ev + 1] = ke;
	}
	// ================= A3 mj_factorM (M and M + h B side by side), A12 mj_fwdAcceleration =================
	const double dt = th->dt;
	double qH[NV > 0 ? NV : 1][NV > 0 ? NV : 1], dinv[NV > 0 ? NV : 1], hinv[NV > 0 ? NV : 1];
	sfor<NV>([&](auto I) {
		sfor<NV>([&](auto A) {
			constexpr int i = I, a = A;
			if constexpr (Q::anc(a, i)) qH[i][a] = qM[i][a] + (a == i ? dt * m.dof_damping[i] : 0.0);
		});
	});
	sfor<NV>([&](auto Ki) {
		constexpr int k = NV - 1 - Ki;
		dinv[k] = frcp(qM[k][k]);
		hinv[k] = frcp(qH[k][k]);
		sfor<NV>([&](auto Ii) {
			constexpr int i = NV - 1 - Ii;  // ancestors of k, nearest first
			if constexpr (i < k && Q::anc(i, k)) {
				const double tm = qM[k][i] * dinv[k], th2 = qH[k][i] * hinv[k];
				sfor<NV>([&](auto A) {
					constexpr int a = A;
					if constexpr (Q::anc(a, i)) {
						qM[i][a] -= tm * qM[k][a];
						qH[i][a] -= th2 * qH[k][a];
					}
				});
				qM[k][i] = tm;
				qH[k][i] = th2;
			}
		});
	});
	// the factors in MuJoCo's sparse layout (dof_Madr: dof i with its k-th ancestor, itself first)
	sfor<NM>([&](auto E) {
		constexpr int e = E, i = Q::ent_i(e), a = Q::ent_a(e);
		H[T::H_QLD + e] = qM[i][a];
		H[T::H_QH + e] = qH[i][a];
	});
	sfor<NV>([&](auto I) { H[T::H_QLDIAGINV + I] = dinv[I]; H[T::H_QHDI + I] = hinv[I]; H[T::H_QFRC_SMOOTH + I] = f[I]; });
	// qacc_smooth = M^-1 qfrc_smooth: L' sweep, D, L sweep
	double x[NV > 0 ? NV : 1];
	sfor<NV>([&](auto I) { x[I] = f[I]; });
	sfor<NV>([&](auto Ii) {
		constexpr int i = NV - 1 - Ii;
		sfor<NV>([&](auto A) {
			constexpr int a = A;
			if constexpr (a < i && Q::anc(a, i)) x[a] -= qM[i][a] * x[i];
		});
	});
	sfor<NV>([&](auto I) { x[I] *= dinv[I]; });
	sfor<NV>([&](auto I) {
		constexpr int i = I;
		sfor<NV>([&](auto Ai) {
			constexpr int a = NV - 1 - Ai;  // nearest ancestor first, as mj_solveLD walks them
			if constexpr (a < i && Q::anc(a, i)) x[i] -= qM[i][a] * x[a];
		});
	});
	sfor<NV>([&](auto I) { H[T::H_QACC_SMOOTH + I] = x[I]; });
}

// the stand-alone kernel's body: envs [env_lo, env_hi), 64 per wavefront, pair slots in LDS.  flags bit 0: the LAST step of the host's launch
template <class T>
DEVI void smooth_lane_env(const KernelParams MJB_AS4 *__restrict__ P, const unsigned int step, const int flags, const int env_lo, const int env_hi,
                          unsigned char *const smem)
{
	const int lane = (int)threadIdx.x;
	const int env_raw = env_lo + (int)(blockIdx.x * 64u) + lane;
	const bool live = env_raw < env_hi;
	smooth_lane_env_core<T>(P, live ? env_raw : env_hi - 1, live, step, (flags & 1) != 0, reinterpret_cast<Pair *>(smem) + lane);
}

}  // namespace mjb_sm
