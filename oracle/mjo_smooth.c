/* mjo_smooth.c — CPU oracle (TEST INFRASTRUCTURE): smooth dynamics, sensors, integration, data
 * life-cycle.  See mjo.h for provenance ("parity unpinned") and usage restrictions.
 *
 * Every function names the MuJoCo 2.3.7 engine function ([UPSTREAM], not under /root/reference)
 * whose published algorithm it restates; the reference reaches all of them only through
 * mj_step / mj_forward / mj_resetData at /root/reference mujoco_ros/src/mujoco_env.cpp:498,552,593,
 * :329,:621 and :252.
 */
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "mjo.h"
#include "mjo_math.h"

#define MJO_MAXVAL 1e10 /* mjMAXVAL */

static int dim_of(const mjb_model_desc *m, const char *rows)
{
#define MJB_SIZE(name) if (!strcmp(rows, #name)) return m->name;
#define MJB_OPT_I(name)
#define MJB_OPT_D(name, n)
#define MJB_ARR_I(name, rows, cols)
#define MJB_ARR_D(name, rows, cols)
#include "../include/mjb_model_fields.def"
#undef MJB_SIZE
#undef MJB_OPT_I
#undef MJB_OPT_D
#undef MJB_ARR_I
#undef MJB_ARR_D
	if (!strcmp(rows, "one")) return 1;
	return 0;
}

/* ------------------------------------------------------------------ data life-cycle */
mjo_data *mjo_make_data(const mjb_model_desc *m)
{
	mjo_data *d = (mjo_data *)calloc(1, sizeof(mjo_data));
	if (!d) return NULL;
#define ALLOC(name, n) d->name = calloc((size_t)((n) > 0 ? (n) : 1), sizeof(*d->name));
#define MJB_DS(name, rows, cols) ALLOC(name, dim_of(m, #rows) * (cols))
#define MJB_DD(name, rows, cols) ALLOC(name, dim_of(m, #rows) * (cols))
#define MJB_DD2(name, rows, cols) ALLOC(name, dim_of(m, #rows) * dim_of(m, #cols))
#define MJB_DI(name, rows, cols) ALLOC(name, dim_of(m, #rows) * (cols))
#include "../include/mjb_data_fields.def"
#undef MJB_DS
#undef MJB_DD
#undef MJB_DD2
#undef MJB_DI
	ALLOC(scratch_MM, 2 * m->nM + m->nv)
	ALLOC(scratch_nv, m->nv)
	ALLOC(scratch_nv2, m->nv)
	ALLOC(rk_warmstart, m->nv)
	ALLOC(rk_buf, m->nq + 5 * m->nv + m->nsensordata + 1 + 2 * m->na)
#undef ALLOC
	mjo_reset_data(m, d);
	return d;
}

int mjo_model_desc_size(void) { return (int)sizeof(mjb_model_desc); }

void mjo_free_data(mjo_data *d)
{
	if (!d) return;
#define MJB_DS(name, rows, cols) free(d->name);
#define MJB_DD(name, rows, cols) free(d->name);
#define MJB_DD2(name, rows, cols) free(d->name);
#define MJB_DI(name, rows, cols) free(d->name);
#include "../include/mjb_data_fields.def"
#undef MJB_DS
#undef MJB_DD
#undef MJB_DD2
#undef MJB_DI
	free(d->env_geom_size);
	free(d->env_geom_type);
	free(d->scratch_MM);
	free(d->scratch_nv);
	free(d->scratch_nv2);
	free(d->rk_warmstart);
	free(d->rk_buf);
	free(d);
}

double *mjo_field(const mjb_model_desc *m, mjo_data *d, int field, int *n)
{
	switch (field) {
#define MJB_DS(name, rows, cols) case MJB_F_##name: if (n) *n = dim_of(m, #rows) * (cols); return d->name;
#define MJB_DD(name, rows, cols) case MJB_F_##name: if (n) *n = dim_of(m, #rows) * (cols); return d->name;
#define MJB_DD2(name, rows, cols) case MJB_F_##name: if (n) *n = dim_of(m, #rows) * dim_of(m, #cols); return d->name;
#define MJB_DI(name, rows, cols)
#include "../include/mjb_data_fields.def"
#undef MJB_DS
#undef MJB_DD
#undef MJB_DD2
#undef MJB_DI
	default: return NULL;
	}
}

int *mjo_field_int(const mjb_model_desc *m, mjo_data *d, int field, int *n)
{
	switch (field) {
#define MJB_DS(name, rows, cols)
#define MJB_DD(name, rows, cols)
#define MJB_DD2(name, rows, cols)
#define MJB_DI(name, rows, cols) case MJB_F_##name: if (n) *n = dim_of(m, #rows) * (cols); return d->name;
#include "../include/mjb_data_fields.def"
#undef MJB_DS
#undef MJB_DD
#undef MJB_DD2
#undef MJB_DI
	default: return NULL;
	}
}

/* mj_resetData: qpos = qpos0, everything else persistent zero */
void mjo_reset_data(const mjb_model_desc *m, mjo_data *d)
{
#define MJB_DS(name, rows, cols) memset(d->name, 0, sizeof(double) * (size_t)(dim_of(m, #rows) * (cols)));
#define MJB_DD(name, rows, cols) memset(d->name, 0, sizeof(double) * (size_t)(dim_of(m, #rows) * (cols)));
#define MJB_DD2(name, rows, cols) memset(d->name, 0, sizeof(double) * (size_t)(dim_of(m, #rows) * dim_of(m, #cols)));
#define MJB_DI(name, rows, cols) memset(d->name, 0, sizeof(int) * (size_t)(dim_of(m, #rows) * (cols)));
#include "../include/mjb_data_fields.def"
#undef MJB_DS
#undef MJB_DD
#undef MJB_DD2
#undef MJB_DI
	memcpy(d->qpos, m->qpos0, sizeof(double) * (size_t)m->nq);
	for (int b = 0; b < m->nbody && m->nmocap > 0; b++) {
		int mid = m->body_mocapid[b];
		if (mid < 0) continue;
		memcpy(d->mocap_pos + 3 * mid, m->body_pos + 3 * b, 3 * sizeof(double));
		memcpy(d->mocap_quat + 4 * mid, m->body_quat + 4 * b, 4 * sizeof(double));
	}
}

/* ------------------------------------------------------------------ A1: mj_kinematics */
/* mj_local2Global */
static void local2global(const mjo_data *d, double *xpos, double *xmat, const double *pos, const double *quat,
                         int body, int sameframe)
{
	if (sameframe) {
		v3_copy(xpos, d->xpos + 3 * body);
		memcpy(xmat, d->xmat + 9 * body, 9 * sizeof(double));
	} else {
		double vec[3], q[4];
		m3_mulvec(vec, d->xmat + 9 * body, pos);
		v3_add(xpos, vec, d->xpos + 3 * body);
		q_mul(q, d->xquat + 4 * body, quat);
		q_to_mat(xmat, q);
	}
}

void mjo_kinematics(const mjb_model_desc *m, mjo_data *d)
{
	/* world */
	v3_zero(d->xpos);
	d->xquat[0] = 1; d->xquat[1] = d->xquat[2] = d->xquat[3] = 0;
	v3_zero(d->xipos);
	memset(d->xmat, 0, 9 * sizeof(double));
	memset(d->ximat, 0, 9 * sizeof(double));
	d->xmat[0] = d->xmat[4] = d->xmat[8] = 1;
	d->ximat[0] = d->ximat[4] = d->ximat[8] = 1;

	/* normalize all quaternions in qpos (this is why the reference's tests see normalised
	 * free-joint quaternions after loading initial states: ros_interface_test.cpp:342-351) */
	for (int j = 0; j < m->njnt; j++) {
		if (m->jnt_type[j] == MJB_JNT_BALL) q_normalize(d->qpos + m->jnt_qposadr[j]);
		else if (m->jnt_type[j] == MJB_JNT_FREE) q_normalize(d->qpos + m->jnt_qposadr[j] + 3);
	}

	for (int i = 1; i < m->nbody; i++) {
		double xpos[3], xquat[4];
		int pid = m->body_parentid[i];
		int jntadr = m->body_jntadr[i], jntnum = m->body_jntnum[i];
		int mid = m->nmocap > 0 ? m->body_mocapid[i] : -1;
		if (mid >= 0) {
			/* mocap body (child of the world, no joints): pose from the mocap fields, quaternion normalised */
			v3_copy(xpos, d->mocap_pos + 3 * mid);
			memcpy(xquat, d->mocap_quat + 4 * mid, 4 * sizeof(double));
			q_normalize(xquat);
		} else if (jntnum == 1 && m->jnt_type[jntadr] == MJB_JNT_FREE) {
			int qadr = m->jnt_qposadr[jntadr];
			v3_copy(xpos, d->qpos + qadr);
			memcpy(xquat, d->qpos + qadr + 3, 4 * sizeof(double));
			q_normalize(xquat);
			v3_copy(d->xanchor + 3 * jntadr, xpos);
			v3_copy(d->xaxis + 3 * jntadr, m->jnt_axis + 3 * jntadr);
		} else {
			v3_copy(xpos, m->body_pos + 3 * i);
			memcpy(xquat, m->body_quat + 4 * i, 4 * sizeof(double));
			if (pid) {
				double vec[3], q[4];
				m3_mulvec(vec, d->xmat + 9 * pid, xpos);
				v3_add(xpos, vec, d->xpos + 3 * pid);
				q_mul(q, d->xquat + 4 * pid, xquat);
				memcpy(xquat, q, sizeof q);
			}
			for (int j = jntadr; j < jntadr + jntnum; j++) {
				int qadr = m->jnt_qposadr[j];
				double *xanchor = d->xanchor + 3 * j, *xaxis = d->xaxis + 3 * j;
				q_rotvec(xaxis, m->jnt_axis + 3 * j, xquat);
				q_rotvec(xanchor, m->jnt_pos + 3 * j, xquat);
				v3_addto(xanchor, xpos);
				if (m->jnt_type[j] == MJB_JNT_SLIDE) {
					v3_addtoscl(xpos, xaxis, d->qpos[qadr] - m->qpos0[qadr]);
				} else { /* ball or hinge */
					double qloc[4], q[4], vec[3];
					if (m->jnt_type[j] == MJB_JNT_BALL) {
						memcpy(qloc, d->qpos + qadr, sizeof qloc);
						q_normalize(qloc);
					} else {
						q_axis_angle(qloc, m->jnt_axis + 3 * j, d->qpos[qadr] - m->qpos0[qadr]);
					}
					q_mul(q, xquat, qloc);
					memcpy(xquat, q, sizeof q);
					/* correct for off-centre rotation */
					q_rotvec(vec, m->jnt_pos + 3 * j, xquat);
					v3_sub(xpos, xanchor, vec);
				}
			}
		}
		q_normalize(xquat);
		v3_copy(d->xpos + 3 * i, xpos);
		memcpy(d->xquat + 4 * i, xquat, 4 * sizeof(double));
		q_to_mat(d->xmat + 9 * i, xquat);
	}
	for (int i = 1; i < m->nbody; i++)
		local2global(d, d->xipos + 3 * i, d->ximat + 9 * i, m->body_ipos + 3 * i, m->body_iquat + 4 * i, i,
		             m->body_sameframe[i]);
	for (int i = 0; i < m->ngeom; i++)
		local2global(d, d->geom_xpos + 3 * i, d->geom_xmat + 9 * i, m->geom_pos + 3 * i, m->geom_quat + 4 * i,
		             m->geom_bodyid[i], m->geom_sameframe[i]);
	for (int i = 0; i < m->nsite; i++) {
		local2global(d, d->site_xpos + 3 * i, d->site_xmat + 9 * i, m->site_pos + 3 * i, m->site_quat + 4 * i,
		             m->site_bodyid[i], m->site_sameframe[i]);
		q_mul(d->site_xquat + 4 * i, d->xquat + 4 * m->site_bodyid[i], m->site_quat + 4 * i); /* engine-side field */
	}
}

/* ------------------------------------------------------------------ A1: mj_comPos */
void mjo_com_pos(const mjb_model_desc *m, mjo_data *d)
{
	memset(d->subtree_com, 0, sizeof(double) * 3 * (size_t)m->nbody);
	for (int i = m->nbody - 1; i >= 0; i--) {
		v3_addtoscl(d->subtree_com + 3 * i, d->xipos + 3 * i, m->body_mass[i]);
		if (i) v3_addto(d->subtree_com + 3 * m->body_parentid[i], d->subtree_com + 3 * i);
		if (m->body_subtreemass[i] < MJO_MINVAL) v3_copy(d->subtree_com + 3 * i, d->xipos + 3 * i);
		else v3_scl(d->subtree_com + 3 * i, d->subtree_com + 3 * i, 1.0 / fmax(MJO_MINVAL, m->body_subtreemass[i]));
	}
	/* body inertias about the subtree com of the kinematic-tree root */
	memset(d->cinert, 0, sizeof(double) * 10);
	for (int i = 1; i < m->nbody; i++) {
		double offset[3];
		v3_sub(offset, d->xipos + 3 * i, d->subtree_com + 3 * m->body_rootid[i]);
		inert_com(d->cinert + 10 * i, m->body_inertia + 3 * i, d->ximat + 9 * i, offset, m->body_mass[i]);
	}
	/* motion dofs in the same frame */
	for (int j = 0; j < m->njnt; j++) {
		int bi = m->jnt_bodyid[j], da = 6 * m->jnt_dofadr[j];
		double offset[3];
		v3_sub(offset, d->subtree_com + 3 * m->body_rootid[bi], d->xanchor + 3 * j);
		int skip = 0;
		switch (m->jnt_type[j]) {
		case MJB_JNT_FREE:
			/* translations */
			memset(d->cdof + da, 0, 18 * sizeof(double));
			for (int k = 0; k < 3; k++) d->cdof[da + 3 + 7 * k] = 1;
			skip = 18;
			/* fall through: rotations like a ball joint */
		case MJB_JNT_BALL:
			for (int k = 0; k < 3; k++) {
				double axis[3] = { d->xmat[9 * bi + k], d->xmat[9 * bi + k + 3], d->xmat[9 * bi + k + 6] };
				dof_com(d->cdof + da + skip + 6 * k, axis, offset);
			}
			break;
		case MJB_JNT_SLIDE:
			dof_com(d->cdof + da, d->xaxis + 3 * j, NULL);
			break;
		default: /* hinge */
			dof_com(d->cdof + da, d->xaxis + 3 * j, offset);
		}
	}
}

/* ------------------------------------------------------------------ A2: mj_crb */
void mjo_crb(const mjb_model_desc *m, mjo_data *d)
{
	memcpy(d->crb, d->cinert, sizeof(double) * 10 * (size_t)m->nbody);
	for (int i = m->nbody - 1; i > 0; i--) {
		int p = m->body_parentid[i];
		if (p > 0)
			for (int k = 0; k < 10; k++) d->crb[10 * p + k] += d->crb[10 * i + k];
	}
	memset(d->qM, 0, sizeof(double) * (size_t)m->nM);
	for (int i = 0; i < m->nv; i++) {
		int adr = m->dof_Madr[i];
		double buf[6];
		d->qM[adr] = m->dof_armature[i];
		mul_inert_vec(buf, d->crb + 10 * m->dof_bodyid[i], d->cdof + 6 * i);
		for (int j = i; j >= 0; j = m->dof_parentid[j]) d->qM[adr++] += dot6(d->cdof + 6 * j, buf);
	}
}

/* mj_factorI: sparse L'*D*L of a matrix in qM layout */
static void factor_i(const mjb_model_desc *m, const double *M, double *qLD, double *qLDiagInv)
{
	memcpy(qLD, M, sizeof(double) * (size_t)m->nM);
	for (int k = m->nv - 1; k >= 0; k--) {
		int Madr_kk = m->dof_Madr[k], Madr_ki = Madr_kk + 1;
		int i = m->dof_parentid[k];
		while (i >= 0) {
			double tmp = qLD[Madr_ki] / qLD[Madr_kk];
			int Madr_ij = m->dof_Madr[i], Madr_kj = Madr_ki;
			for (int j = i; j >= 0; j = m->dof_parentid[j]) qLD[Madr_ij++] -= tmp * qLD[Madr_kj++];
			qLD[Madr_ki] = tmp;
			i = m->dof_parentid[i];
			Madr_ki++;
		}
	}
	for (int i = 0; i < m->nv; i++) qLDiagInv[i] = 1.0 / qLD[m->dof_Madr[i]];
}

/* mj_solveLD (n = 1) */
static void solve_ld(const mjb_model_desc *m, double *x, const double *qLD, const double *qLDiagInv)
{
	for (int i = m->nv - 1; i >= 0; i--) {
		if (x[i] != 0) {
			int adr = m->dof_Madr[i] + 1;
			for (int j = m->dof_parentid[i]; j >= 0; j = m->dof_parentid[j]) x[j] -= qLD[adr++] * x[i];
		}
	}
	for (int i = 0; i < m->nv; i++) x[i] *= qLDiagInv[i];
	for (int i = 0; i < m->nv; i++) {
		int adr = m->dof_Madr[i] + 1;
		for (int j = m->dof_parentid[i]; j >= 0; j = m->dof_parentid[j]) x[i] -= qLD[adr++] * x[j];
	}
}

/* A3: mj_factorM */
void mjo_factor_m(const mjb_model_desc *m, mjo_data *d) { factor_i(m, d->qM, d->qLD, d->qLDiagInv); }
void mjo_solve_m(const mjb_model_desc *m, mjo_data *d, double *x) { solve_ld(m, x, d->qLD, d->qLDiagInv); }

/* mj_transmission, joint transmission on hinge/slide joints only */
void mjo_transmission(const mjb_model_desc *m, mjo_data *d)
{
	for (int i = 0; i < m->nu; i++) {
		int j = m->actuator_trnid[2 * i];
		if (m->actuator_trntype[i] == MJB_TRN_TENDON) d->actuator_length[i] = d->ten_length[j] * m->actuator_gear[6 * i];  /* (mj_tendon ran before) */
		else d->actuator_length[i] = d->qpos[m->jnt_qposadr[j]] * m->actuator_gear[6 * i];
	}
}

/* ------------------------------------------------------------------ A8: mj_comVel */
void mjo_com_vel(const mjb_model_desc *m, mjo_data *d)
{
	memset(d->cvel, 0, 6 * sizeof(double));
	for (int i = 1; i < m->nbody; i++) {
		double cvel[6], tmp[6];
		int bda = m->body_dofadr[i];
		memcpy(cvel, d->cvel + 6 * m->body_parentid[i], sizeof cvel);
		int j = 0;
		for (int jj = 0; jj < m->body_jntnum[i]; jj++) {
			int type = m->jnt_type[m->body_jntadr[i] + jj];
			double *cdof = d->cdof + 6 * (bda + j), *cdofdot = d->cdof_dot + 6 * (bda + j);
			if (type == MJB_JNT_FREE) {
				memset(cdofdot, 0, 18 * sizeof(double));
				mul_dof_vec(tmp, cdof, d->qvel + bda + j, 3);
				for (int k = 0; k < 6; k++) cvel[k] += tmp[k];
				j += 3;
				cdof += 18;
				cdofdot += 18;
				type = MJB_JNT_BALL;
			}
			if (type == MJB_JNT_BALL) {
				for (int k = 0; k < 3; k++) cross_motion(cdofdot + 6 * k, cvel, cdof + 6 * k);
				mul_dof_vec(tmp, cdof, d->qvel + bda + j, 3);
				for (int k = 0; k < 6; k++) cvel[k] += tmp[k];
				j += 3;
			} else {
				cross_motion(cdofdot, cvel, cdof);
				mul_dof_vec(tmp, cdof, d->qvel + bda + j, 1);
				for (int k = 0; k < 6; k++) cvel[k] += tmp[k];
				j += 1;
			}
		}
		memcpy(d->cvel + 6 * i, cvel, sizeof cvel);
	}
	/* actuator_velocity = actuator_moment . qvel; a tendon transmission's moment is gear * the tendon's coefficients */
	for (int i = 0; i < m->nu; i++) {
		int jn = m->actuator_trnid[2 * i];
		if (m->actuator_trntype[i] == MJB_TRN_TENDON) {
			double v = 0;
			for (int w = m->tendon_adr[jn]; w < m->tendon_adr[jn] + m->tendon_num[jn]; w++)
				v += m->actuator_gear[6 * i] * m->wrap_prm[w] * d->qvel[m->jnt_dofadr[m->wrap_objid[w]]];
			d->actuator_velocity[i] = v;
		} else
			d->actuator_velocity[i] = m->actuator_gear[6 * i] * d->qvel[m->jnt_dofadr[jn]];
	}
}

/* ------------------------------------------------------------------ A8: mj_passive (springs + dampers) */
void mjo_passive(const mjb_model_desc *m, mjo_data *d)
{
	memset(d->qfrc_passive, 0, sizeof(double) * (size_t)m->nv);
	if (m->disableflags & MJB_DSBL_PASSIVE) return;
	for (int j = 0; j < m->njnt; j++) {
		double k = m->jnt_stiffness[j];
		if (k == 0) continue;
		int pa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
		switch (m->jnt_type[j]) {
		case MJB_JNT_FREE:
			for (int c = 0; c < 3; c++) d->qfrc_passive[da + c] = -k * (d->qpos[pa + c] - m->qpos_spring[pa + c]);
			pa += 3;
			da += 3;
			/* fall through */
		case MJB_JNT_BALL: {
			double q[4], dif[3];
			memcpy(q, d->qpos + pa, sizeof q);
			q_normalize(q);
			q_sub(dif, q, m->qpos_spring + pa);
			for (int c = 0; c < 3; c++) d->qfrc_passive[da + c] = -k * dif[c];
			break;
		}
		default:
			d->qfrc_passive[da] = -k * (d->qpos[pa] - m->qpos_spring[pa]);
		}
	}
	for (int i = 0; i < m->nv; i++) d->qfrc_passive[i] -= m->dof_damping[i] * d->qvel[i];
	/* tendon springs and dampers: qfrc_passive += J' (-k (L - L_spring) - b v) */
	for (int t = 0; t < m->ntendon; t++) {
		double frc = -m->tendon_stiffness[t] * (d->ten_length[t] - m->tendon_lengthspring[t]) -
		             m->tendon_damping[t] * d->ten_velocity[t];
		if (frc == 0) continue;
		for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++)
			d->qfrc_passive[m->jnt_dofadr[m->wrap_objid[w]]] += m->wrap_prm[w] * frc;
	}
}

/* mj_tendon for fixed tendons: length = sum coef * qpos[joint]; mj_fwdVelocity: velocity = J qvel */
void mjo_tendon(const mjb_model_desc *m, mjo_data *d)
{
	for (int t = 0; t < m->ntendon; t++) {
		double len = 0;
		for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++)
			len += m->wrap_prm[w] * d->qpos[m->jnt_qposadr[m->wrap_objid[w]]];
		d->ten_length[t] = len;
	}
}

void mjo_tendon_vel(const mjb_model_desc *m, mjo_data *d)
{
	for (int t = 0; t < m->ntendon; t++) {
		double v = 0;
		for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++)
			v += m->wrap_prm[w] * d->qvel[m->jnt_dofadr[m->wrap_objid[w]]];
		d->ten_velocity[t] = v;
	}
}

/* ------------------------------------------------------------------ A9: mj_rne (flg_acc = 0) */
void mjo_rne(const mjb_model_desc *m, mjo_data *d)
{
	double tmp[6], tmp1[6];
	memset(d->cacc, 0, 6 * sizeof(double));
	if (!(m->disableflags & MJB_DSBL_GRAVITY))
		for (int k = 0; k < 3; k++) d->cacc[3 + k] = -m->gravity[k];
	for (int i = 1; i < m->nbody; i++) {
		int bda = m->body_dofadr[i];
		mul_dof_vec(tmp, d->cdof_dot + 6 * bda, d->qvel + bda, m->body_dofnum[i]);
		for (int k = 0; k < 6; k++) d->cacc[6 * i + k] = d->cacc[6 * m->body_parentid[i] + k] + tmp[k];
		mul_inert_vec(d->cfrc_body + 6 * i, d->cinert + 10 * i, d->cacc + 6 * i);
		mul_inert_vec(tmp, d->cinert + 10 * i, d->cvel + 6 * i);
		cross_force(tmp1, d->cvel + 6 * i, tmp);
		for (int k = 0; k < 6; k++) d->cfrc_body[6 * i + k] += tmp1[k];
	}
	memset(d->cfrc_body, 0, 6 * sizeof(double));
	for (int i = m->nbody - 1; i > 0; i--)
		if (m->body_parentid[i])
			for (int k = 0; k < 6; k++) d->cfrc_body[6 * m->body_parentid[i] + k] += d->cfrc_body[6 * i + k];
	for (int i = 0; i < m->nv; i++) d->qfrc_bias[i] = dot6(d->cdof + 6 * i, d->cfrc_body + 6 * m->dof_bodyid[i]);
}

/* ------------------------------------------------------------------ A12: mj_fwdActuation */
void mjo_fwd_actuation(const mjb_model_desc *m, mjo_data *d)
{
	memset(d->qfrc_actuator, 0, sizeof(double) * (size_t)m->nv);
	if (m->nu == 0 || (m->disableflags & MJB_DSBL_ACTUATION)) {
		memset(d->actuator_force, 0, sizeof(double) * (size_t)m->nu);
		memset(d->act_dot, 0, sizeof(double) * (size_t)m->na);  /* (mj_fwdActuation clears act_dot before it returns early) */
		return;
	}
	for (int i = 0; i < m->nu; i++) {
		double ctrl = d->ctrl[i];
		if (m->actuator_ctrllimited[i] && !(m->disableflags & MJB_DSBL_CLAMPCTRL)) {
			const double *r = m->actuator_ctrlrange + 2 * i;
			ctrl = ctrl < r[0] ? r[0] : (ctrl > r[1] ? r[1] : ctrl);
		}
		const double *gp = m->actuator_gainprm + 3 * i, *bp = m->actuator_biasprm + 3 * i;
		double gain = gp[0], bias = 0;
		if (m->actuator_gaintype[i] == MJB_GAIN_AFFINE)
			gain = gp[0] + gp[1] * d->actuator_length[i] + gp[2] * d->actuator_velocity[i];
		if (m->actuator_biastype[i] == MJB_BIAS_AFFINE)
			bias = bp[0] + bp[1] * d->actuator_length[i] + bp[2] * d->actuator_velocity[i];
		/* stateful actuators (mj_fwdActuation): act_dot from the clamped ctrl, and the gain multiplies the ACTIVATION */
		double input = ctrl;
		const int ja = m->na > 0 ? m->actuator_actadr[i] : -1;
		if (ja >= 0) {
			if (m->actuator_dyntype[i] == MJB_DYN_INTEGRATOR) d->act_dot[ja] = ctrl;
			else {
				const double tau = m->actuator_dynprm[3 * i] > MJO_MINVAL ? m->actuator_dynprm[3 * i] : MJO_MINVAL;
				d->act_dot[ja] = (ctrl - d->act[ja]) / tau;
			}
			input = d->act[ja];
		}
		double force = gain * input + bias;
		if (m->actuator_forcelimited[i]) {
			const double *r = m->actuator_forcerange + 2 * i;
			force = force < r[0] ? r[0] : (force > r[1] ? r[1] : force);
		}
		d->actuator_force[i] = force;
	}
	/* qfrc_actuator = moment' * force */
	for (int i = 0; i < m->nu; i++) {
		int j = m->actuator_trnid[2 * i];
		if (m->actuator_trntype[i] == MJB_TRN_TENDON) {
			for (int w = m->tendon_adr[j]; w < m->tendon_adr[j] + m->tendon_num[j]; w++)
				d->qfrc_actuator[m->jnt_dofadr[m->wrap_objid[w]]] += m->actuator_gear[6 * i] * m->wrap_prm[w] * d->actuator_force[i];
		} else
			d->qfrc_actuator[m->jnt_dofadr[j]] += m->actuator_gear[6 * i] * d->actuator_force[i];
	}
}

/* mj_xfrcAccumulate: Cartesian force/torque at the body com projected with the cdof Jacobian */
static void xfrc_accumulate(const mjb_model_desc *m, mjo_data *d, double *qfrc)
{
	for (int b = 1; b < m->nbody; b++) {
		const double *x = d->xfrc_applied + 6 * b;
		if (x[0] == 0 && x[1] == 0 && x[2] == 0 && x[3] == 0 && x[4] == 0 && x[5] == 0) continue;
		double offset[3];
		v3_sub(offset, d->xipos + 3 * b, d->subtree_com + 3 * m->body_rootid[b]);
		/* walk the dofs affecting body b */
		int bb = b;
		while (bb > 0 && m->body_dofnum[bb] == 0) bb = m->body_parentid[bb];
		if (bb == 0) continue;
		for (int i = m->body_dofadr[bb] + m->body_dofnum[bb] - 1; i >= 0; i = m->dof_parentid[i]) {
			const double *cd = d->cdof + 6 * i;
			double jp[3];
			v3_cross(jp, cd, offset);
			v3_addto(jp, cd + 3);
			qfrc[i] += v3_dot(jp, x) + v3_dot(cd, x + 3);
		}
	}
}

/* A12: mj_fwdAcceleration */
void mjo_fwd_acceleration(const mjb_model_desc *m, mjo_data *d)
{
	for (int i = 0; i < m->nv; i++) {
		d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i];
		d->qfrc_smooth[i] += d->qfrc_applied[i];
		d->qfrc_smooth[i] += d->qfrc_actuator[i];
	}
	xfrc_accumulate(m, d, d->qfrc_smooth);
	memcpy(d->qacc_smooth, d->qfrc_smooth, sizeof(double) * (size_t)m->nv);
	solve_ld(m, d->qacc_smooth, d->qLD, d->qLDiagInv);
}

/* ------------------------------------------------------------------ A15: sensors */
static void frame_of(const mjb_model_desc *m, const mjo_data *d, int objtype, int id, const double **pos,
                     const double **mat, double *quat)
{
	switch (objtype) {
	case MJB_OBJ_BODY:
		*pos = d->xipos + 3 * id; *mat = d->ximat + 9 * id;
		q_mul(quat, d->xquat + 4 * id, m->body_iquat + 4 * id);
		break;
	case MJB_OBJ_XBODY:
		*pos = d->xpos + 3 * id; *mat = d->xmat + 9 * id;
		memcpy(quat, d->xquat + 4 * id, 4 * sizeof(double));
		break;
	case MJB_OBJ_GEOM:
		*pos = d->geom_xpos + 3 * id; *mat = d->geom_xmat + 9 * id;
		q_mul(quat, d->xquat + 4 * m->geom_bodyid[id], m->geom_quat + 4 * id);
		break;
	default: /* site */
		*pos = d->site_xpos + 3 * id; *mat = d->site_xmat + 9 * id;
		q_mul(quat, d->xquat + 4 * m->site_bodyid[id], m->site_quat + 4 * id);
	}
}

/* mj_objectVelocity */
static void object_velocity(const mjb_model_desc *m, const mjo_data *d, int objtype, int id, double *res, int local)
{
	const double *pos, *mat;
	double q[4];
	int body = objtype == MJB_OBJ_GEOM ? m->geom_bodyid[id] : (objtype == MJB_OBJ_SITE ? m->site_bodyid[id] : id);
	frame_of(m, d, objtype, id, &pos, &mat, q);
	transform_spatial_motion(res, d->cvel + 6 * body, pos, d->subtree_com + 3 * m->body_rootid[body],
	                         local ? mat : NULL);
}

/* mj_contactForce: contact force / torque in the contact frame (normal first) from the solver's row forces */
static void contact_force(const mjb_model_desc *m, const mjo_data *d, int c, double *res)
{
	for (int k = 0; k < 6; k++) res[k] = 0;
	int adr = d->contact_efc_address[c], dim = d->contact_dim[c];
	if (adr < 0) return;
	if (dim == 1) {
		res[0] = d->efc_force[adr];
	} else if (m->cone == MJB_CONE_ELLIPTIC) {
		for (int k = 0; k < dim; k++) res[k] = d->efc_force[adr + k];
	} else {
		const double *fri = d->contact_friction + 5 * c;
		for (int k = 0; k < 2 * (dim - 1); k++) res[0] += d->efc_force[adr + k];
		for (int k = 0; k < dim - 1; k++) res[1 + k] = (d->efc_force[adr + 2 * k] - d->efc_force[adr + 2 * k + 1]) * fri[k];
	}
}

/* rows of this contact exist?  (contact_efc_address is only meaningful for contacts that made it into the rows) */
static int contact_active(const mjb_model_desc *m, const mjo_data *d, int c)
{
	if (!(d->contact_dist[c] < d->contact_includemargin[c])) return 0;
	int adr = d->contact_efc_address[c], dim = d->contact_dim[c];
	int nrow = dim == 1 ? 1 : (m->cone == MJB_CONE_ELLIPTIC ? dim : 2 * (dim - 1));
	return adr >= 0 && adr + nrow <= d->nefc[0] && (d->efc_type[adr] >= MJB_CNSTR_CONTACT_FRICTIONLESS) && d->efc_id[adr] == c;
}

/* mj_rnePostConstraint (restated from memory of MuJoCo 2.3.7 engine_core_smooth.c): body accelerations cacc with
 * qacc included, external forces cfrc_ext (xfrc_applied + contact forces) and the interaction force cfrc_int each
 * body exchanges with its parent, all as spatial vectors (rotation first) about the subtree com of the tree root */
void mjo_rne_post_constraint(const mjb_model_desc *m, mjo_data *d)
{
	int nb = m->nbody;
	memset(d->cfrc_ext, 0, sizeof(double) * 6 * (size_t)nb);
	for (int i = 1; i < nb; i++) {
		const double *x = d->xfrc_applied + 6 * i;
		if (x[0] == 0 && x[1] == 0 && x[2] == 0 && x[3] == 0 && x[4] == 0 && x[5] == 0) continue;
		double cf[6] = { x[3], x[4], x[5], x[0], x[1], x[2] }, r[6];
		transform_spatial_force(r, cf, d->subtree_com + 3 * m->body_rootid[i], d->xipos + 3 * i, NULL);
		for (int k = 0; k < 6; k++) d->cfrc_ext[6 * i + k] += r[k];
	}
	for (int c = 0; c < d->ncon[0]; c++) {
		if (!contact_active(m, d, c)) continue;
		double lf[6], cf[6], r[6];
		contact_force(m, d, c, lf);
		m3_mulvecT(cf + 3, d->contact_frame + 9 * c, lf);   /* force  = frame' * lf[0:3] */
		m3_mulvecT(cf, d->contact_frame + 9 * c, lf + 3);   /* torque = frame' * lf[3:6] */
		int b1 = m->geom_bodyid[d->contact_geom[2 * c]], b2 = m->geom_bodyid[d->contact_geom[2 * c + 1]];
		if (b1) {
			transform_spatial_force(r, cf, d->subtree_com + 3 * m->body_rootid[b1], d->contact_pos + 3 * c, NULL);
			for (int k = 0; k < 6; k++) d->cfrc_ext[6 * b1 + k] -= r[k];
		}
		if (b2) {
			transform_spatial_force(r, cf, d->subtree_com + 3 * m->body_rootid[b2], d->contact_pos + 3 * c, NULL);
			for (int k = 0; k < 6; k++) d->cfrc_ext[6 * b2 + k] += r[k];
		}
	}
	memset(d->cacc, 0, 6 * sizeof(double));
	if (!(m->disableflags & MJB_DSBL_GRAVITY))
		for (int k = 0; k < 3; k++) d->cacc[3 + k] = -m->gravity[k];
	memset(d->cfrc_int, 0, 6 * sizeof(double));
	for (int i = 1; i < nb; i++) {
		int bda = m->body_dofadr[i], nd = m->body_dofnum[i];
		double t1[6], t2[6], fb[6];
		mul_dof_vec(t1, d->cdof_dot + 6 * bda, d->qvel + bda, nd);
		mul_dof_vec(t2, d->cdof + 6 * bda, d->qacc + bda, nd);
		for (int k = 0; k < 6; k++) d->cacc[6 * i + k] = d->cacc[6 * m->body_parentid[i] + k] + t1[k] + t2[k];
		mul_inert_vec(fb, d->cinert + 10 * i, d->cacc + 6 * i);
		mul_inert_vec(t1, d->cinert + 10 * i, d->cvel + 6 * i);
		cross_force(t2, d->cvel + 6 * i, t1);
		for (int k = 0; k < 6; k++) d->cfrc_int[6 * i + k] = fb[k] + t2[k] - d->cfrc_ext[6 * i + k];
	}
	for (int i = nb - 1; i > 0; i--)
		for (int k = 0; k < 6; k++) d->cfrc_int[6 * m->body_parentid[i] + k] += d->cfrc_int[6 * i + k];
}

int mjo_needs_rne_post(const mjb_model_desc *m)
{
	for (int i = 0; i < m->nsensor; i++) {
		int t = m->sensor_type[i];
		if (t == MJB_SENS_TOUCH || t == MJB_SENS_ACCELEROMETER || t == MJB_SENS_FORCE || t == MJB_SENS_TORQUE ||
		    t == MJB_SENS_FRAMELINACC || t == MJB_SENS_FRAMEANGACC)
			return 1;
	}
	return 0;
}

/* mj_objectAcceleration */
static void object_acceleration(const mjb_model_desc *m, const mjo_data *d, int objtype, int id, double *res, int local)
{
	const double *pos, *mat;
	double q[4], vel[6], cr[3];
	int body = objtype == MJB_OBJ_GEOM ? m->geom_bodyid[id] : (objtype == MJB_OBJ_SITE ? m->site_bodyid[id] : id);
	frame_of(m, d, objtype, id, &pos, &mat, q);
	const double *com = d->subtree_com + 3 * m->body_rootid[body];
	transform_spatial_motion(res, d->cacc + 6 * body, pos, com, local ? mat : NULL);
	transform_spatial_motion(vel, d->cvel + 6 * body, pos, com, local ? mat : NULL);
	v3_cross(cr, vel, vel + 3);   /* rotating-frame correction: omega x v */
	v3_addto(res + 3, cr);
}

/* does the ray p + s dir (s >= 0) meet the site volume?  sphere / box sites (mju_rayGeom restricted to what the touch
 * sensor needs: a hit / no-hit answer) */
static int ray_hits_site(const mjb_model_desc *m, const mjo_data *d, int site, const double *p, const double *dir)
{
	double rel[3], lp[3], ld[3];
	v3_sub(rel, p, d->site_xpos + 3 * site);
	m3_mulvecT(lp, d->site_xmat + 9 * site, rel);
	m3_mulvecT(ld, d->site_xmat + 9 * site, dir);
	const double *sz = m->site_size + 3 * site;
	if (m->site_type[site] == MJB_GEOM_SPHERE) {
		double b = v3_dot(lp, ld), c = v3_dot(lp, lp) - sz[0] * sz[0], a = v3_dot(ld, ld);
		if (c <= 0) return 1;              /* origin inside */
		double det = b * b - a * c;
		return det >= 0 && -b + sqrt(det) >= 0 && a > 0;
	}
	/* box: slab test */
	double tmin = 0, tmax = 1e300;
	for (int k = 0; k < 3; k++) {
		if (fabs(ld[k]) < MJO_MINVAL) {
			if (fabs(lp[k]) > sz[k]) return 0;
			continue;
		}
		double t1 = (-sz[k] - lp[k]) / ld[k], t2 = (sz[k] - lp[k]) / ld[k];
		if (t1 > t2) { double sw = t1; t1 = t2; t2 = sw; }
		if (t1 > tmin) tmin = t1;
		if (t2 < tmax) tmax = t2;
		if (tmin > tmax) return 0;
	}
	return 1;
}

/* ---- mj_ray / mju_rayGeom (engine_ray.c) for the primitives of the engine: distance along the unit-less ray pnt + x vec to the nearest geom, -1 for none.
 * ray_quad: the smaller non-negative root of a x^2 + 2 b x + c = 0 (both roots in xx). */
static double ray_quad(double a, double b, double c, double *xx)
{
	double det = b * b - a * c;
	if (det < MJO_MINVAL) { xx[0] = xx[1] = -1; return -1; }
	det = sqrt(det);
	xx[0] = (-b - det) / a;
	xx[1] = (-b + det) / a;
	return xx[0] >= 0 ? xx[0] : (xx[1] >= 0 ? xx[1] : -1);
}

static double ray_geom(const double *pos, const double *mat, const double *size, const double *pnt, const double *vec, int type)
{
	double dif[3], lp[3], lv[3], xx[2];
	v3_sub(dif, pnt, pos);
	m3_mulvecT(lp, mat, dif);
	m3_mulvecT(lv, mat, vec);
	switch (type) {
	case MJB_GEOM_PLANE: {
		if (lv[2] > -MJO_MINVAL) return -1; /* only from the front, never parallel */
		double x = -lp[2] / lv[2];
		if (x < 0) return -1;
		double p0 = lp[0] + x * lv[0], p1 = lp[1] + x * lv[1];
		if ((size[0] <= 0 || fabs(p0) <= size[0]) && (size[1] <= 0 || fabs(p1) <= size[1])) return x;
		return -1;
	}
	case MJB_GEOM_SPHERE: return ray_quad(v3_dot(lv, lv), v3_dot(lv, lp), v3_dot(lp, lp) - size[0] * size[0], xx);
	case MJB_GEOM_CAPSULE: {
		double x = -1;
		double sol = ray_quad(lv[0] * lv[0] + lv[1] * lv[1], lv[0] * lp[0] + lv[1] * lp[1], lp[0] * lp[0] + lp[1] * lp[1] - size[0] * size[0], xx);
		if (sol >= 0 && fabs(lp[2] + sol * lv[2]) <= size[1]) x = sol;
		for (int side = 1; side >= -1; side -= 2) { /* top cap, then bottom cap: the half of the sphere beyond the flat side */
			double ld[3] = { lp[0], lp[1], lp[2] - side * size[1] };
			ray_quad(v3_dot(lv, lv), v3_dot(lv, ld), v3_dot(ld, ld) - size[0] * size[0], xx);
			for (int i = 0; i < 2; i++)
				if (xx[i] >= 0 && side * (lp[2] + xx[i] * lv[2]) >= size[1] && (x < 0 || xx[i] < x)) x = xx[i];
		}
		return x;
	}
	case MJB_GEOM_BOX: {
		double x = -1;
		for (int i = 0; i < 3; i++) {
			if (fabs(lv[i]) <= MJO_MINVAL) continue;
			for (int side = -1; side <= 1; side += 2) {
				double sol = (side * size[i] - lp[i]) / lv[i];
				if (sol < 0) continue;
				int j = (i + 1) % 3, k = (i + 2) % 3;
				if (fabs(lp[j] + sol * lv[j]) <= size[j] && fabs(lp[k] + sol * lv[k]) <= size[k] && (x < 0 || sol < x)) x = sol;
			}
		}
		return x;
	}
	default: return -1;
	}
}

/* mj_ray(m, d, pnt, vec, geomgroup = NULL, flg_static = 1, bodyexclude, NULL): geoms of bodyexclude and invisible geoms (alpha 0) are skipped */
static double ray_all(const mjb_model_desc *m, const mjo_data *d, const double *pnt, const double *vec, int bodyexclude)
{
	double dist = -1;
	const double *gsz = d->env_geom_size ? d->env_geom_size : m->geom_size;
	for (int g = 0; g < m->ngeom; g++) {
		if (m->geom_bodyid[g] == bodyexclude || m->geom_rgba[4 * g + 3] == 0) continue;
		int type = d->env_geom_type ? d->env_geom_type[g] : m->geom_type[g];
		double x = ray_geom(d->geom_xpos + 3 * g, d->geom_xmat + 9 * g, gsz + 3 * g, pnt, vec, type);
		if (x >= 0 && (x < dist || dist < 0)) dist = x;
	}
	return dist;
}

/* mj_subtreeVel (engine_core_smooth.c): linear velocity and angular momentum of every subtree -- body momenta from cvel (taken at the root's
 * subtree com), linear momenta summed leaf to root and divided by the subtree mass, then the angular momenta about each subtree's own com, leaf to
 * root, with the two transport terms (body com -> subtree com, child subtree -> parent subtree).  Returns subtree `id`'s pair. */
void mjo_subtree_vel(const mjb_model_desc *m, const mjo_data *d, int id, double *linvel, double *angmom)
{
	int nb = m->nbody;
	double *bodyvel = (double *)malloc(sizeof(double) * 6 * (size_t)nb), *sl = (double *)malloc(sizeof(double) * 3 * (size_t)nb),
	       *sa = (double *)malloc(sizeof(double) * 3 * (size_t)nb);
	for (int i = 0; i < nb; i++) {
		double dif[3], dv[3], tmp[3], w[3];
		v3_sub(dif, d->xipos + 3 * i, d->subtree_com + 3 * m->body_rootid[i]);
		v3_cross(dv, d->cvel + 6 * i, dif);
		v3_copy(bodyvel + 6 * i, d->cvel + 6 * i);
		v3_add(bodyvel + 6 * i + 3, d->cvel + 6 * i + 3, dv);
		v3_scl(sl + 3 * i, bodyvel + 6 * i + 3, m->body_mass[i]);
		/* body angular momentum: ximat diag(inertia) ximat' w */
		m3_mulvecT(tmp, d->ximat + 9 * i, bodyvel + 6 * i);
		for (int k = 0; k < 3; k++) w[k] = tmp[k] * m->body_inertia[3 * i + k];
		m3_mulvec(sa + 3 * i, d->ximat + 9 * i, w);
	}
	for (int i = nb - 1; i >= 0; i--) {
		if (i) v3_addto(sl + 3 * m->body_parentid[i], sl + 3 * i);
		v3_scl(sl + 3 * i, sl + 3 * i, 1.0 / fmax(MJO_MINVAL, m->body_subtreemass[i]));
	}
	for (int i = nb - 1; i > 0; i--) {
		int p = m->body_parentid[i];
		double dx[3], dv[3], dp[3], dL[3];
		/* momentum of the body about its subtree's com */
		v3_sub(dx, d->xipos + 3 * i, d->subtree_com + 3 * i);
		v3_sub(dv, bodyvel + 6 * i + 3, sl + 3 * i);
		v3_scl(dp, dv, m->body_mass[i]);
		v3_cross(dL, dx, dp);
		v3_addto(sa + 3 * i, dL);
		/* to the parent */
		v3_addto(sa + 3 * p, sa + 3 * i);
		v3_sub(dx, d->subtree_com + 3 * i, d->subtree_com + 3 * p);
		v3_sub(dv, sl + 3 * i, sl + 3 * p);
		v3_scl(dv, dv, m->body_subtreemass[i]);
		v3_cross(dL, dx, dv);
		v3_addto(sa + 3 * p, dL);
	}
	v3_copy(linvel, sl + 3 * id);
	v3_copy(angmom, sa + 3 * id);
	free(bodyvel); free(sl); free(sa);
}

void mjo_sensor(const mjb_model_desc *m, mjo_data *d, int stage)
{
	if (m->disableflags & MJB_DSBL_SENSOR) return;
	for (int i = 0; i < m->nsensor; i++) {
		if (m->sensor_needstage[i] != stage) continue;
		int type = m->sensor_type[i], id = m->sensor_objid[i], ot = m->sensor_objtype[i];
		int rid = m->sensor_refid[i], rt = m->sensor_reftype[i];
		double *out = d->sensordata + m->sensor_adr[i];
		int is_real = 1;
		switch (type) {
		case MJB_SENS_JOINTPOS: out[0] = d->qpos[m->jnt_qposadr[id]]; break;
		case MJB_SENS_ACTUATORPOS: out[0] = d->actuator_length[id]; break;
		case MJB_SENS_BALLQUAT:
			memcpy(out, d->qpos + m->jnt_qposadr[id], 4 * sizeof(double));
			q_normalize(out);
			is_real = 0;
			break;
		case MJB_SENS_FRAMEPOS: case MJB_SENS_FRAMEQUAT: case MJB_SENS_FRAMEXAXIS: case MJB_SENS_FRAMEYAXIS:
		case MJB_SENS_FRAMEZAXIS: {
			const double *pos, *mat, *rpos = NULL, *rmat = NULL;
			double q[4], rq[4];
			frame_of(m, d, ot, id, &pos, &mat, q);
			if (rid >= 0) frame_of(m, d, rt, rid, &rpos, &rmat, rq);
			if (type == MJB_SENS_FRAMEPOS) {
				if (rid < 0) v3_copy(out, pos);
				else {
					double dif[3];
					v3_sub(dif, pos, rpos);
					m3_mulvecT(out, rmat, dif);
				}
			} else if (type == MJB_SENS_FRAMEQUAT) {
				if (rid < 0) memcpy(out, q, sizeof q);
				else {
					double neg[4] = { rq[0], -rq[1], -rq[2], -rq[3] };
					q_mul(out, neg, q);
				}
				is_real = 0;
			} else {
				int c = type - MJB_SENS_FRAMEXAXIS;
				double ax[3] = { mat[c], mat[3 + c], mat[6 + c] };
				if (rid < 0) v3_copy(out, ax);
				else m3_mulvecT(out, rmat, ax);
				is_real = 0;
			}
			break;
		}
		case MJB_SENS_SUBTREECOM: v3_copy(out, d->subtree_com + 3 * id); break;
		case MJB_SENS_CLOCK: out[0] = d->time[0]; break;
		case MJB_SENS_JOINTVEL: out[0] = d->qvel[m->jnt_dofadr[id]]; break;
		case MJB_SENS_ACTUATORVEL: out[0] = d->actuator_velocity[id]; break;
		case MJB_SENS_BALLANGVEL: v3_copy(out, d->qvel + m->jnt_dofadr[id]); break;
		case MJB_SENS_VELOCIMETER: case MJB_SENS_GYRO: {
			double xvel[6];
			object_velocity(m, d, MJB_OBJ_SITE, id, xvel, 1);
			v3_copy(out, type == MJB_SENS_GYRO ? xvel : xvel + 3);
			break;
		}
		case MJB_SENS_FRAMELINVEL: case MJB_SENS_FRAMEANGVEL: {
			double xvel[6];
			object_velocity(m, d, ot, id, xvel, 0);
			if (rid >= 0) {
				const double *pos, *mat, *rpos, *rmat;
				double q[4], rq[4], rvel[6];
				frame_of(m, d, ot, id, &pos, &mat, q);
				frame_of(m, d, rt, rid, &rpos, &rmat, rq);
				object_velocity(m, d, rt, rid, rvel, 0);
				for (int k = 0; k < 6; k++) xvel[k] -= rvel[k];
				if (type == MJB_SENS_FRAMELINVEL) {
					double rel[3], cr[3];
					v3_sub(rel, pos, rpos);
					v3_cross(cr, rel, rvel);
					v3_addto(xvel + 3, cr);
				}
				m3_mulvecT(out, rmat, type == MJB_SENS_FRAMELINVEL ? xvel + 3 : xvel);
			} else {
				v3_copy(out, type == MJB_SENS_FRAMELINVEL ? xvel + 3 : xvel);
			}
			break;
		}
		case MJB_SENS_ACTUATORFRC: out[0] = d->actuator_force[id]; break;
		case MJB_SENS_TENDONPOS: out[0] = d->ten_length[id]; break;
		case MJB_SENS_TENDONVEL: out[0] = d->ten_velocity[id]; break;
		case MJB_SENS_ACCELEROMETER: {
			double acc[6];
			object_acceleration(m, d, MJB_OBJ_SITE, id, acc, 1);
			v3_copy(out, acc + 3);
			break;
		}
		case MJB_SENS_FORCE: case MJB_SENS_TORQUE: {
			int body = m->site_bodyid[id];
			double tmp[6];
			transform_spatial_force(tmp, d->cfrc_int + 6 * body, d->site_xpos + 3 * id,
			                        d->subtree_com + 3 * m->body_rootid[body], d->site_xmat + 9 * id);
			v3_copy(out, type == MJB_SENS_FORCE ? tmp + 3 : tmp);
			break;
		}
		case MJB_SENS_TOUCH: {
			int body = m->site_bodyid[id];
			out[0] = 0;
			for (int c = 0; c < d->ncon[0]; c++) {
				int b1 = m->geom_bodyid[d->contact_geom[2 * c]], b2 = m->geom_bodyid[d->contact_geom[2 * c + 1]];
				if (!contact_active(m, d, c) || (body != b1 && body != b2)) continue;
				double lf[6], ray[3];
				contact_force(m, d, c, lf);
				if (lf[0] <= 0) continue;
				double sg = body == b2 ? -1.0 : 1.0;
				for (int k = 0; k < 3; k++) ray[k] = sg * d->contact_frame[9 * c + k];
				if (ray_hits_site(m, d, id, d->contact_pos + 3 * c, ray)) out[0] += lf[0];
			}
			break;
		}
		case MJB_SENS_FRAMELINACC: case MJB_SENS_FRAMEANGACC: {
			double acc[6];
			object_acceleration(m, d, ot, id, acc, 0);
			v3_copy(out, type == MJB_SENS_FRAMELINACC ? acc + 3 : acc);
			break;
		}
		/* mj_sensorPos / mj_sensorVel / mj_sensorAcc, limit sensors: the first limit row of the joint / tendon, zero without one
		 * (/root/reference mujoco_ros_sensors/src/mujoco_sensor_handler_plugin.cpp:331-343 serialises them as scalars) */
		case MJB_SENS_JOINTLIMITPOS: case MJB_SENS_JOINTLIMITVEL: case MJB_SENS_JOINTLIMITFRC:
		case MJB_SENS_TENDONLIMITPOS: case MJB_SENS_TENDONLIMITVEL: case MJB_SENS_TENDONLIMITFRC: {
			int want = type <= MJB_SENS_JOINTLIMITFRC ? MJB_CNSTR_LIMIT_JOINT : MJB_CNSTR_LIMIT_TENDON;
			int kind = (type - MJB_SENS_JOINTLIMITPOS) % 3; /* 0 pos, 1 vel, 2 frc */
			out[0] = 0;
			for (int r = 0; r < d->nefc[0]; r++)
				if (d->efc_type[r] == want && d->efc_id[r] == id) {
					out[0] = kind == 0 ? d->efc_pos[r] - d->efc_margin[r] : (kind == 1 ? d->efc_vel[r] : d->efc_force[r]);
					break;
				}
			break;
		}
		case MJB_SENS_JOINTACTFRC: out[0] = d->qfrc_actuator[m->jnt_dofadr[id]]; break;
		case MJB_SENS_MAGNETOMETER: m3_mulvecT(out, d->site_xmat + 9 * id, m->magnetic); break; /* the global flux in the site's frame */
		case MJB_SENS_RANGEFINDER: { /* along the site's z axis, the site's own body excluded; -1: nothing hit */
			const double *mt = d->site_xmat + 9 * id;
			double rv[3] = { mt[2], mt[5], mt[8] };
			out[0] = ray_all(m, d, d->site_xpos + 3 * id, rv, m->site_bodyid[id]);
			break;
		}
		case MJB_SENS_SUBTREELINVEL: case MJB_SENS_SUBTREEANGMOM: {
			double lin[3], ang[3];
			mjo_subtree_vel(m, d, id, lin, ang);
			v3_copy(out, type == MJB_SENS_SUBTREELINVEL ? lin : ang);
			break;
		}
		default: break;
		}
		double cutoff = m->sensor_cutoff[i];
		if (cutoff > 0 && is_real)
			for (int k = 0; k < m->sensor_dim[i]; k++) {
				if (type == MJB_SENS_TOUCH || type == MJB_SENS_RANGEFINDER) out[k] = out[k] > cutoff ? cutoff : out[k]; /* mjDATATYPE_POSITIVE */
				else out[k] = out[k] < -cutoff ? -cutoff : (out[k] > cutoff ? cutoff : out[k]);
			}
	}
}

/* ------------------------------------------------------------------ A16: mj_Euler */
static void integrate_pos(const mjb_model_desc *m, double *qpos, const double *qvel, double dt)
{
	for (int j = 0; j < m->njnt; j++) {
		int pa = m->jnt_qposadr[j], va = m->jnt_dofadr[j];
		switch (m->jnt_type[j]) {
		case MJB_JNT_FREE:
			for (int k = 0; k < 3; k++) qpos[pa + k] += dt * qvel[va + k];
			pa += 3;
			va += 3;
			/* fall through */
		case MJB_JNT_BALL: q_integrate(qpos + pa, qvel + va, dt); break;
		default: qpos[pa] += dt * qvel[va];
		}
	}
}

/* mj_advance, activations: act += h act_dot, clamped to actrange when the actuator is actlimited */
static void advance_act(const mjb_model_desc *m, double *act, const double *act_dot, double dt)
{
	if (m->na <= 0) return;
	for (int i = 0; i < m->nu; i++) {
		const int j = m->actuator_actadr[i];
		if (j < 0) continue;
		act[j] += dt * act_dot[j];
		if (m->actuator_actlimited[i]) {
			const double lo = m->actuator_actrange[2 * i], hi = m->actuator_actrange[2 * i + 1];
			act[j] = act[j] < lo ? lo : (act[j] > hi ? hi : act[j]);
		}
	}
}

void mjo_euler(const mjb_model_desc *m, mjo_data *d)
{
	int nv = m->nv;
	double dt = m->timestep[0];
	double *qacc = d->scratch_nv;
	int damping = 0;
	double *dd = d->scratch_nv2;  /* -diag(D): what the integrator treats implicitly */
	if (m->integrator == MJB_INT_IMPLICITFAST) {
		/* mj_implicit, mjINT_IMPLICITFAST: D = mjd_passive_vel + mjd_actuator_vel (no Coriolis terms), symmetrised; with joint dampers and
		 * joint transmissions it is diagonal: -damping_i + gear^2 (biasprm[2] + gainprm[2] * input).  mjDSBL_EULERDAMP is not consulted. */
		damping = 1;
		for (int i = 0; i < nv; i++) dd[i] = (m->disableflags & MJB_DSBL_PASSIVE) ? 0.0 : m->dof_damping[i];
		for (int i = 0; i < m->nu && !(m->disableflags & MJB_DSBL_ACTUATION); i++) {
			double bv = m->actuator_biastype[i] == MJB_BIAS_AFFINE ? m->actuator_biasprm[3 * i + 2] : 0.0;
			if (m->actuator_gaintype[i] == MJB_GAIN_AFFINE) {
				const int ja = m->na > 0 ? m->actuator_actadr[i] : -1;
				bv += m->actuator_gainprm[3 * i + 2] * (ja >= 0 ? d->act[ja] : d->ctrl[i]);
			}
			const double g = m->actuator_gear[6 * i];
			if (m->actuator_trntype[i] == MJB_TRN_TENDON) continue;  /* (moment' bv moment is not diagonal: the loader refuses bv != 0 there) */
			dd[m->jnt_dofadr[m->actuator_trnid[2 * i]]] -= g * g * bv;
		}
	} else {
		if (!(m->disableflags & MJB_DSBL_EULERDAMP))
			for (int i = 0; i < nv; i++)
				if (m->dof_damping[i] > 0) { damping = 1; break; }
		for (int i = 0; i < nv; i++) dd[i] = m->dof_damping[i];
	}
	if (!damping) {
		memcpy(qacc, d->qacc, sizeof(double) * (size_t)nv);
	} else {
		double *MhB = d->scratch_MM, *qH = d->scratch_MM + m->nM, *qHDiagInv = d->scratch_MM + 2 * m->nM;
		memcpy(MhB, d->qM, sizeof(double) * (size_t)m->nM);
		for (int i = 0; i < nv; i++) MhB[m->dof_Madr[i]] += dt * dd[i];
		factor_i(m, MhB, qH, qHDiagInv);
		for (int i = 0; i < nv; i++) qacc[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i];
		solve_ld(m, qacc, qH, qHDiagInv);
	}
	/* mj_advance */
	advance_act(m, d->act, d->act_dot, dt);
	for (int i = 0; i < nv; i++) d->qvel[i] += dt * qacc[i];
	integrate_pos(m, d->qpos, d->qvel, dt);
	d->time[0] += dt;
}

/* ------------------------------------------------------------------ pipeline */
void mjo_fwd_position(const mjb_model_desc *m, mjo_data *d)
{
	mjo_kinematics(m, d);
	mjo_com_pos(m, d);
	mjo_tendon(m, d);
	mjo_crb(m, d);
	mjo_factor_m(m, d);
	mjo_collision(m, d);
	mjo_make_constraint(m, d);
	mjo_transmission(m, d);
	mjo_project_constraint(m, d);
}

void mjo_fwd_velocity(const mjb_model_desc *m, mjo_data *d)
{
	mjo_com_vel(m, d);
	mjo_tendon_vel(m, d);
	mjo_passive(m, d);
	mjo_reference_constraint(m, d);
	mjo_rne(m, d);
}

static int bad(const double *x, int n)
{
	for (int i = 0; i < n; i++)
		if (!(x[i] == x[i]) || x[i] > MJO_MAXVAL || x[i] < -MJO_MAXVAL) return 1;
	return 0;
}

unsigned long long mjo_warning(const mjo_data *d, int which)
{
	return (which >= 0 && which < MJB_NWARNING) ? d->warning[which] : 0;
}

/* mj_energyPos + mj_energyVel: potential = -sum m g.xipos + spring energies, kinetic = 0.5 qvel' M qvel */
void mjo_energy(const mjb_model_desc *m, mjo_data *d)
{
	double pe = 0, ke = 0;
	if (!(m->disableflags & MJB_DSBL_GRAVITY))
		for (int b = 1; b < m->nbody; b++) pe -= m->body_mass[b] * v3_dot(m->gravity, d->xipos + 3 * b);
	if (!(m->disableflags & MJB_DSBL_PASSIVE)) {
		for (int j = 0; j < m->njnt; j++) {
			double k = m->jnt_stiffness[j];
			if (k == 0) continue;
			int pa = m->jnt_qposadr[j], jt = m->jnt_type[j];
			if (jt == MJB_JNT_FREE) {
				for (int c = 0; c < 3; c++) {
					double dq = d->qpos[pa + c] - m->qpos_spring[pa + c];
					pe += 0.5 * k * dq * dq;
				}
				pa += 3;
			}
			if (jt == MJB_JNT_FREE || jt == MJB_JNT_BALL) {
				double q[4], dif[3];
				memcpy(q, d->qpos + pa, sizeof q);
				q_normalize(q);
				q_sub(dif, q, m->qpos_spring + pa);
				pe += 0.5 * k * v3_dot(dif, dif);
			} else {
				double dq = d->qpos[pa] - m->qpos_spring[pa];
				pe += 0.5 * k * dq * dq;
			}
		}
		for (int t = 0; t < m->ntendon; t++) {
			double dl = d->ten_length[t] - m->tendon_lengthspring[t];
			pe += 0.5 * m->tendon_stiffness[t] * dl * dl;
		}
	}
	for (int i = 0; i < m->nv; i++) {
		int adr = m->dof_Madr[i];
		for (int j = i; j >= 0; j = m->dof_parentid[j], adr++)
			ke += (i == j ? 0.5 : 1.0) * d->qM[adr] * d->qvel[i] * d->qvel[j];
	}
	d->energy[0] = pe;
	d->energy[1] = ke;
}

void mjo_step1(const mjb_model_desc *m, mjo_data *d)
{
	/* mj_checkPos / mj_checkVel: warning + reset on NaN / huge (qpos first: its reset clears qvel) */
	if (bad(d->qpos, m->nq)) {
		d->warning[MJB_WARN_BADQPOS]++;
		mjo_reset_data(m, d);
	} else if (bad(d->qvel, m->nv)) {
		d->warning[MJB_WARN_BADQVEL]++;
		mjo_reset_data(m, d);
	}
	mjo_fwd_position(m, d);
	mjo_sensor(m, d, MJB_STAGE_POS);
	mjo_fwd_velocity(m, d);
	mjo_sensor(m, d, MJB_STAGE_VEL);
	if (m->enableflags & MJB_ENBL_ENERGY) mjo_energy(m, d);
}

static void forward_rest(const mjb_model_desc *m, mjo_data *d)
{
	mjo_fwd_actuation(m, d);
	mjo_fwd_acceleration(m, d);
	mjo_fwd_constraint(m, d);
	if (mjo_needs_rne_post(m)) mjo_rne_post_constraint(m, d);
	mjo_sensor(m, d, MJB_STAGE_ACC);
}

void mjo_forward(const mjb_model_desc *m, mjo_data *d)
{
	mjo_fwd_position(m, d);
	mjo_sensor(m, d, MJB_STAGE_POS);
	mjo_fwd_velocity(m, d);
	mjo_sensor(m, d, MJB_STAGE_VEL);
	if (m->enableflags & MJB_ENBL_ENERGY) mjo_energy(m, d);
	forward_rest(m, d);
}

/* mj_RungeKutta(m, d, 4) (MuJoCo 2.3.7 engine_forward.c; the classic tableau A = diag(1/2, 1/2, 1), B = (1/6, 1/3, 1/3, 1/6)):
 * X0 = (qpos, qvel), F0 = (qvel, qacc) of the step's own mj_forward; stage i = 1..3 starts from X0 advanced by h with the
 * derivative a_i F_{i-1} (positions through mj_integratePos, i.e. quaternions on the sphere), at time t0 + c_i h, evaluated by
 * mj_forwardSkip(.., mjSTAGE_NONE, skipsensor = 1); the step then advances X0 by h with sum_j B_j F_j (mj_advance with an explicit
 * velocity).  The constraint solver's warmstart is the one saved by the previous step's mj_advance for every evaluation, and the
 * last evaluation's qacc becomes the next one.  Activations are part of X (act) and F (act_dot); only the final advance clamps them. */
typedef struct {
	double *q0, *v0, *accv, *acca, *dxv, *sens, *t0, *a0, *acct;  /* a0: act at X0; acct: sum B act_dot */
} rk4_state;
static const double RK_A[3] = { 0.5, 0.5, 1.0 }, RK_B[4] = { 1.0 / 6.0, 1.0 / 3.0, 1.0 / 3.0, 1.0 / 6.0 }, RK_C[3] = { 0.5, 0.5, 1.0 };

static rk4_state rk4_view(const mjb_model_desc *m, double *buf)
{
	rk4_state S;
	S.q0 = buf;
	S.v0 = S.q0 + m->nq;
	S.accv = S.v0 + m->nv;
	S.acca = S.accv + m->nv;
	S.dxv = S.acca + 2 * m->nv;  /* (one nv block kept free where the kernel parks the warmstart) */
	S.sens = S.dxv + m->nv;
	S.t0 = S.sens + m->nsensordata;
	S.a0 = S.t0 + 1;
	S.acct = S.a0 + m->na;
	return S;
}

/* X0, the sensors and the time of the step, F0 = (qvel, qacc) of its own evaluation into the weighted sums */
static void rk4_begin(const mjb_model_desc *m, mjo_data *d, rk4_state S)
{
	const int nq = m->nq, nv = m->nv, ns = m->nsensordata;
	memcpy(S.q0, d->qpos, sizeof(double) * (size_t)nq);
	memcpy(S.v0, d->qvel, sizeof(double) * (size_t)nv);
	memcpy(S.sens, d->sensordata, sizeof(double) * (size_t)ns);
	S.t0[0] = d->time[0];
	for (int k = 0; k < nv; k++) {
		S.accv[k] = 0.0 + RK_B[0] * d->qvel[k];
		S.acca[k] = 0.0 + RK_B[0] * d->qacc[k];
	}
	for (int k = 0; k < m->na; k++) {
		S.a0[k] = d->act[k];
		S.acct[k] = 0.0 + RK_B[0] * d->act_dot[k];
	}
}

/* X_i = X0 (+) h a_i F_{i-1}; the warmstart every evaluation of this step starts from is the one the step came in with --
 * the solvers here save qacc as they finish, so it is put back */
static void rk4_set_stage(const mjb_model_desc *m, mjo_data *d, rk4_state S, int i)
{
	const int nq = m->nq, nv = m->nv;
	const double h = m->timestep[0];
	for (int k = 0; k < nv; k++) S.dxv[k] = 0.0 + RK_A[i - 1] * d->qvel[k];
	const double *qa = d->qacc;
	memcpy(d->qpos, S.q0, sizeof(double) * (size_t)nq);
	integrate_pos(m, d->qpos, S.dxv, h);
	for (int k = 0; k < nv; k++) d->qvel[k] = S.v0[k] + h * (0.0 + RK_A[i - 1] * qa[k]);
	for (int k = 0; k < m->na; k++) d->act[k] = S.a0[k] + h * (0.0 + RK_A[i - 1] * d->act_dot[k]);
	d->time[0] = S.t0[0] + RK_C[i - 1] * h;
	memcpy(d->qacc_warmstart, d->rk_warmstart, sizeof(double) * (size_t)nv);
}

static void rk4_accumulate(const mjb_model_desc *m, mjo_data *d, rk4_state S, int i)
{
	for (int k = 0; k < m->nv; k++) {
		S.accv[k] += RK_B[i] * d->qvel[k];
		S.acca[k] += RK_B[i] * d->qacc[k];
	}
	for (int k = 0; k < m->na; k++) S.acct[k] += RK_B[i] * d->act_dot[k];
}

static void rk4_finish(const mjb_model_desc *m, mjo_data *d, rk4_state S)
{
	const int nq = m->nq, nv = m->nv, ns = m->nsensordata;
	const double h = m->timestep[0];
	memcpy(d->qpos, S.q0, sizeof(double) * (size_t)nq);
	for (int k = 0; k < nv; k++) d->qvel[k] = S.v0[k] + h * S.acca[k];
	integrate_pos(m, d->qpos, S.accv, h);
	memcpy(d->act, S.a0, sizeof(double) * (size_t)m->na);
	advance_act(m, d->act, S.acct, h);
	d->time[0] = S.t0[0] + h;
	memcpy(d->sensordata, S.sens, sizeof(double) * (size_t)ns);  /* (sensors are skipped in the sub-stage evaluations) */
}

void mjo_rk4(const mjb_model_desc *m, mjo_data *d)
{
	const rk4_state S = rk4_view(m, d->rk_buf);
	rk4_begin(m, d, S);
	for (int i = 1; i < 4; i++) {
		rk4_set_stage(m, d, S, i);
		mjo_forward(m, d);
		rk4_accumulate(m, d, S, i);
	}
	rk4_finish(m, d, S);
}

void mjo_step2(const mjb_model_desc *m, mjo_data *d)
{
	if (m->integrator == MJB_INT_RK4) memcpy(d->rk_warmstart, d->qacc_warmstart, sizeof(double) * (size_t)m->nv);
	forward_rest(m, d);
	/* mj_checkAcc */
	if (bad(d->qacc, m->nv)) {
		d->warning[MJB_WARN_BADQACC]++;
		mjo_reset_data(m, d);
		if (m->integrator == MJB_INT_RK4) memcpy(d->rk_warmstart, d->qacc_warmstart, sizeof(double) * (size_t)m->nv);
		mjo_forward(m, d);
	}
	if (m->integrator == MJB_INT_RK4) mjo_rk4(m, d);
	else mjo_euler(m, d);
}

/* mjo_step2 of an RK4 step cut at the callback points of its four evaluations -- mj_forwardSkip invokes mjcb_passive / mjcb_control
 * in every one of them, which is why the reference has lastStageCallback at all (plugin_utils.h:119-125).  Stage rk = 0 .. 2 finishes
 * evaluation rk (actuation .. acceleration-stage sensors; rk == 0: the warmstart and mj_checkAcc of mjo_step2), folds it into the
 * weighted sums, sets X_{rk+1} and runs the position / velocity stages of evaluation rk + 1: the caller's callbacks then see that
 * evaluation's view (time = t0 + c h included).  Stage 3 finishes evaluation 3 and advances.  Stages 0, 1, 2, 3 in a row == mjo_step2. */
void mjo_step2_rk(const mjb_model_desc *m, mjo_data *d, int rk)
{
	const rk4_state S = rk4_view(m, d->rk_buf);
	if (rk == 0) {
		memcpy(d->rk_warmstart, d->qacc_warmstart, sizeof(double) * (size_t)m->nv);
		forward_rest(m, d);
		if (bad(d->qacc, m->nv)) {
			d->warning[MJB_WARN_BADQACC]++;
			mjo_reset_data(m, d);
			memcpy(d->rk_warmstart, d->qacc_warmstart, sizeof(double) * (size_t)m->nv);
			mjo_forward(m, d);
		}
		rk4_begin(m, d, S);
	} else {
		forward_rest(m, d);
		rk4_accumulate(m, d, S, rk);
	}
	if (rk == 3) {
		rk4_finish(m, d, S);
		return;
	}
	rk4_set_stage(m, d, S, rk + 1);
	mjo_fwd_position(m, d);
	mjo_sensor(m, d, MJB_STAGE_POS);
	mjo_fwd_velocity(m, d);
	mjo_sensor(m, d, MJB_STAGE_VEL);
	if (m->enableflags & MJB_ENBL_ENERGY) mjo_energy(m, d);
}

void mjo_step(const mjb_model_desc *m, mjo_data *d)
{
	mjo_step1(m, d);
	mjo_step2(m, d);
}

/* ------------------------------------------------------------------ ctrl noise (Philox + OU) */
static inline uint32_t mulhilo32(uint32_t a, uint32_t b, uint32_t *hi)
{
	uint64_t p = (uint64_t)a * (uint64_t)b;
	*hi = (uint32_t)(p >> 32);
	return (uint32_t)p;
}

/* Philox-4x32-10 (Salmon et al., SC'11), reference constants */
void mjo_philox4x32(uint32_t c[4], const uint32_t key[2])
{
	uint32_t k0 = key[0], k1 = key[1];
	for (int r = 0; r < 10; r++) {
		uint32_t hi0, hi1;
		uint32_t lo0 = mulhilo32(0xD2511F53u, c[0], &hi0);
		uint32_t lo1 = mulhilo32(0xCD9E8D57u, c[2], &hi1);
		uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
		c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
		k0 += 0x9E3779B9u;
		k1 += 0xBB67AE85u;
	}
}

double mjo_normal(uint64_t seed, uint64_t env, uint32_t step, uint32_t idx)
{
	uint32_t c[4] = { (uint32_t)env, (uint32_t)(env >> 32), step, idx };
	uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
	mjo_philox4x32(c, key);
	double u1 = ((double)c[0] + 0.5) * (1.0 / 4294967296.0);
	double u2 = ((double)c[1] + 0.5) * (1.0 / 4294967296.0);
	return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925 * u2);
}

/* /root/reference mujoco_ros/src/mujoco_env.cpp:469-481 */
void mjo_ctrl_noise(const mjb_model_desc *m, mjo_data *d, double noise_std, double noise_rate, uint64_t seed,
                    uint64_t env, uint32_t step)
{
	if (noise_std == 0) return;
	double rate = exp(-m->timestep[0] / fmax(noise_rate, MJO_MINVAL));
	double scale = noise_std * sqrt(1 - rate * rate);
	for (int i = 0; i < m->nu; i++) {
		d->ctrlnoise[i] = rate * d->ctrlnoise[i] + scale * mjo_normal(seed, env, step, (uint32_t)i);
		d->ctrl[i] = d->ctrlnoise[i];
	}
}

/* ------------------------------------------------------------------ threaded rollout */
typedef struct {
	const mjb_model_desc *m;
	int lo, hi, nsteps;
	double *qpos, *qvel, *sensordata;
	const double *ctrl;
	double std, rate;
	uint64_t seed;
	int64_t env_offset;
} rollout_arg;

static void *rollout_worker(void *p)
{
	rollout_arg *a = (rollout_arg *)p;
	const mjb_model_desc *m = a->m;
	mjo_data *d = mjo_make_data(m);
	for (int e = a->lo; e < a->hi; e++) {
		mjo_reset_data(m, d);
		memcpy(d->qpos, a->qpos + (size_t)e * m->nq, sizeof(double) * (size_t)m->nq);
		memcpy(d->qvel, a->qvel + (size_t)e * m->nv, sizeof(double) * (size_t)m->nv);
		if (a->ctrl) memcpy(d->ctrl, a->ctrl + (size_t)e * m->nu, sizeof(double) * (size_t)m->nu);
		for (int s = 0; s < a->nsteps; s++) {
			mjo_ctrl_noise(m, d, a->std, a->rate, a->seed, (uint64_t)(a->env_offset + e), (uint32_t)s);
			mjo_step(m, d);
		}
		memcpy(a->qpos + (size_t)e * m->nq, d->qpos, sizeof(double) * (size_t)m->nq);
		memcpy(a->qvel + (size_t)e * m->nv, d->qvel, sizeof(double) * (size_t)m->nv);
		if (a->sensordata)
			memcpy(a->sensordata + (size_t)e * m->nsensordata, d->sensordata, sizeof(double) * (size_t)m->nsensordata);
	}
	mjo_free_data(d);
	return NULL;
}

int mjo_rollout(const mjb_model_desc *m, int nenv, int nsteps, double *qpos, double *qvel, const double *ctrl,
                double *sensordata, double noise_std, double noise_rate, uint64_t seed, int64_t env_offset,
                int nthreads)
{
	if (nthreads < 1) nthreads = 1;
	if (nthreads > nenv) nthreads = nenv > 0 ? nenv : 1;
	pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
	rollout_arg *args = (rollout_arg *)calloc((size_t)nthreads, sizeof(rollout_arg));
	for (int t = 0; t < nthreads; t++) {
		args[t] = (rollout_arg){ m, (int)((int64_t)nenv * t / nthreads), (int)((int64_t)nenv * (t + 1) / nthreads),
			                     nsteps, qpos, qvel, sensordata, ctrl, noise_std, noise_rate, seed, env_offset };
		if (nthreads == 1) rollout_worker(&args[t]);
		else pthread_create(&th[t], NULL, rollout_worker, &args[t]);
	}
	if (nthreads > 1)
		for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
	free(th);
	free(args);
	return 0;
}
