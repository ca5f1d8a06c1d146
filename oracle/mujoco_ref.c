/* mujoco_ref.c -- thin C shim over the REAL MuJoCo C API.  TEST / BENCH INFRASTRUCTURE (see mjo.h).
 *
 * Built ONLY on a machine where $MUJOCO_DIR holds MuJoCo's own headers and shared library
 * ($MUJOCO_DIR/include/mujoco/mujoco.h, $MUJOCO_DIR/lib/libmujoco.so*): oracle/mujoco_ref.py compiles this file
 * against them into oracle/_ref/libmjref.so and then (a) checks the oracle restatement against real `mj_step` --
 * the call the reference makes at mujoco_ros/src/mujoco_env.cpp:498,552,593 -- and (b) times real `mj_step` as the
 * `cpu_baseline.mujoco` leg of bench.py.  Neither the build container nor the GPU box ships MuJoCo:
 * there this file is never compiled and every consumer reports "NOT MEASURED (library absent)".  No stand-in header
 * or library is ever written: without the real ones this shim is unbuildable by design. */
#include <mujoco/mujoco.h>
#include <string.h>

typedef struct mjref {
	mjModel *m;
	mjData *d;
} mjref;

const char *mjref_version(void) { return mj_versionString(); }

mjref *mjref_load(const char *xml_path, char *err, int errsz)
{
	mjref *out = NULL;
	mjModel *m = mj_loadXML(xml_path, NULL, err, errsz);
	if (!m) return NULL;
	mjData *d = mj_makeData(m);
	if (!d) {
		mj_deleteModel(m);
		return NULL;
	}
	out = (mjref *)mju_malloc(sizeof(mjref));
	out->m = m;
	out->d = d;
	return out;
}

void mjref_free(mjref *r)
{
	if (!r) return;
	mj_deleteData(r->d);
	mj_deleteModel(r->m);
	mju_free(r);
}

/* out[0..7] = nq nv nu nsensordata nbody ngeom ncon nefc */
void mjref_sizes(const mjref *r, int *out)
{
	out[0] = r->m->nq; out[1] = r->m->nv; out[2] = r->m->nu; out[3] = r->m->nsensordata;
	out[4] = r->m->nbody; out[5] = r->m->ngeom; out[6] = r->d->ncon; out[7] = r->d->nefc;
}

void mjref_reset(mjref *r) { mj_resetData(r->m, r->d); }

void mjref_set_state(mjref *r, const double *qpos, const double *qvel, const double *ctrl)
{
	if (qpos) memcpy(r->d->qpos, qpos, sizeof(mjtNum) * (size_t)r->m->nq);
	if (qvel) memcpy(r->d->qvel, qvel, sizeof(mjtNum) * (size_t)r->m->nv);
	if (ctrl) memcpy(r->d->ctrl, ctrl, sizeof(mjtNum) * (size_t)r->m->nu);
}

void mjref_get_state(const mjref *r, double *qpos, double *qvel, double *qacc, double *sensordata)
{
	if (qpos) memcpy(qpos, r->d->qpos, sizeof(mjtNum) * (size_t)r->m->nq);
	if (qvel) memcpy(qvel, r->d->qvel, sizeof(mjtNum) * (size_t)r->m->nv);
	if (qacc) memcpy(qacc, r->d->qacc, sizeof(mjtNum) * (size_t)r->m->nv);
	if (sensordata) memcpy(sensordata, r->d->sensordata, sizeof(mjtNum) * (size_t)r->m->nsensordata);
}

void mjref_forward(mjref *r) { mj_forward(r->m, r->d); }

/* n x mj_step; ctrl_seq == NULL keeps ctrl, else ctrl_seq[step][nu] is written before every step (the OU noise sequence
 * the engine generated, so that both sides see identical inputs) */
void mjref_step(mjref *r, int n, const double *ctrl_seq)
{
	for (int s = 0; s < n; s++) {
		if (ctrl_seq) memcpy(r->d->ctrl, ctrl_seq + (size_t)s * r->m->nu, sizeof(mjtNum) * (size_t)r->m->nu);
		mj_step(r->m, r->d);
	}
}

/* derived quantities after mj_forward / mj_step, by name; returns the element count (0: unknown field) */
int mjref_get(const mjref *r, const char *name, double *out, int cap)
{
	const mjModel *m = r->m;
	const mjData *d = r->d;
	const mjtNum *src = NULL;
	int n = 0;
#define F(nm, cnt) if (!strcmp(name, #nm)) { src = d->nm; n = (cnt); }
	F(xpos, 3 * m->nbody) F(xquat, 4 * m->nbody) F(xipos, 3 * m->nbody) F(subtree_com, 3 * m->nbody) F(cinert, 10 * m->nbody)
	F(cdof, 6 * m->nv) F(qM, m->nM) F(qLD, m->nM) F(qfrc_bias, m->nv) F(qfrc_passive, m->nv) F(qfrc_smooth, m->nv)
	F(qacc_smooth, m->nv) F(qfrc_constraint, m->nv) F(qacc, m->nv) F(cvel, 6 * m->nbody) F(geom_xpos, 3 * m->ngeom)
	F(efc_pos, d->nefc) F(efc_D, d->nefc) F(efc_R, d->nefc) F(efc_aref, d->nefc) F(efc_force, d->nefc) F(efc_vel, d->nefc)
	F(efc_margin, d->nefc) F(energy, 2)
#undef F
	if (!strcmp(name, "contact_dist")) {
		n = d->ncon;
		for (int i = 0; i < n && i < cap; i++) out[i] = d->contact[i].dist;
		return n;
	}
	if (!strcmp(name, "contact_pos")) {
		n = 3 * d->ncon;
		for (int i = 0; i < d->ncon && 3 * i + 2 < cap; i++) memcpy(out + 3 * i, d->contact[i].pos, 3 * sizeof(mjtNum));
		return n;
	}
	if (!strcmp(name, "contact_frame")) {
		n = 9 * d->ncon;
		for (int i = 0; i < d->ncon && 9 * i + 8 < cap; i++) memcpy(out + 9 * i, d->contact[i].frame, 9 * sizeof(mjtNum));
		return n;
	}
	if (!strcmp(name, "efc_J")) { /* dense rows, whatever MuJoCo's internal format */
		n = d->nefc * m->nv;
		if (n > cap) return n;
		if (mj_isSparse(m)) {
			memset(out, 0, sizeof(double) * (size_t)n);
			for (int i = 0; i < d->nefc; i++)
				for (int k = 0; k < d->efc_J_rownnz[i]; k++)
					out[(size_t)i * m->nv + d->efc_J_colind[d->efc_J_rowadr[i] + k]] = d->efc_J[d->efc_J_rowadr[i] + k];
		} else {
			memcpy(out, d->efc_J, sizeof(double) * (size_t)n);
		}
		return n;
	}
	if (!src) return 0;
	memcpy(out, src, sizeof(double) * (size_t)(n < cap ? n : cap));
	return n;
}

/* model constants by name (the same names as include/mjb_model_fields.def), for checking the MJCF-subset compiler */
int mjref_model(const mjref *r, const char *name, double *out, int cap)
{
	const mjModel *m = r->m;
	const mjtNum *src = NULL;
	int n = 0;
#define F(nm, cnt) if (!strcmp(name, #nm)) { src = m->nm; n = (cnt); }
	F(qpos0, m->nq) F(body_mass, m->nbody) F(body_inertia, 3 * m->nbody) F(body_ipos, 3 * m->nbody) F(body_iquat, 4 * m->nbody)
	F(body_subtreemass, m->nbody) F(body_invweight0, 2 * m->nbody) F(dof_invweight0, m->nv) F(dof_damping, m->nv)
	F(dof_armature, m->nv) F(geom_size, 3 * m->ngeom) F(geom_rbound, m->ngeom) F(geom_friction, 3 * m->ngeom)
	F(jnt_range, 2 * m->njnt) F(actuator_gainprm, mjNGAIN * m->nu) F(actuator_biasprm, mjNBIAS * m->nu)
#undef F
	if (!strcmp(name, "meaninertia")) {
		out[0] = m->stat.meaninertia;
		return 1;
	}
	if (!src) return 0;
	memcpy(out, src, sizeof(double) * (size_t)(n < cap ? n : cap));
	return n;
}
