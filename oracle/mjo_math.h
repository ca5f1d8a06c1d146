/* mjo_math.h — small fp64 vector/quaternion/spatial helpers of the CPU oracle (test infrastructure).
 * Each restates the same-named MuJoCo 2.3.7 utility ([UPSTREAM] engine_util_blas.c /
 * engine_util_spatial.c / engine_util_misc.c); operation order is kept as published so results are
 * reproducible to the last bit with -ffp-contract=off. */
#ifndef MJO_MATH_H_
#define MJO_MATH_H_

#include <math.h>
#include <string.h>

#include "mjo.h"

static inline void v3_zero(double *r) { r[0] = r[1] = r[2] = 0; }
static inline void v3_copy(double *r, const double *a) { r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; }
static inline void v3_add(double *r, const double *a, const double *b) { r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2]; }
static inline void v3_sub(double *r, const double *a, const double *b) { r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2]; }
static inline void v3_addto(double *r, const double *a) { r[0] += a[0]; r[1] += a[1]; r[2] += a[2]; }
static inline void v3_addtoscl(double *r, const double *a, double s) { r[0] += a[0] * s; r[1] += a[1] * s; r[2] += a[2] * s; }
static inline void v3_scl(double *r, const double *a, double s) { r[0] = a[0] * s; r[1] = a[1] * s; r[2] = a[2] * s; }
static inline double v3_dot(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void v3_cross(double *r, const double *a, const double *b)
{
	r[0] = a[1] * b[2] - a[2] * b[1];
	r[1] = a[2] * b[0] - a[0] * b[2];
	r[2] = a[0] * b[1] - a[1] * b[0];
}
/* mju_normalize3: returns the norm; degenerate -> (1,0,0) */
static inline double v3_normalize(double *v)
{
	double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
	if (n < MJO_MINVAL) {
		v[0] = 1; v[1] = 0; v[2] = 0;
	} else {
		double s = 1 / n;
		v[0] *= s; v[1] *= s; v[2] *= s;
	}
	return n;
}
/* mju_normalize4 */
static inline double q_normalize(double *q)
{
	double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
	if (n < MJO_MINVAL) {
		q[0] = 1; q[1] = 0; q[2] = 0; q[3] = 0;
	} else if (fabs(n - 1) > MJO_MINVAL) {
		double s = 1 / n;
		q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s;
	}
	return n;
}
/* mju_mulQuat */
static inline void q_mul(double *r, const double *a, const double *b)
{
	double t0 = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
	double t1 = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
	double t2 = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
	double t3 = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
	r[0] = t0; r[1] = t1; r[2] = t2; r[3] = t3;
}
/* mju_quat2Mat (row-major 3x3) */
static inline void q_to_mat(double *r, const double *q)
{
	if (q[0] == 1 && q[1] == 0 && q[2] == 0 && q[3] == 0) {
		r[0] = 1; r[1] = 0; r[2] = 0; r[3] = 0; r[4] = 1; r[5] = 0; r[6] = 0; r[7] = 0; r[8] = 1;
		return;
	}
	double q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
	double q11 = q[1] * q[1], q12 = q[1] * q[2], q13 = q[1] * q[3];
	double q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
	r[0] = q00 + q11 - q22 - q33;
	r[4] = q00 - q11 + q22 - q33;
	r[8] = q00 - q11 - q22 + q33;
	r[1] = 2 * (q12 - q03);
	r[2] = 2 * (q13 + q02);
	r[3] = 2 * (q12 + q03);
	r[5] = 2 * (q23 - q01);
	r[6] = 2 * (q13 - q02);
	r[7] = 2 * (q23 + q01);
}
/* mju_rotVecMat: r = M v */
static inline void m3_mulvec(double *r, const double *M, const double *v)
{
	double t0 = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
	double t1 = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
	double t2 = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
	r[0] = t0; r[1] = t1; r[2] = t2;
}
/* mju_rotVecMatT: r = M' v */
static inline void m3_mulvecT(double *r, const double *M, const double *v)
{
	double t0 = M[0] * v[0] + M[3] * v[1] + M[6] * v[2];
	double t1 = M[1] * v[0] + M[4] * v[1] + M[7] * v[2];
	double t2 = M[2] * v[0] + M[5] * v[1] + M[8] * v[2];
	r[0] = t0; r[1] = t1; r[2] = t2;
}
/* mju_rotVecQuat (2.3.7 form: through the rotation matrix) */
static inline void q_rotvec(double *r, const double *v, const double *q)
{
	if (v[0] == 0 && v[1] == 0 && v[2] == 0) {
		v3_zero(r);
	} else if (q[0] == 1 && q[1] == 0 && q[2] == 0 && q[3] == 0) {
		v3_copy(r, v);
	} else {
		double M[9];
		q_to_mat(M, q);
		m3_mulvec(r, M, v);
	}
}
/* mju_axisAngle2Quat */
static inline void q_axis_angle(double *r, const double *axis, double angle)
{
	if (angle == 0) {
		r[0] = 1; r[1] = 0; r[2] = 0; r[3] = 0;
	} else {
		double s = sin(angle * 0.5);
		r[0] = cos(angle * 0.5);
		r[1] = axis[0] * s; r[2] = axis[1] * s; r[3] = axis[2] * s;
	}
}
/* mju_quatIntegrate */
static inline void q_integrate(double *q, const double *vel, double scale)
{
	double tmp[3], qrot[4];
	v3_copy(tmp, vel);
	double angle = scale * v3_normalize(tmp);
	q_axis_angle(qrot, tmp, angle);
	q_normalize(q);
	q_mul(q, q, qrot);
}
/* mju_negQuat / mju_subQuat (velocity taking qb to qa, in qb's local frame) / mju_quat2Vel */
static inline void q_sub(double *res, const double *qa, const double *qb)
{
	double qneg[4] = { qb[0], -qb[1], -qb[2], -qb[3] }, qdif[4];
	q_mul(qdif, qneg, qa);
	/* quat2Vel with dt = 1 */
	double axis[3] = { qdif[1], qdif[2], qdif[3] };
	double sin_a_2 = v3_normalize(axis);
	double speed = 2 * atan2(sin_a_2, qdif[0]);
	if (speed > 3.14159265358979323846) speed -= 2 * 3.14159265358979323846;
	v3_scl(res, axis, speed);
}
/* the rotation angle of a ball joint and its unit axis, as mj_instantiateLimit takes them: mju_quat2Vel(aa, quat, 1) -- the angle in (-pi, pi]
 * times the axis -- then mju_normalize3: the angle's magnitude is returned, aa becomes the (sign-adjusted) unit axis, (1, 0, 0) at zero rotation */
static inline double mjo_ball_angle(const double *quat, double *aa)
{
	double axis[3] = { quat[1], quat[2], quat[3] };
	double sin_a_2 = v3_normalize(axis);
	double speed = 2 * atan2(sin_a_2, quat[0]);
	if (speed > 3.14159265358979323846) speed -= 2 * 3.14159265358979323846;
	v3_scl(aa, axis, speed);
	return v3_normalize(aa);
}
/* 3x3 products: r = A B, r = A' B, r = A B' */
static inline void m3_mul(double *r, const double *A, const double *B)
{
	double t[9];
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
	memcpy(r, t, sizeof t);
}

/* ---- spatial (6D, rotation first) ---- */
static inline double dot6(const double *a, const double *b)
{
	return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}
/* mju_crossMotion */
static inline void cross_motion(double *res, const double *vel, const double *v)
{
	res[0] = -vel[2] * v[1] + vel[1] * v[2];
	res[1] = vel[2] * v[0] - vel[0] * v[2];
	res[2] = -vel[1] * v[0] + vel[0] * v[1];
	res[3] = -vel[2] * v[4] + vel[1] * v[5];
	res[4] = vel[2] * v[3] - vel[0] * v[5];
	res[5] = -vel[1] * v[3] + vel[0] * v[4];
	res[3] += -vel[5] * v[1] + vel[4] * v[2];
	res[4] += vel[5] * v[0] - vel[3] * v[2];
	res[5] += -vel[4] * v[0] + vel[3] * v[1];
}
/* mju_crossForce */
static inline void cross_force(double *res, const double *vel, const double *f)
{
	res[0] = -vel[2] * f[1] + vel[1] * f[2];
	res[1] = vel[2] * f[0] - vel[0] * f[2];
	res[2] = -vel[1] * f[0] + vel[0] * f[1];
	res[3] = -vel[2] * f[4] + vel[1] * f[5];
	res[4] = vel[2] * f[3] - vel[0] * f[5];
	res[5] = -vel[1] * f[3] + vel[0] * f[4];
	res[0] += -vel[5] * f[4] + vel[4] * f[5];
	res[1] += vel[5] * f[3] - vel[3] * f[5];
	res[2] += -vel[4] * f[3] + vel[3] * f[4];
}
/* mju_inertCom: 10-vector inertia about a point offset by `dif` from the body com */
static inline void inert_com(double *res, const double *inert, const double *mat, const double *dif, double mass)
{
	double tmp[9];
	tmp[0] = mat[0] * inert[0]; tmp[1] = mat[3] * inert[0]; tmp[2] = mat[6] * inert[0];
	tmp[3] = mat[1] * inert[1]; tmp[4] = mat[4] * inert[1]; tmp[5] = mat[7] * inert[1];
	tmp[6] = mat[2] * inert[2]; tmp[7] = mat[5] * inert[2]; tmp[8] = mat[8] * inert[2];
	res[0] = mat[0] * tmp[0] + mat[1] * tmp[3] + mat[2] * tmp[6];
	res[1] = mat[3] * tmp[1] + mat[4] * tmp[4] + mat[5] * tmp[7];
	res[2] = mat[6] * tmp[2] + mat[7] * tmp[5] + mat[8] * tmp[8];
	res[3] = mat[0] * tmp[1] + mat[1] * tmp[4] + mat[2] * tmp[7];
	res[4] = mat[0] * tmp[2] + mat[1] * tmp[5] + mat[2] * tmp[8];
	res[5] = mat[3] * tmp[2] + mat[4] * tmp[5] + mat[5] * tmp[8];
	res[0] += mass * (dif[1] * dif[1] + dif[2] * dif[2]);
	res[1] += mass * (dif[0] * dif[0] + dif[2] * dif[2]);
	res[2] += mass * (dif[0] * dif[0] + dif[1] * dif[1]);
	res[3] -= mass * dif[0] * dif[1];
	res[4] -= mass * dif[0] * dif[2];
	res[5] -= mass * dif[1] * dif[2];
	res[6] = mass * dif[0];
	res[7] = mass * dif[1];
	res[8] = mass * dif[2];
	res[9] = mass;
}
/* mju_mulInertVec */
static inline void mul_inert_vec(double *res, const double *i, const double *v)
{
	res[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
	res[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
	res[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
	res[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
	res[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
	res[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
/* mju_dofCom */
static inline void dof_com(double *res, const double *axis, const double *offset)
{
	if (offset) {
		v3_copy(res, axis);
		v3_cross(res + 3, axis, offset);
	} else {
		v3_zero(res);
		v3_copy(res + 3, axis);
	}
}
/* mju_mulDofVec: res = sum_k mat[k] * vec[k] (6-vectors) */
static inline void mul_dof_vec(double *res, const double *mat, const double *vec, int n)
{
	for (int i = 0; i < 6; i++) res[i] = 0;
	for (int k = 0; k < n; k++)
		for (int i = 0; i < 6; i++) res[i] += mat[6 * k + i] * vec[k];
}
/* mju_transformSpatial for the motion case used by mj_objectVelocity:
 * move a com-based spatial velocity `vec` to point `newpos` (old reference `oldpos`) and rotate
 * into frame `rotnew2old` (row-major; may be NULL). flg_force = 0. */
static inline void transform_spatial_motion(double *res, const double *vec, const double *newpos,
                                            const double *oldpos, const double *rotnew2old)
{
	double cros[3], dif[3], tran[6];
	memcpy(tran, vec, 6 * sizeof(double));
	v3_sub(dif, newpos, oldpos);
	v3_cross(cros, dif, vec);
	v3_sub(tran + 3, vec + 3, cros);
	if (rotnew2old) {
		m3_mulvecT(res, rotnew2old, tran);
		m3_mulvecT(res + 3, rotnew2old, tran + 3);
	} else {
		memcpy(res, tran, 6 * sizeof(double));
	}
}

/* mju_transformSpatial with flg_force = 1: (torque, force) moved from `oldpos` to `newpos`, optionally rotated */
static inline void transform_spatial_force(double *res, const double *vec, const double *newpos, const double *oldpos,
                                           const double *rotnew2old)
{
	double cros[3], dif[3], tran[6];
	memcpy(tran, vec, 6 * sizeof(double));
	v3_sub(dif, newpos, oldpos);
	v3_cross(cros, dif, vec + 3);
	v3_sub(tran, vec, cros);
	if (rotnew2old) {
		m3_mulvecT(res, rotnew2old, tran);
		m3_mulvecT(res + 3, rotnew2old, tran + 3);
	} else {
		memcpy(res, tran, 6 * sizeof(double));
	}
}

#endif
