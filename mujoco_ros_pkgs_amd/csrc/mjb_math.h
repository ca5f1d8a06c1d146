// mjb_math.h — fp64 device helpers for the gfx950 step kernels (registers in, registers out).
// Conventions are MuJoCo's: quaternions (w,x,y,z), row-major 3x3, spatial vectors rotation-first,
// 10-vector inertias (Ixx Iyy Izz Ixy Ixz Iyz mx my mz m).
#pragma once

#ifndef __HIPCC_RTC__  // (hiprtc brings its own runtime header)
#include <hip/hip_runtime.h>
#endif

#define DEVI static __device__ __forceinline__
#define MJB_MINVAL 1e-15
#define MJB_MAXVAL 1e10
#define MJB_MINIMP 0.0001  // mjMINIMP / mjMAXIMP: legal range of the solimp impedances (getsolparam)
#define MJB_MAXIMP 0.9999

DEVI void ld3(double *r, const double *p) { r[0] = p[0]; r[1] = p[1]; r[2] = p[2]; }
DEVI void ld4(double *r, const double *p) { r[0] = p[0]; r[1] = p[1]; r[2] = p[2]; r[3] = p[3]; }
DEVI void ld6(double *r, const double *p) { for (int k = 0; k < 6; k++) r[k] = p[k]; }
DEVI void ld9(double *r, const double *p) { for (int k = 0; k < 9; k++) r[k] = p[k]; }
DEVI void ld10(double *r, const double *p) { for (int k = 0; k < 10; k++) r[k] = p[k]; }
DEVI void st3(double *p, const double *r) { p[0] = r[0]; p[1] = r[1]; p[2] = r[2]; }
DEVI void st4(double *p, const double *r) { p[0] = r[0]; p[1] = r[1]; p[2] = r[2]; p[3] = r[3]; }
DEVI void st6(double *p, const double *r) { for (int k = 0; k < 6; k++) p[k] = r[k]; }
DEVI void st9(double *p, const double *r) { for (int k = 0; k < 9; k++) p[k] = r[k]; }
// model constants (scalar or vector loads from the constant address space)
template <typename P> DEVI void ldc3(double *r, P p) { r[0] = p[0]; r[1] = p[1]; r[2] = p[2]; }
template <typename P> DEVI void ldc4(double *r, P p) { r[0] = p[0]; r[1] = p[1]; r[2] = p[2]; r[3] = p[3]; }

DEVI double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
DEVI void cross3(double *r, const double *a, const double *b)
{
	r[0] = a[1] * b[2] - a[2] * b[1];
	r[1] = a[2] * b[0] - a[0] * b[2];
	r[2] = a[0] * b[1] - a[1] * b[0];
}
DEVI double dot6r(const double *a, const double *b)
{
	return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}

DEVI double normalize3(double *v)
{
	double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
	if (n < MJB_MINVAL) {
		v[0] = 1; v[1] = 0; v[2] = 0;
	} else {
		double s = 1.0 / n;
		v[0] *= s; v[1] *= s; v[2] *= s;
	}
	return n;
}
DEVI void normalize4(double *q)
{
	double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
	if (n < MJB_MINVAL) {
		q[0] = 1; q[1] = 0; q[2] = 0; q[3] = 0;
	} else if (fabs(n - 1) > MJB_MINVAL) {
		double s = 1.0 / n;
		q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s;
	}
}
DEVI void qmul(double *r, const double *a, const double *b)
{
	double t0 = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
	double t1 = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
	double t2 = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
	double t3 = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
	r[0] = t0; r[1] = t1; r[2] = t2; r[3] = t3;
}
DEVI bool quat_is_identity(const double *q) { return q[0] == 1 && q[1] == 0 && q[2] == 0 && q[3] == 0; }
// (mju_quat2Mat has a shortcut for the exact identity quaternion; the formula below yields exactly I for (1, 0, 0, 0), so the
//  branch is dropped: with it the compiler merged the two results through a scratch slot -- a scratch round trip per body in
//  the kinematics chain and the only recurring scratch traffic of the kernels)
DEVI void quat2mat(double *r, const double *q)
{
	const double q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
	const double q11 = q[1] * q[1], q12 = q[1] * q[2], q13 = q[1] * q[3];
	const double q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
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
// same map without the exact-identity shortcut (value-identical: the formula yields exactly I for (1,0,0,0))
DEVI void quat2mat_nocheck(double *r, const double *q)
{
	const double q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
	const double q11 = q[1] * q[1], q12 = q[1] * q[2], q13 = q[1] * q[3];
	const double q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
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
DEVI void matvec3(double *r, const double *M, const double *v)
{
	double t0 = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
	double t1 = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
	double t2 = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
	r[0] = t0; r[1] = t1; r[2] = t2;
}
DEVI void matTvec3(double *r, const double *M, const double *v)
{
	double t0 = M[0] * v[0] + M[3] * v[1] + M[6] * v[2];
	double t1 = M[1] * v[0] + M[4] * v[1] + M[7] * v[2];
	double t2 = M[2] * v[0] + M[5] * v[1] + M[8] * v[2];
	r[0] = t0; r[1] = t1; r[2] = t2;
}
// rotate a vector by a quaternion (through the rotation matrix, as MuJoCo 2.3.7's mju_rotVecQuat)
DEVI void rotvec_quat(double *r, const double *v, const double *q)
{
	if (v[0] == 0 && v[1] == 0 && v[2] == 0) {
		r[0] = r[1] = r[2] = 0;
	} else if (quat_is_identity(q)) {
		r[0] = v[0]; r[1] = v[1]; r[2] = v[2];
	} else {
		double M[9];
		quat2mat(M, q);
		matvec3(r, M, v);
	}
}
DEVI void axis_angle_quat(double *r, const double *axis, double angle)
{
	if (angle == 0) {
		r[0] = 1; r[1] = 0; r[2] = 0; r[3] = 0;
	} else {
		double s, c;
		sincos(angle * 0.5, &s, &c);
		r[0] = c; r[1] = axis[0] * s; r[2] = axis[1] * s; r[3] = axis[2] * s;
	}
}
DEVI void quat_integrate(double *q, const double *vel, double scale)
{
	double ax[3] = { vel[0], vel[1], vel[2] }, qr[4];
	double angle = scale * normalize3(ax);
	axis_angle_quat(qr, ax, angle);
	normalize4(q);
	qmul(q, q, qr);
}
// 3-vector that rotates qb onto qa, expressed in qb's frame (mju_subQuat)
DEVI void quat_sub(double *res, const double *qa, const double *qb)
{
	double qn[4] = { qb[0], -qb[1], -qb[2], -qb[3] }, qd[4];
	qmul(qd, qn, qa);
	double ax[3] = { qd[1], qd[2], qd[3] };
	double s = normalize3(ax);
	double speed = 2 * atan2(s, qd[0]);
	if (speed > 3.14159265358979323846) speed -= 2 * 3.14159265358979323846;
	res[0] = ax[0] * speed; res[1] = ax[1] * speed; res[2] = ax[2] * speed;
}

// ---- spatial algebra
DEVI void cross_motion(double *res, const double *vel, const double *v)
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
DEVI void cross_force(double *res, const double *vel, const double *f)
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
// inertia of a body (principal moments `inert`, frame `mat`, mass) about a point offset by `dif`
DEVI void inert_com(double *res, const double *inert, const double *mat, const double *dif, double mass)
{
	double t0 = mat[0] * inert[0], t1 = mat[3] * inert[0], t2 = mat[6] * inert[0];
	double t3 = mat[1] * inert[1], t4 = mat[4] * inert[1], t5 = mat[7] * inert[1];
	double t6 = mat[2] * inert[2], t7 = mat[5] * inert[2], t8 = mat[8] * inert[2];
	res[0] = mat[0] * t0 + mat[1] * t3 + mat[2] * t6;
	res[1] = mat[3] * t1 + mat[4] * t4 + mat[5] * t7;
	res[2] = mat[6] * t2 + mat[7] * t5 + mat[8] * t8;
	res[3] = mat[0] * t1 + mat[1] * t4 + mat[2] * t7;
	res[4] = mat[0] * t2 + mat[1] * t5 + mat[2] * t8;
	res[5] = mat[3] * t2 + mat[4] * t5 + mat[5] * t8;
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
DEVI void mul_inert_vec(double *res, const double *i, const double *v)
{
	res[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
	res[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
	res[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
	res[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
	res[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
	res[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}

// ---- Philox-4x32-10 counter-based generator + Box-Muller (identical stream to oracle/mjo_smooth.c)
DEVI double philox_normal(unsigned long long seed, unsigned long long env, unsigned int step, unsigned int idx)
{
	unsigned int c0 = (unsigned int)env, c1 = (unsigned int)(env >> 32), c2 = step, c3 = idx;
	unsigned int k0 = (unsigned int)seed, k1 = (unsigned int)(seed >> 32);
#pragma unroll
	for (int r = 0; r < 10; r++) {
		unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
		unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
		unsigned int n0 = (unsigned int)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned int)p1;
		unsigned int n2 = (unsigned int)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned int)p0;
		c0 = n0; c1 = n1; c2 = n2; c3 = n3;
		k0 += 0x9E3779B9u;
		k1 += 0xBB67AE85u;
	}
	double u1 = ((double)c0 + 0.5) * (1.0 / 4294967296.0);
	double u2 = ((double)c1 + 0.5) * (1.0 / 4294967296.0);
	return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925 * u2);
}
