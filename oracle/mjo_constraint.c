/* mjo_constraint.c — CPU oracle (TEST INFRASTRUCTURE): collision, constraint rows, PGS solve.
 * See mjo.h for provenance ("parity unpinned") and usage restrictions.
 *
 * Restates MuJoCo 2.3.7 [UPSTREAM] engine_collision_driver.c / engine_collision_primitive.c /
 * engine_core_constraint.c / engine_solver.c / engine_forward.c (warmstart) for the primitive subset
 * of include/mjb.h: geoms plane / sphere / capsule / box (pairs listed in collpair_geom), joint
 * limits on hinge / slide joints, frictionless and pyramidal contacts, PGS.  Reached in the reference
 * only through mj_step / mj_forward (/root/reference mujoco_ros/src/mujoco_env.cpp:498,552,593,329,621).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "mjo.h"
#include "mjo_math.h"

/* ------------------------------------------------------------------ contact frame (mju_makeFrame) */
static void make_frame(double *frame)
{
	v3_normalize(frame);
	if (sqrt(v3_dot(frame + 3, frame + 3)) < 0.5) {
		v3_zero(frame + 3);
		if (frame[1] < 0.5 && frame[1] > -0.5) frame[4] = 1;
		else frame[5] = 1;
	}
	double t = v3_dot(frame, frame + 3);
	frame[3] -= t * frame[0];
	frame[4] -= t * frame[1];
	frame[5] -= t * frame[2];
	v3_normalize(frame + 3);
	v3_cross(frame + 6, frame, frame + 3);
}

typedef struct {
	double dist, pos[3], frame[9];
} rawcon;

/* ------------------------------------------------------------------ A5: primitive narrow phase */
/* sphere-sphere core (mjraw_SphereSphere): centres p1/p2, radii r1/r2 */
/* Ties.  A scene built on a grid -- boxes stacked with identity orientations, a capsule lying along a box edge, geoms exactly touching -- puts the
 * narrow phase's comparisons on their knife edge: which face axis is the least penetrated, whether a vertex lies inside a side plane, whether a surface
 * at distance == margin is a contact.  Decided by the last bit they come out differently in two implementations of the same steps (fma contraction is
 * enough).  Every such comparison carries this offset: equal candidates keep the first in order, a point on a plane is on it, a distance equal to the
 * margin is inside it; the knife edges move to where no grid puts a scene.  (csrc/mjb_constraint.h: MJB_TIE, the same places.) */
#define MJO_TIE 1e-12
static int raw_sphere_sphere(rawcon *c, const double *p1, double r1, const double *p2, double r2, double margin)
{
	double dif[3];
	v3_sub(dif, p2, p1);
	double cdist = sqrt(v3_dot(dif, dif));
	if (cdist > margin + r1 + r2) return 0;
	c->dist = cdist - r1 - r2;
	memset(c->frame, 0, sizeof c->frame);
	v3_copy(c->frame, dif);
	v3_normalize(c->frame); /* degenerate -> (1,0,0) */
	for (int k = 0; k < 3; k++) c->pos[k] = p1[k] + c->frame[k] * (r1 + 0.5 * c->dist);
	return 1;
}

/* plane (pos1, normal = mat1 z-axis) vs sphere centre p, radius r (mjc_PlaneSphere) */
static int raw_plane_sphere(rawcon *c, const double *pos1, const double *mat1, const double *p, double r, double margin)
{
	double n[3] = { mat1[2], mat1[5], mat1[8] }, tmp[3];
	v3_sub(tmp, p, pos1);
	double cdist = v3_dot(tmp, n);
	if (cdist > margin + r) return 0;
	c->dist = cdist - r;
	memset(c->frame, 0, sizeof c->frame);
	v3_copy(c->frame, n);
	for (int k = 0; k < 3; k++) c->pos[k] = p[k] - n[k] * (r + 0.5 * c->dist);
	return 1;
}

static int plane_capsule(rawcon *c, const double *pos1, const double *mat1, const double *pos2, const double *mat2,
                         const double *size2, double margin)
{
	double axis[3] = { mat2[2], mat2[5], mat2[8] }, seg[3], p[3];
	v3_scl(seg, axis, size2[1]);
	v3_add(p, pos2, seg);
	int n = raw_plane_sphere(c, pos1, mat1, p, size2[0], margin);
	v3_sub(p, pos2, seg);
	n += raw_plane_sphere(c + n, pos1, mat1, p, size2[0], margin);
	/* align the first tangent with the capsule axis (made orthogonal by make_frame) */
	for (int i = 0; i < n; i++) v3_copy(c[i].frame + 3, axis);
	return n;
}

static int plane_box(rawcon *c, const double *pos1, const double *mat1, const double *pos2, const double *mat2,
                     const double *size2, double margin)
{
	double n[3] = { mat1[2], mat1[5], mat1[8] }, dif[3];
	v3_sub(dif, pos2, pos1);
	double dist = v3_dot(dif, n);
	int cnt = 0;
	for (int i = 0; i < 8; i++) {
		double vec[3] = { (i & 1) ? size2[0] : -size2[0], (i & 2) ? size2[1] : -size2[1], (i & 4) ? size2[2] : -size2[2] };
		double corner[3];
		m3_mulvec(corner, mat2, vec);
		double ldist = v3_dot(n, corner);
		if (dist + ldist > margin || ldist > 0) continue;
		c[cnt].dist = dist + ldist;
		memset(c[cnt].frame, 0, sizeof c[cnt].frame);
		v3_copy(c[cnt].frame, n);
		v3_addto(corner, pos2);
		for (int k = 0; k < 3; k++) c[cnt].pos[k] = corner[k] - n[k] * c[cnt].dist * 0.5;
		if (++cnt >= 4) return 4;
	}
	return cnt;
}

static int sphere_capsule(rawcon *c, const double *pos1, double r1, const double *pos2, const double *mat2,
                          const double *size2, double margin)
{
	double axis[3] = { mat2[2], mat2[5], mat2[8] }, vec[3], p[3];
	v3_sub(vec, pos1, pos2);
	double x = v3_dot(axis, vec);
	x = x < -size2[1] ? -size2[1] : (x > size2[1] ? size2[1] : x);
	for (int k = 0; k < 3; k++) p[k] = pos2[k] + axis[k] * x;
	return raw_sphere_sphere(c, pos1, r1, p, size2[0], margin);
}

static double clipd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

static int capsule_capsule(rawcon *c, const double *pos1, const double *mat1, const double *size1, const double *pos2,
                           const double *mat2, const double *size2, double margin)
{
	double a1[3] = { mat1[2], mat1[5], mat1[8] }, a2[3] = { mat2[2], mat2[5], mat2[8] }, dif[3];
	v3_sub(dif, pos1, pos2);
	double ma = v3_dot(a1, a1), mb = -v3_dot(a1, a2), mc = v3_dot(a2, a2);
	double u = -v3_dot(a1, dif), v = v3_dot(a2, dif);
	double det = ma * mc - mb * mb;
	double p1[3], p2[3];
	if (fabs(det) >= MJO_MINVAL) {
		double x1 = (mc * u - mb * v) / det, x2 = (ma * v - mb * u) / det;
		if (x1 > size1[1]) {
			x1 = size1[1];
			x2 = (v - mb * size1[1]) / mc;
		} else if (x1 < -size1[1]) {
			x1 = -size1[1];
			x2 = (v + mb * size1[1]) / mc;
		}
		if (x2 > size2[1]) {
			x2 = size2[1];
			x1 = clipd((u - mb * size2[1]) / ma, -size1[1], size1[1]);
		} else if (x2 < -size2[1]) {
			x2 = -size2[1];
			x1 = clipd((u + mb * size2[1]) / ma, -size1[1], size1[1]);
		}
		for (int k = 0; k < 3; k++) {
			p1[k] = pos1[k] + a1[k] * x1;
			p2[k] = pos2[k] + a2[k] * x2;
		}
		return raw_sphere_sphere(c, p1, size1[0], p2, size2[0], margin);
	}
	/* parallel axes: the two ends of capsule 1 against segment 2, then (if fewer than 2) the ends of 2 against 1 */
	int n = 0;
	for (int s = 1; s >= -1 && n < 2; s -= 2) {
		for (int k = 0; k < 3; k++) p1[k] = pos1[k] + a1[k] * s * size1[1];
		double d2[3];
		v3_sub(d2, p1, pos2);
		double x2 = clipd(v3_dot(a2, d2) / mc, -size2[1], size2[1]);
		for (int k = 0; k < 3; k++) p2[k] = pos2[k] + a2[k] * x2;
		n += raw_sphere_sphere(c + n, p1, size1[0], p2, size2[0], margin);
	}
	for (int s = 1; s >= -1 && n < 2; s -= 2) {
		for (int k = 0; k < 3; k++) p2[k] = pos2[k] + a2[k] * s * size2[1];
		double d1[3];
		v3_sub(d1, p2, pos1);
		double x1 = clipd(v3_dot(a1, d1) / ma, -size1[1], size1[1]);
		if (fabs(fabs(x1) - size1[1]) < MJO_MINVAL) continue; /* already produced by the first loop */
		for (int k = 0; k < 3; k++) p1[k] = pos1[k] + a1[k] * x1;
		n += raw_sphere_sphere(c + n, p1, size1[0], p2, size2[0], margin);
	}
	return n;
}

static int sphere_box(rawcon *c, const double *pos1, double r1, const double *pos2, const double *mat2,
                      const double *size2, double margin)
{
	double tmp[3], center[3], clamped[3];
	v3_sub(tmp, pos1, pos2);
	m3_mulvecT(center, mat2, tmp);
	for (int i = 0; i < 3; i++) clamped[i] = clipd(center[i], -size2[i], size2[i]);
	double dv[3];
	v3_sub(dv, clamped, center);
	double dist = sqrt(v3_dot(dv, dv));
	if (dist - r1 > margin) return 0;
	double nloc[3] = { 0, 0, 0 }, ploc[3];
	memset(c->frame, 0, sizeof c->frame);
	if (dist <= MJO_MINVAL) {
		/* centre inside the box: push out through the nearest face */
		double closest = 2 * fmax(size2[0], fmax(size2[1], size2[2]));
		int k = 0;
		for (int i = 0; i < 6; i++) {
			double fd = fabs(((i % 2) ? 1 : -1) * size2[i / 2] - center[i / 2]);
			if (closest > fd + MJO_TIE) {
				closest = fd;
				k = i;
			}
		}
		nloc[k / 2] = (k % 2) ? -1 : 1;
		for (int i = 0; i < 3; i++) ploc[i] = center[i] + nloc[i] * (r1 - closest) / 2;
		c->dist = -closest - r1;
	} else {
		double deepest[3];
		for (int i = 0; i < 3; i++) {
			deepest[i] = center[i] + dv[i] * (r1 / dist);
			ploc[i] = 0.5 * (clamped[i] + deepest[i]);
			nloc[i] = dv[i] / dist;
		}
		c->dist = dist - r1;
	}
	m3_mulvec(c->frame, mat2, nloc);
	m3_mulvec(c->pos, mat2, ploc);
	v3_addto(c->pos, pos2);
	return 1;
}

/* capsule - box.  MuJoCo's own routine (engine_collision_box.c, mjc_CapsuleBox) is not available in this
 * environment; this is a geometric restatement of its contract (at most two contacts, each one the sphere-box
 * contact of a point of the capsule axis) built on the convex function g(t) = dist(axis point t, box)^2:
 *   1. [tlo, thi] = the minimiser set of g on [-h, h], found by bisection on the monotone slope g'(t);
 *   2. the closest feature at its midpoint decides the candidate axis points: a FACE -> the two ends of the axis
 *      stretch lying over that face (a capsule resting flat or tilted on a face gets both ends, each with its own
 *      depth); axis INSIDE the box -> the ends of the inside stretch; an EDGE / VERTEX -> the ends of the
 *      minimiser set (one point unless the axis is parallel to the edge);
 *   3. each candidate goes through sphere_box(); candidates closer than 1e-6 h collapse into one.
 * The HIP narrow phase (mjb_constraint.h, capsule_box) follows the same steps operation for operation. */
/* (a direction component below 1e-12 counts as parallel to that face pair: it contributes nothing to the slope.  Without the
 *  snap an axis that is parallel to a box edge up to rounding gives a slope of +-1e-18 whose SIGN decides between one contact and
 *  two -- found by tests/test_gpu_capsule_box.py, where the GPU's and this file's rotation matrices differ in the last bit) */
#define MJO_CAPBOX_PAR 1e-12
static double capbox_slope(const double *p0, const double *d, const double *s, double t)
{
	double g = 0;
	for (int i = 0; i < 3; i++) {
		double p = p0[i] + t * d[i];
		double r = p - clipd(p, -s[i], s[i]);
		if (fabs(r) <= 4e-15 * s[i]) r = 0; /* (a residual of a few ulps of the half size is the face plane itself: the HIP routine evaluates the slope AT the crossings) */
		g += r * (fabs(d[i]) <= MJO_CAPBOX_PAR ? 0.0 : d[i]);
	}
	return g;
}

static int capsule_box(rawcon *c, const double *pos1, const double *mat1, const double *size1, const double *pos2,
                       const double *mat2, const double *size2, double margin)
{
	const double r = size1[0], h = size1[1];
	double axis[3] = { mat1[2], mat1[5], mat1[8] }, tmp[3], p0[3], d[3];
	v3_sub(tmp, pos1, pos2);
	m3_mulvecT(p0, mat2, tmp);
	m3_mulvecT(d, mat2, axis);
	const double glo = capbox_slope(p0, d, size2, -h), ghi = capbox_slope(p0, d, size2, h);
	double tlo, thi;
	if (glo >= 0) tlo = -h;
	else if (ghi < 0) tlo = h;
	else {
		double lo = -h, hi = h;
		for (int it = 0; it < 60; it++) {
			double mid = 0.5 * (lo + hi);
			if (capbox_slope(p0, d, size2, mid) >= 0) hi = mid;
			else lo = mid;
		}
		tlo = hi;
	}
	if (ghi <= 0) thi = h;
	else if (glo > 0) thi = -h;
	else {
		double lo = -h, hi = h;
		for (int it = 0; it < 60; it++) {
			double mid = 0.5 * (lo + hi);
			if (capbox_slope(p0, d, size2, mid) > 0) hi = mid;
			else lo = mid;
		}
		thi = lo;
	}
	if (thi < tlo) thi = tlo;
	const double ts = 0.5 * (tlo + thi);
	int nout = 0, face = 0;
	for (int i = 0; i < 3; i++)
		if (fabs(p0[i] + ts * d[i]) > size2[i] + MJO_TIE) {
			nout++;
			face = i;
		}
	double ta = tlo, tb = thi;
	if (nout <= 1) {
		ta = -h;
		tb = h;
		for (int j = 0; j < 3; j++) {
			if (nout == 1 && j == face) continue;
			if (fabs(d[j]) <= MJO_MINVAL) continue;
			double t1 = (-size2[j] - p0[j]) / d[j], t2 = (size2[j] - p0[j]) / d[j];
			if (t1 > t2) {
				double sw = t1;
				t1 = t2;
				t2 = sw;
			}
			if (t1 > ta) ta = t1;
			if (t2 < tb) tb = t2;
		}
		if (ta > tb) ta = tb = ts;
	}
	int n = 0;
	double ctr[3];
	for (int k = 0; k < 3; k++) ctr[k] = pos1[k] + axis[k] * ta;
	n += sphere_box(c + n, ctr, r, pos2, mat2, size2, margin);
	if (tb - ta > 1e-6 * h) {
		for (int k = 0; k < 3; k++) ctr[k] = pos1[k] + axis[k] * tb;
		n += sphere_box(c + n, ctr, r, pos2, mat2, size2, margin);
	}
	return n;
}

/* box - box.  MuJoCo's own routine (engine_collision_box.c, mjc_BoxBox) is not available in this environment; this
 * is the classical separating-axis + face-clipping construction with the same contract (contacts on the feature pair
 * of least penetration, frame normal from geom 1 to geom 2, position midway between the surfaces), up to 8
 * contacts per pair (mjc_BoxBox's count):
 *   1. separation along the 15 candidate axes (3 + 3 face normals, 9 edge x edge); any separation > margin: no
 *      contact; the axis of least penetration wins, edge axes only when they beat the best face axis by 5 %;
 *   2. face axis: the most anti-parallel face of the other box (4 vertices) is clipped against the side planes of
 *      the reference face (Sutherland-Hodgman, <= 8 vertices); every clipped vertex closer than margin to the reference
 *      face becomes a contact (dist = signed height above the face);
 *   3. edge x edge axis: one contact at the closest points of the two supporting edges.
 * The HIP narrow phase (mjb_constraint.h, box_box) follows the same steps operation for operation. */
static int box_box(rawcon *c, const double *pos1, const double *mat1, const double *size1, const double *pos2,
                   const double *mat2, const double *size2, double margin)
{
	double A[3][3], B[3][3], C[3][3], Q[3][3], t[3], tA[3], tB[3];
	for (int i = 0; i < 3; i++)
		for (int k = 0; k < 3; k++) {
			A[i][k] = mat1[3 * k + i]; /* axis i of box 1 in the world = column i */
			B[i][k] = mat2[3 * k + i];
		}
	v3_sub(t, pos2, pos1);
	for (int i = 0; i < 3; i++) {
		tA[i] = v3_dot(t, A[i]);
		tB[i] = v3_dot(t, B[i]);
		for (int j = 0; j < 3; j++) {
			C[i][j] = v3_dot(A[i], B[j]);
			Q[i][j] = fabs(C[i][j]) + 1e-12;
		}
	}
	/* 1. separating axes */
	double best = -1e300;
	int code = -1; /* 0..2 face of box 1, 3..5 face of box 2, 6.. edge pair 6 + 3 i + j */
	for (int i = 0; i < 3; i++) {
		double s = fabs(tA[i]) - (size1[i] + size2[0] * Q[i][0] + size2[1] * Q[i][1] + size2[2] * Q[i][2]);
		if (s > margin) return 0;
		if (s > best + MJO_TIE) { best = s; code = i; }
	}
	for (int j = 0; j < 3; j++) {
		double s = fabs(tB[j]) - (size2[j] + size1[0] * Q[0][j] + size1[1] * Q[1][j] + size1[2] * Q[2][j]);
		if (s > margin) return 0;
		if (s > best + MJO_TIE) { best = s; code = 3 + j; }
	}
	double ebest = -1e300;
	int ecode = -1;
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++) {
			const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
			double l = sqrt(fmax(0.0, 1.0 - C[i][j] * C[i][j]));
			if (l < 1e-6) continue;
			double s = fabs(tA[i2] * C[i1][j] - tA[i1] * C[i2][j]) -
			           (size1[i1] * Q[i2][j] + size1[i2] * Q[i1][j] + size2[j1] * Q[i][j2] + size2[j2] * Q[i][j1]);
			s /= l;
			if (s > margin) return 0;
			if (s > ebest + MJO_TIE) { ebest = s; ecode = 6 + 3 * i + j; }
		}
	if (ecode >= 0 && ebest > best + 0.05 * fabs(best) + 1e-9) {
		/* 3. edge x edge */
		const int i = (ecode - 6) / 3, j = (ecode - 6) % 3;
		double n[3];
		v3_cross(n, A[i], B[j]);
		v3_normalize(n);
		if (v3_dot(n, t) < 0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
		double pa[3], pb[3];
		v3_copy(pa, pos1);
		v3_copy(pb, pos2);
		for (int k = 0; k < 3; k++) {
			if (k != i) {
				double sg = v3_dot(n, A[k]) > 0 ? 1.0 : -1.0;
				for (int q = 0; q < 3; q++) pa[q] += sg * size1[k] * A[k][q];
			}
			if (k != j) {
				double sg = v3_dot(n, B[k]) > 0 ? -1.0 : 1.0;
				for (int q = 0; q < 3; q++) pb[q] += sg * size2[k] * B[k][q];
			}
		}
		/* closest points of the lines pa + al A_i, pb + be B_j, clamped to the edges */
		double d[3];
		v3_sub(d, pb, pa);
		double uaub = C[i][j], q1 = v3_dot(A[i], d), q2 = -v3_dot(B[j], d), den = 1 - uaub * uaub;
		double al = 0, be = 0;
		if (den > 1e-12) {
			al = (q1 + uaub * q2) / den;
			be = (uaub * q1 + q2) / den;
		}
		al = clipd(al, -size1[i], size1[i]);
		be = clipd(be, -size2[j], size2[j]);
		double xa[3], xb[3];
		for (int q = 0; q < 3; q++) {
			xa[q] = pa[q] + al * A[i][q];
			xb[q] = pb[q] + be * B[j][q];
		}
		double dv[3];
		v3_sub(dv, xb, xa);
		c->dist = v3_dot(dv, n);
		if (c->dist > margin) return 0;
		memset(c->frame, 0, sizeof c->frame);
		v3_copy(c->frame, n);
		for (int q = 0; q < 3; q++) c->pos[q] = 0.5 * (xa[q] + xb[q]);
		return 1;
	}
	/* 2. face contact: reference box r (axis ax), incident box o */
	const int ref1 = code < 3, ax = ref1 ? code : code - 3;
	const double(*R)[3] = ref1 ? A : B, (*O)[3] = ref1 ? B : A;
	const double *pr = ref1 ? pos1 : pos2, *po = ref1 ? pos2 : pos1, *hr = ref1 ? size1 : size2, *ho = ref1 ? size2 : size1;
	double nref[3]; /* outward normal of the reference face, pointing to the incident box */
	{
		double sg = (ref1 ? tA[ax] : -tB[ax]) >= -MJO_TIE ? 1.0 : -1.0; /* (centres level along the axis: the + face, whatever the last bit says) */
		for (int q = 0; q < 3; q++) nref[q] = sg * R[ax][q];
	}
	int k = 0;
	double kbest = -1;
	for (int q = 0; q < 3; q++) {
		double a = fabs(v3_dot(O[q], nref));
		if (a > kbest + MJO_TIE) { kbest = a; k = q; }
	}
	const double fs = v3_dot(O[k], nref) > 0 ? -1.0 : 1.0; /* incident face normal = fs O_k (against nref) */
	const int u = (k + 1) % 3, v = (k + 2) % 3, sx = (ax + 1) % 3, sy = (ax + 2) % 3;
	double poly[8][3], tmp[8][3];
	int np = 4;
	for (int w = 0; w < 4; w++) {
		const double su = (w == 0 || w == 3) ? 1.0 : -1.0, sv = (w < 2) ? 1.0 : -1.0;
		double d[3];
		for (int q = 0; q < 3; q++) d[q] = po[q] + fs * ho[k] * O[k][q] + su * ho[u] * O[u][q] + sv * ho[v] * O[v][q] - pr[q];
		poly[w][0] = v3_dot(d, R[sx]);
		poly[w][1] = v3_dot(d, R[sy]);
		poly[w][2] = v3_dot(d, nref) - hr[ax];
	}
	for (int side = 0; side < 4; side++) {
		const int cax = side >> 1;
		const double sgn = (side & 1) ? -1.0 : 1.0, lim = cax == 0 ? hr[sx] : hr[sy];
		int nn = 0;
		for (int w = 0; w < np; w++) {
			const double *p0 = poly[w], *p1 = poly[(w + 1) % np];
			double d0 = sgn * p0[cax] - lim, d1 = sgn * p1[cax] - lim;
			/* (a vertex within MJO_TIE of the side plane is ON it: kept, and no crossing is generated next to it) */
			if (d0 <= MJO_TIE) {
				if (nn < 8) { memcpy(tmp[nn], p0, sizeof tmp[0]); nn++; }
			}
			if ((d0 < -MJO_TIE && d1 > MJO_TIE) || (d0 > MJO_TIE && d1 < -MJO_TIE)) {
				double f = d0 / (d0 - d1);
				if (nn < 8) {
					for (int q = 0; q < 3; q++) tmp[nn][q] = p0[q] + f * (p1[q] - p0[q]);
					nn++;
				}
			}
		}
		np = nn;
		memcpy(poly, tmp, sizeof poly);
		if (np == 0) return 0;
	}
	/* keep the vertices within margin of the reference face */
	int nk = 0;
	for (int w = 0; w < np; w++)
		if (poly[w][2] < margin) {
			memcpy(tmp[nk], poly[w], sizeof tmp[0]);
			nk++;
		}
	if (nk == 0) return 0;
	/* every clipped vertex within the margin is a contact: up to 8, as mjc_BoxBox returns */
	int pick[8], npick = 0;
	for (int w = 0; w < nk && w < 8; w++) pick[npick++] = w;
	/* geom1 -> geom2 normal */
	double nrm[3];
	for (int q = 0; q < 3; q++) nrm[q] = ref1 ? nref[q] : -nref[q];
	for (int w = 0; w < npick; w++) {
		const double *pv = tmp[pick[w]];
		rawcon *o = c + w;
		o->dist = pv[2];
		memset(o->frame, 0, sizeof o->frame);
		v3_copy(o->frame, nrm);
		/* vertex on the incident face, moved half way towards the reference face */
		const double hgt = hr[ax] + 0.5 * pv[2];
		for (int q = 0; q < 3; q++) o->pos[q] = pr[q] + pv[0] * R[sx][q] + pv[1] * R[sy][q] + hgt * nref[q];
	}
	return npick;
}

/* mj_contactParam: mix the two geoms' contact parameters */
static void contact_param(const mjb_model_desc *m, int g1, int g2, int *condim, double *solref, double *solimp,
                          double *friction)
{
	int p1 = m->geom_priority[g1], p2 = m->geom_priority[g2];
	if (p1 != p2) {
		int g = p1 > p2 ? g1 : g2;
		*condim = m->geom_condim[g];
		memcpy(solref, m->geom_solref + 2 * g, 2 * sizeof(double));
		memcpy(solimp, m->geom_solimp + 5 * g, 5 * sizeof(double));
		for (int k = 0; k < 3; k++) friction[k] = m->geom_friction[3 * g + k];
		return;
	}
	*condim = m->geom_condim[g1] > m->geom_condim[g2] ? m->geom_condim[g1] : m->geom_condim[g2];
	double s1 = m->geom_solmix[g1], s2 = m->geom_solmix[g2], mix;
	if (s1 >= MJO_MINVAL && s2 >= MJO_MINVAL) mix = s1 / (s1 + s2);
	else if (s1 < MJO_MINVAL && s2 < MJO_MINVAL) mix = 0.5;
	else if (s1 < MJO_MINVAL) mix = 0.0;
	else mix = 1.0;
	const double *r1 = m->geom_solref + 2 * g1, *r2 = m->geom_solref + 2 * g2;
	if (r1[0] > 0 && r2[0] > 0) {
		for (int k = 0; k < 2; k++) solref[k] = mix * r1[k] + (1 - mix) * r2[k];
	} else {
		for (int k = 0; k < 2; k++) solref[k] = fmin(r1[k], r2[k]);
	}
	for (int k = 0; k < 5; k++) solimp[k] = mix * m->geom_solimp[5 * g1 + k] + (1 - mix) * m->geom_solimp[5 * g2 + k];
	for (int k = 0; k < 3; k++) friction[k] = fmax(m->geom_friction[3 * g1 + k], m->geom_friction[3 * g2 + k]);
}

void mjo_set_geom_size(const mjb_model_desc *m, mjo_data *d, const double *size)
{
	free(d->env_geom_size);
	d->env_geom_size = NULL;
	if (size) {
		d->env_geom_size = (double *)malloc(sizeof(double) * 3 * (size_t)(m->ngeom > 0 ? m->ngeom : 1));
		memcpy(d->env_geom_size, size, sizeof(double) * 3 * (size_t)m->ngeom);
	}
}

void mjo_set_geom_type(const mjb_model_desc *m, mjo_data *d, const int *type)
{
	free(d->env_geom_type);
	d->env_geom_type = NULL;
	if (type) {
		d->env_geom_type = (int *)malloc(sizeof(int) * (size_t)(m->ngeom > 0 ? m->ngeom : 1));
		memcpy(d->env_geom_type, type, sizeof(int) * (size_t)m->ngeom);
	}
}

void mjo_register_collision(mjo_data *d, int geom_type1, int geom_type2, int func)
{
	int lo = geom_type1 < geom_type2 ? geom_type1 : geom_type2, hi = geom_type1 < geom_type2 ? geom_type2 : geom_type1;
	if (lo >= 0 && hi < 8) d->colfunc[8 * lo + hi] = func;
}

/* ------------------------------------------------------------------ A4+A5: mj_collision */
void mjo_collision(const mjb_model_desc *m, mjo_data *d)
{
	d->ncon[0] = 0;
	if (m->nconmax <= 0 || (m->disableflags & (MJB_DSBL_CONSTRAINT | MJB_DSBL_CONTACT))) return;
	int ncon = 0, overflow = 0;
	for (int p = 0; p < m->ncollpair; p++) {
		int g1 = m->collpair_geom[2 * p], g2 = m->collpair_geom[2 * p + 1];
		int t1 = d->env_geom_type ? d->env_geom_type[g1] : m->geom_type[g1], t2 = d->env_geom_type ? d->env_geom_type[g2] : m->geom_type[g2];
		if (t1 > t2) { /* a per-env type change reversed the (type1 <= type2) order of the pair */
			int tg = g1; g1 = g2; g2 = tg;
			int tt = t1; t1 = t2; t2 = tt;
		}
		const double *pos1 = d->geom_xpos + 3 * g1, *pos2 = d->geom_xpos + 3 * g2;
		const double *mat1 = d->geom_xmat + 9 * g1, *mat2 = d->geom_xmat + 9 * g2;
		const double *gsz = d->env_geom_size ? d->env_geom_size : m->geom_size;
		const double *size1 = gsz + 3 * g1, *size2 = gsz + 3 * g2;
		double margin = fmax(m->geom_margin[g1], m->geom_margin[g2]);
		double gap = fmax(m->geom_gap[g1], m->geom_gap[g2]);
		/* <contact><pair>: what the pair states replaces the geoms' mix (friction[5] solref[2] solimp[5] margin gap, NaN = not stated) */
		const double *pp = m->collpair_explicit[p] ? m->collpair_param + 14 * p : NULL;
		if (pp && (pp[12] == pp[12])) margin = pp[12];
		if (pp && (pp[13] == pp[13])) gap = pp[13];
		const double mt = margin + MJO_TIE; /* what the culls and the pair functions test against (MJO_TIE) */
		/* broad phase: bounding spheres (plane: signed distance of the other geom's sphere) */
		double rb1 = m->geom_rbound[g1], rb2 = m->geom_rbound[g2];
		if (rb1 > 0 && rb2 > 0) {
			double dv[3];
			v3_sub(dv, pos2, pos1);
			double bound = mt + rb1 + rb2;
			if (v3_dot(dv, dv) > bound * bound) continue;
		} else if (t1 == MJB_GEOM_PLANE && rb2 > 0) {
			double n[3] = { mat1[2], mat1[5], mat1[8] }, dv[3];
			v3_sub(dv, pos2, pos1);
			if (v3_dot(dv, n) > mt + rb2) continue;
		}
		rawcon rc[8];
		int n = 0;
		const int cfun = d->colfunc[8 * t1 + t2]; /* registerCollisionFunction override (pairs are stored with t1 <= t2) */
		if (cfun == MJB_COLFUNC_NONE) continue;
		if (cfun == MJB_COLFUNC_SPHERES) {
			if (t1 == MJB_GEOM_PLANE) n = raw_plane_sphere(rc, pos1, mat1, pos2, rb2, mt);
			else n = raw_sphere_sphere(rc, pos1, rb1, pos2, rb2, mt);
		} else
		if (t1 == MJB_GEOM_PLANE && t2 == MJB_GEOM_SPHERE) n = raw_plane_sphere(rc, pos1, mat1, pos2, size2[0], mt);
		else if (t1 == MJB_GEOM_PLANE && t2 == MJB_GEOM_CAPSULE) n = plane_capsule(rc, pos1, mat1, pos2, mat2, size2, mt);
		else if (t1 == MJB_GEOM_PLANE && t2 == MJB_GEOM_BOX) n = plane_box(rc, pos1, mat1, pos2, mat2, size2, mt);
		else if (t1 == MJB_GEOM_SPHERE && t2 == MJB_GEOM_SPHERE) n = raw_sphere_sphere(rc, pos1, size1[0], pos2, size2[0], mt);
		else if (t1 == MJB_GEOM_SPHERE && t2 == MJB_GEOM_CAPSULE) n = sphere_capsule(rc, pos1, size1[0], pos2, mat2, size2, mt);
		else if (t1 == MJB_GEOM_SPHERE && t2 == MJB_GEOM_BOX) n = sphere_box(rc, pos1, size1[0], pos2, mat2, size2, mt);
		else if (t1 == MJB_GEOM_CAPSULE && t2 == MJB_GEOM_CAPSULE) n = capsule_capsule(rc, pos1, mat1, size1, pos2, mat2, size2, mt);
		else if (t1 == MJB_GEOM_CAPSULE && t2 == MJB_GEOM_BOX) n = capsule_box(rc, pos1, mat1, size1, pos2, mat2, size2, mt);
		else if (t1 == MJB_GEOM_BOX && t2 == MJB_GEOM_BOX) n = box_box(rc, pos1, mat1, size1, pos2, mat2, size2, mt);
		if (n == 0) continue;
		int condim;
		double solref[2], solimp[5], fri[3];
		contact_param(m, g1, g2, &condim, solref, solimp, fri);
		double fri5[5] = { fri[0], fri[0], fri[1], fri[2], fri[2] };
		if (pp) {
			if (m->collpair_condim[p] > 0) condim = m->collpair_condim[p];
			if ((pp[0] == pp[0])) memcpy(fri5, pp, sizeof fri5);
			if ((pp[5] == pp[5])) memcpy(solref, pp + 5, sizeof solref);
			if ((pp[7] == pp[7])) memcpy(solimp, pp + 7, sizeof solimp);
		}
		for (int i = 0; i < n; i++) {  /* (the pair functions return only contacts with dist <= margin; mj_collideGeoms adds them all) */
			if (ncon >= m->nconmax) {  /* mj_addContact: full -> the contact is dropped, mjWARN_CONTACTFULL */
				overflow = 1;
				continue;
			}
			make_frame(rc[i].frame);
			d->contact_dist[ncon] = rc[i].dist;
			v3_copy(d->contact_pos + 3 * ncon, rc[i].pos);
			memcpy(d->contact_frame + 9 * ncon, rc[i].frame, 9 * sizeof(double));
			d->contact_includemargin[ncon] = margin - gap;
			memcpy(d->contact_friction + 5 * ncon, fri5, sizeof fri5);
			memcpy(d->contact_solref + 2 * ncon, solref, sizeof solref);
			memcpy(d->contact_solimp + 5 * ncon, solimp, sizeof solimp);
			d->contact_geom[2 * ncon] = g1;
			d->contact_geom[2 * ncon + 1] = g2;
			d->contact_dim[ncon] = condim;
			d->contact_efc_address[ncon] = -1;
			ncon++;
		}
	}
	d->ncon[0] = ncon;
	if (overflow) d->warning[MJB_WARN_CONTACTFULL]++;
}

/* ------------------------------------------------------------------ A6: mj_makeConstraint */
/* translational Jacobian row of a world point attached to `body`, projected on `dir`, ADDED with sign */
static void add_jac_point(const mjb_model_desc *m, const mjo_data *d, double *row, int body, const double *point,
                          const double *dir, double sign)
{
	while (body > 0 && m->body_dofnum[body] == 0) body = m->body_parentid[body];
	if (body == 0) return;
	double offset[3];
	v3_sub(offset, point, d->subtree_com + 3 * m->body_rootid[body]);
	for (int i = m->body_dofadr[body] + m->body_dofnum[body] - 1; i >= 0; i = m->dof_parentid[i]) {
		const double *cd = d->cdof + 6 * i;
		double jp[3];
		v3_cross(jp, cd, offset);
		v3_addto(jp, cd + 3);
		row[i] += sign * v3_dot(dir, jp);
	}
}

/* rotational Jacobian row projected on `dir` */
static void add_jac_rot(const mjb_model_desc *m, const mjo_data *d, double *row, int body, const double *dir, double sign)
{
	while (body > 0 && m->body_dofnum[body] == 0) body = m->body_parentid[body];
	if (body == 0) return;
	for (int i = m->body_dofadr[body] + m->body_dofnum[body] - 1; i >= 0; i = m->dof_parentid[i])
		row[i] += sign * v3_dot(dir, d->cdof + 6 * i);
}

/* getimpedance */
static void impedance(const double *solimp, double pos, double margin, double *imp, double *impP)
{
	if (solimp[0] == solimp[1] || solimp[2] <= MJO_MINVAL) {
		*imp = 0.5 * (solimp[0] + solimp[1]);
		*impP = 0;
		return;
	}
	double x = (pos - margin) / solimp[2], sgn = 1;
	if (x < 0) {
		x = -x;
		sgn = -1;
	}
	if (x >= 1 || x <= 0) {
		*imp = x >= 1 ? solimp[1] : solimp[0];
		*impP = 0;
		return;
	}
	double y, yP;
	if (solimp[4] == 1) {
		y = x;
		yP = 1;
	} else if (x <= solimp[3]) {
		double a = 1 / pow(solimp[3], solimp[4] - 1);
		y = a * pow(x, solimp[4]);
		yP = solimp[4] * a * pow(x, solimp[4] - 1);
	} else {
		double b = 1 / pow(1 - solimp[3], solimp[4] - 1);
		y = 1 - b * pow(1 - x, solimp[4]);
		yP = solimp[4] * b * pow(1 - x, solimp[4] - 1);
	}
	*imp = solimp[0] + y * (solimp[1] - solimp[0]);
	*impP = yP * sgn * (solimp[1] - solimp[0]) / solimp[2];
}

/* R, D, KBIP of one row (mj_makeImpedance) */
static void row_params_x(const mjb_model_desc *m, mjo_data *d, int i, const double *solref_in, const double *solimp_in,
                         double diag_approx, double imp_pos, double imp_margin)
{
	/* getsolparam: a mixed-sign solref is replaced by the default (0.02, 1); refsafe; solimp clamped to its legal ranges
	 * (mjMINIMP = 0.0001, mjMAXIMP = 0.9999, width >= 0, power >= 1) */
	double solref[2] = { solref_in[0], solref_in[1] };
	if ((solref[0] > 0) != (solref[1] > 0)) {
		solref[0] = 0.02;
		solref[1] = 1.0;
	}
	if (!(m->disableflags & MJB_DSBL_REFSAFE) && solref[0] > 0) solref[0] = fmax(solref[0], 2 * m->timestep[0]);
	const double solimp[5] = { fmin(0.9999, fmax(0.0001, solimp_in[0])), fmin(0.9999, fmax(0.0001, solimp_in[1])),
		                       fmax(0.0, solimp_in[2]), fmin(0.9999, fmax(0.0001, solimp_in[3])), fmax(1.0, solimp_in[4]) };
	double imp, impP;
	impedance(solimp, imp_pos, imp_margin, &imp, &impP);
	d->efc_R[i] = fmax(MJO_MINVAL, (1 - imp) * diag_approx / imp);
	double dmax = solimp[1], K, B;
	if (solref[0] > 0) {
		K = 1 / fmax(MJO_MINVAL, dmax * dmax * solref[0] * solref[0] * solref[1] * solref[1]);
		B = 2 / fmax(MJO_MINVAL, dmax * solref[0]);
	} else {
		K = -solref[0] / fmax(MJO_MINVAL, dmax * dmax);
		B = -solref[1] / fmax(MJO_MINVAL, dmax);
	}
	d->efc_KBIP[4 * i] = K;
	d->efc_KBIP[4 * i + 1] = B;
	d->efc_KBIP[4 * i + 2] = imp;
	d->efc_KBIP[4 * i + 3] = impP;
}

static void row_params(const mjb_model_desc *m, mjo_data *d, int i, const double *solref_in, const double *solimp,
                       double diag_approx)
{
	row_params_x(m, d, i, solref_in, solimp, diag_approx, d->efc_pos[i], d->efc_margin[i]);
}

/* res = quat * (0, axis)   (mju_mulQuatAxis) */
static void quat_mul_axis(double *res, const double *q, const double *a)
{
	res[0] = -q[1] * a[0] - q[2] * a[1] - q[3] * a[2];
	res[1] = q[0] * a[0] + q[2] * a[2] - q[3] * a[1];
	res[2] = q[0] * a[1] + q[3] * a[0] - q[1] * a[2];
	res[3] = q[0] * a[2] + q[1] * a[1] - q[2] * a[0];
}

/* mj_instantiateEquality (restated from memory of MuJoCo 2.3.7 engine_core_constraint.c -- the library is absent):
 * connect (3 rows: the anchor seen from both bodies must coincide), weld (6 rows: that plus the relative
 * orientation neg(q2) * q1 * relpose, axis part scaled by torquescale, with the Jacobian correction
 * 0.5 neg(q2) (w1 - w2) q1 relpose), joint (1 row: q1 - q1_0 = poly(q2 - q2_0)).  All rows of a connect / weld share
 * one impedance evaluated at the norm of the residual (getposdim).  Residual sign and Jacobian: body1 - body2. */
static int make_equality(const mjb_model_desc *m, mjo_data *d, int nefc, int *full)
{
	const int nv = m->nv;
	static const double unit[3][3] = { { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
	if (nv > 64) return nefc;
	for (int e = 0; e < m->neq; e++) {
		if (!m->eq_active[e]) continue;
		const double *data = m->eq_data + 11 * e;
		const int type = m->eq_type[e], id1 = m->eq_obj1id[e], id2 = m->eq_obj2id[e];
		double cpos[6] = { 0 }, jac[6][64], diag[6];
		int dim = 0;
		for (int k = 0; k < 6; k++) memset(jac[k], 0, sizeof(double) * (size_t)nv);
		if (type == MJB_EQ_CONNECT || type == MJB_EQ_WELD) {
			const int id[2] = { id1, id2 };
			double pos[2][3];
			for (int j = 0; j < 2; j++) {
				const double *anchor = type == MJB_EQ_CONNECT ? data + 3 * j : data + 3 * (1 - j);
				m3_mulvec(pos[j], d->xmat + 9 * id[j], anchor);
				v3_addto(pos[j], d->xpos + 3 * id[j]);
			}
			v3_sub(cpos, pos[0], pos[1]);
			for (int k = 0; k < 3; k++) {
				add_jac_point(m, d, jac[k], id[0], pos[0], unit[k], 1.0);
				add_jac_point(m, d, jac[k], id[1], pos[1], unit[k], -1.0);
			}
			const double tran = m->body_invweight0[2 * id1] + m->body_invweight0[2 * id2];
			const double rot = m->body_invweight0[2 * id1 + 1] + m->body_invweight0[2 * id2 + 1];
			diag[0] = diag[1] = diag[2] = tran;
			dim = 3;
			if (type == MJB_EQ_WELD) {
				const double ts = data[10];
				double quat[4], quat1[4], quat2[4];
				q_mul(quat, d->xquat + 4 * id[0], data + 6);
				quat1[0] = d->xquat[4 * id[1]];
				for (int k = 1; k < 4; k++) quat1[k] = -d->xquat[4 * id[1] + k];
				q_mul(quat2, quat1, quat);
				for (int k = 0; k < 3; k++) cpos[3 + k] = ts * quat2[1 + k];
				for (int k = 0; k < 3; k++) {
					add_jac_rot(m, d, jac[3 + k], id[0], unit[k], 1.0);
					add_jac_rot(m, d, jac[3 + k], id[1], unit[k], -1.0);
				}
				for (int i = 0; i < nv; i++) {
					const double axis[3] = { jac[3][i], jac[4][i], jac[5][i] };
					double qa[4], q3[4];
					quat_mul_axis(qa, quat1, axis);
					q_mul(q3, qa, quat);
					for (int k = 0; k < 3; k++) jac[3 + k][i] = 0.5 * ts * q3[1 + k];
				}
				diag[3] = diag[4] = diag[5] = rot;
				dim = 6;
			}
		} else if (type == MJB_EQ_JOINT) {
			const int a1 = m->jnt_qposadr[id1], d1 = m->jnt_dofadr[id1];
			double x = 0, deriv = 0, poly = data[0];
			if (id2 >= 0) {
				const int a2 = m->jnt_qposadr[id2];
				x = d->qpos[a2] - m->qpos0[a2];
				poly = data[0] + x * (data[1] + x * (data[2] + x * (data[3] + x * data[4])));
				deriv = data[1] + x * (2 * data[2] + x * (3 * data[3] + x * 4 * data[4]));
				jac[0][m->jnt_dofadr[id2]] = -deriv;
			}
			cpos[0] = d->qpos[a1] - m->qpos0[a1] - poly;
			jac[0][d1] += 1;
			diag[0] = m->dof_invweight0[d1] + (id2 >= 0 ? m->dof_invweight0[m->jnt_dofadr[id2]] : 0.0);
			dim = 1;
		} else if (type == MJB_EQ_TENDON) {
			/* (L1 - L1_0) - poly(L2 - L2_0) = 0 on tendon lengths; J = J1 - poly' J2 */
			double x = 0, deriv = 0, poly = data[0];
			if (id2 >= 0) {
				x = d->ten_length[id2] - m->tendon_length0[id2];
				poly = data[0] + x * (data[1] + x * (data[2] + x * (data[3] + x * data[4])));
				deriv = data[1] + x * (2 * data[2] + x * (3 * data[3] + x * 4 * data[4]));
				for (int w = m->tendon_adr[id2]; w < m->tendon_adr[id2] + m->tendon_num[id2]; w++)
					jac[0][m->jnt_dofadr[m->wrap_objid[w]]] -= deriv * m->wrap_prm[w];
			}
			for (int w = m->tendon_adr[id1]; w < m->tendon_adr[id1] + m->tendon_num[id1]; w++)
				jac[0][m->jnt_dofadr[m->wrap_objid[w]]] += m->wrap_prm[w];
			cpos[0] = d->ten_length[id1] - m->tendon_length0[id1] - poly;
			diag[0] = m->tendon_invweight0[id1] + (id2 >= 0 ? m->tendon_invweight0[id2] : 0.0);
			dim = 1;
		} else {
			continue;
		}
		if (nefc + dim > m->nefcmax) {
			*full = 1;
			break;
		}
		double nrm = 0;
		for (int k = 0; k < dim; k++) nrm += cpos[k] * cpos[k];
		nrm = sqrt(nrm);
		for (int k = 0; k < dim; k++) {
			memcpy(d->efc_J + (size_t)nefc * nv, jac[k], sizeof(double) * (size_t)nv);
			d->efc_pos[nefc] = cpos[k];
			d->efc_margin[nefc] = 0;
			d->efc_type[nefc] = MJB_CNSTR_EQUALITY;
			d->efc_id[nefc] = e;
			row_params_x(m, d, nefc, m->eq_solref + 2 * e, m->eq_solimp + 5 * e, diag[k], dim > 1 ? nrm : cpos[0], 0.0);
			nefc++;
		}
	}
	return nefc;
}

void mjo_make_constraint(const mjb_model_desc *m, mjo_data *d)
{
	int nv = m->nv, nefc = 0;
	/* Capacity rule (include/mjb.h, mjb_warning): items are taken in MuJoCo's row order -- equality, dof / tendon friction,
	 * joint limit (both sides of a joint are one item), tendon limit, contact; the first item whose rows do not fit in
	 * nefcmax and EVERY item after it are dropped, and mjWARN_CNSTRFULL is raised. */
	int full = 0;
	d->nefc[0] = 0;
	if (m->nefcmax <= 0 || (m->disableflags & MJB_DSBL_CONSTRAINT)) return;
	if (!(m->disableflags & MJB_DSBL_EQUALITY)) nefc = make_equality(m, d, nefc, &full);
	for (int i = 0; i < m->nefcmax; i++) d->efc_frictionloss[i] = 0;
	/* dry joint friction (mj_instantiateFriction, dof part): one row per dof with frictionloss > 0, J = e_dof,
	 * pos = margin = 0; the row's force is limited to +-frictionloss by the solvers */
	if (!(m->disableflags & MJB_DSBL_FRICTIONLOSS)) {
		for (int i = 0; i < nv; i++) {
			if (m->dof_frictionloss[i] <= 0 || full) continue;
			if (nefc >= m->nefcmax) {
				full = 1;
				continue;
			}
			double *row = d->efc_J + (size_t)nefc * nv;
			memset(row, 0, sizeof(double) * (size_t)nv);
			row[i] = 1;
			d->efc_pos[nefc] = 0;
			d->efc_margin[nefc] = 0;
			d->efc_frictionloss[nefc] = m->dof_frictionloss[i];
			d->efc_type[nefc] = MJB_CNSTR_FRICTION_DOF;
			d->efc_id[nefc] = i;
			row_params(m, d, nefc, m->dof_solref + 2 * i, m->dof_solimp + 5 * i, m->dof_invweight0[i]);
			nefc++;
		}
		/* tendon part: J = the tendon's moment arm row */
		for (int t = 0; t < m->ntendon; t++) {
			if (m->tendon_frictionloss[t] <= 0 || full) continue;
			if (nefc >= m->nefcmax) {
				full = 1;
				continue;
			}
			double *row = d->efc_J + (size_t)nefc * nv;
			memset(row, 0, sizeof(double) * (size_t)nv);
			for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++)
				row[m->jnt_dofadr[m->wrap_objid[w]]] += m->wrap_prm[w];
			d->efc_pos[nefc] = 0;
			d->efc_margin[nefc] = 0;
			d->efc_frictionloss[nefc] = m->tendon_frictionloss[t];
			d->efc_type[nefc] = MJB_CNSTR_FRICTION_TENDON;
			d->efc_id[nefc] = t;
			row_params(m, d, nefc, m->tendon_solref_fri + 2 * t, m->tendon_solimp_fri + 5 * t, m->tendon_invweight0[t]);
			nefc++;
		}
	}
	/* joint limits (mj_instantiateLimit): hinge / slide, and ball joints (the rotation angle against max(range)) */
	if (!(m->disableflags & MJB_DSBL_LIMIT)) {
		for (int j = 0; j < m->njnt; j++) {
			if (!m->jnt_limited[j] || m->jnt_type[j] == MJB_JNT_FREE) continue;
			if (m->jnt_type[j] == MJB_JNT_BALL) {
				/* axis-angle of the joint quaternion (mju_quat2Vel with dt = 1: the angle in (-pi, pi]), value = its norm, ONE row whose
				 * Jacobian is minus the unit axis on the joint's three dofs */
				double aa[3], margin = m->jnt_margin[j];
				double value = mjo_ball_angle(d->qpos + m->jnt_qposadr[j], aa);
				double r0 = m->jnt_range[2 * j], r1 = m->jnt_range[2 * j + 1];
				double dist = (r0 > r1 ? r0 : r1) - value;
				if (!(dist < margin) || full) continue;
				if (nefc + 1 > m->nefcmax) {
					full = 1;
					continue;
				}
				double *row = d->efc_J + (size_t)nefc * nv;
				memset(row, 0, sizeof(double) * (size_t)nv);
				for (int k = 0; k < 3; k++) row[m->jnt_dofadr[j] + k] = -aa[k];
				d->efc_pos[nefc] = dist;
				d->efc_margin[nefc] = margin;
				d->efc_type[nefc] = MJB_CNSTR_LIMIT_JOINT;
				d->efc_id[nefc] = j;
				row_params(m, d, nefc, m->jnt_solref + 2 * j, m->jnt_solimp + 5 * j, m->dof_invweight0[m->jnt_dofadr[j]]);
				nefc++;
				continue;
			}
			double value = d->qpos[m->jnt_qposadr[j]], margin = m->jnt_margin[j];
			int nside = (value - m->jnt_range[2 * j] < margin) + (m->jnt_range[2 * j + 1] - value < margin);
			if (nside == 0 || full) continue;
			if (nefc + nside > m->nefcmax) {
				full = 1;
				continue;
			}
			for (int side = -1; side <= 1; side += 2) {
				double dist = side * (m->jnt_range[2 * j + (side + 1) / 2] - value);
				if (dist < margin) {
					double *row = d->efc_J + (size_t)nefc * nv;
					memset(row, 0, sizeof(double) * (size_t)nv);
					row[m->jnt_dofadr[j]] = -side;
					d->efc_pos[nefc] = dist;
					d->efc_margin[nefc] = margin;
					d->efc_type[nefc] = MJB_CNSTR_LIMIT_JOINT;
					d->efc_id[nefc] = j;
					row_params(m, d, nefc, m->jnt_solref + 2 * j, m->jnt_solimp + 5 * j, m->dof_invweight0[m->jnt_dofadr[j]]);
					nefc++;
				}
			}
		}
	}
	/* tendon limits (mj_instantiateLimit, tendon part): rows after the joint limits */
	if (!(m->disableflags & MJB_DSBL_LIMIT)) {
		for (int t = 0; t < m->ntendon; t++) {
			if (!m->tendon_limited[t]) continue;
			double value = d->ten_length[t], margin = m->tendon_margin[t];
			int nside = (value - m->tendon_range[2 * t] < margin) + (m->tendon_range[2 * t + 1] - value < margin);
			if (nside == 0 || full) continue;
			if (nefc + nside > m->nefcmax) {
				full = 1;
				continue;
			}
			for (int side = -1; side <= 1; side += 2) {
				double dist = side * (m->tendon_range[2 * t + (side + 1) / 2] - value);
				if (dist < margin) {
					double *row = d->efc_J + (size_t)nefc * nv;
					memset(row, 0, sizeof(double) * (size_t)nv);
					for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++)
						row[m->jnt_dofadr[m->wrap_objid[w]]] += -side * m->wrap_prm[w];
					d->efc_pos[nefc] = dist;
					d->efc_margin[nefc] = margin;
					d->efc_type[nefc] = MJB_CNSTR_LIMIT_TENDON;
					d->efc_id[nefc] = t;
					row_params(m, d, nefc, m->tendon_solref_lim + 2 * t, m->tendon_solimp_lim + 5 * t, m->tendon_invweight0[t]);
					nefc++;
				}
			}
		}
	}
	/* contacts (mj_instantiateContact), frictionless or pyramidal */
	if (!(m->disableflags & MJB_DSBL_CONTACT)) {
		for (int c = 0; c < d->ncon[0]; c++) {
			if (!(d->contact_dist[c] < d->contact_includemargin[c])) continue;
			int dim = d->contact_dim[c];
			int nrow = dim == 1 ? 1 : (m->cone == MJB_CONE_ELLIPTIC ? dim : 2 * (dim - 1));
			if (full) break;
			if (nefc + nrow > m->nefcmax) {
				full = 1;
				break;
			}
			int b1 = m->geom_bodyid[d->contact_geom[2 * c]], b2 = m->geom_bodyid[d->contact_geom[2 * c + 1]];
			const double *frame = d->contact_frame + 9 * c, *pos = d->contact_pos + 3 * c, *fri = d->contact_friction + 5 * c;
			/* Jacobian difference (body2 - body1) in the contact frame: up to 6 rows */
			double jac[6][64];
			if (nv > 64) return; /* oracle capacity */
			for (int k = 0; k < dim && k < 6; k++) {
				memset(jac[k], 0, sizeof(double) * (size_t)nv);
				if (k < 3) {
					add_jac_point(m, d, jac[k], b2, pos, frame + 3 * k, 1.0);
					add_jac_point(m, d, jac[k], b1, pos, frame + 3 * k, -1.0);
				} else {
					add_jac_rot(m, d, jac[k], b2, frame + 3 * (k - 3), 1.0);
					add_jac_rot(m, d, jac[k], b1, frame + 3 * (k - 3), -1.0);
				}
			}
			double tran = m->body_invweight0[2 * b1] + m->body_invweight0[2 * b2];
			double rot = m->body_invweight0[2 * b1 + 1] + m->body_invweight0[2 * b2 + 1];
			d->contact_efc_address[c] = nefc;
			if (dim == 1) {
				memcpy(d->efc_J + (size_t)nefc * nv, jac[0], sizeof(double) * (size_t)nv);
				d->efc_pos[nefc] = d->contact_dist[c];
				d->efc_margin[nefc] = d->contact_includemargin[c];
				d->efc_type[nefc] = MJB_CNSTR_CONTACT_FRICTIONLESS;
				d->efc_id[nefc] = c;
				row_params(m, d, nefc, d->contact_solref + 2 * c, d->contact_solimp + 5 * c, tran);
				nefc++;
			} else if (m->cone == MJB_CONE_ELLIPTIC) {
				/* elliptic cone: row 0 = normal (pos = dist), rows 1..dim-1 = tangential / torsional / rolling
				 * directions (pos = margin = 0); regularisation R_j = R_0 mu^2 / friction_j^2, mu = friction_0 / sqrt(impratio) */
				int first = nefc;
				for (int k = 0; k < dim; k++) {
					memcpy(d->efc_J + (size_t)nefc * nv, jac[k], sizeof(double) * (size_t)nv);
					d->efc_pos[nefc] = k == 0 ? d->contact_dist[c] : 0.0;
					d->efc_margin[nefc] = k == 0 ? d->contact_includemargin[c] : 0.0;
					d->efc_type[nefc] = MJB_CNSTR_CONTACT_ELLIPTIC;
					d->efc_id[nefc] = c;
					row_params(m, d, nefc, d->contact_solref + 2 * c, d->contact_solimp + 5 * c, k < 3 ? tran : rot);
					nefc++;
				}
				double mu = fri[0] / sqrt(fmax(MJO_MINVAL, m->impratio[0]));
				for (int k = 1; k < dim; k++)
					d->efc_R[first + k] = fmax(MJO_MINVAL, d->efc_R[first] * mu * mu / (fri[k - 1] * fri[k - 1]));
			} else {
				int first = nefc;
				for (int k = 1; k < dim; k++)
					for (int s = 0; s < 2; s++) {
						double *row = d->efc_J + (size_t)nefc * nv;
						double f = s == 0 ? fri[k - 1] : -fri[k - 1];
						for (int i = 0; i < nv; i++) row[i] = jac[0][i] + f * jac[k][i];
						d->efc_pos[nefc] = d->contact_dist[c];
						d->efc_margin[nefc] = d->contact_includemargin[c];
						d->efc_type[nefc] = MJB_CNSTR_CONTACT_PYRAMIDAL;
						d->efc_id[nefc] = c;
						double da = tran + fri[k - 1] * fri[k - 1] * ((k - 1) < 2 ? tran : rot);
						row_params(m, d, nefc, d->contact_solref + 2 * c, d->contact_solimp + 5 * c, da);
						nefc++;
					}
				/* pyramidal regularisation: Rpy = 2 mu^2 R[first], mu = friction[0] / sqrt(impratio) */
				double mu = fri[0] / sqrt(fmax(MJO_MINVAL, m->impratio[0]));
				double Rpy = 2 * mu * mu * d->efc_R[first];
				for (int r = first; r < nefc; r++) d->efc_R[r] = fmax(MJO_MINVAL, Rpy);
			}
		}
	}
	for (int i = 0; i < nefc; i++) d->efc_D[i] = 1 / d->efc_R[i];
	d->nefc[0] = nefc;
	if (full) d->warning[MJB_WARN_CNSTRFULL]++;
}

/* A7: mj_projectConstraint: AR = J M^-1 J' + diag(R) */
void mjo_project_constraint(const mjb_model_desc *m, mjo_data *d)
{
	int nv = m->nv, nefc = d->nefc[0], ld = m->nefcmax;
	if (nefc == 0) return;
	double *tmp = d->scratch_nv2;
	for (int i = 0; i < nefc; i++) {
		memcpy(tmp, d->efc_J + (size_t)i * nv, sizeof(double) * (size_t)nv);
		mjo_solve_m(m, d, tmp);
		memcpy(d->efc_B + (size_t)i * nv, tmp, sizeof(double) * (size_t)nv);
		for (int j = 0; j < nefc; j++) {
			double s = 0;
			const double *rj = d->efc_J + (size_t)j * nv;
			for (int k = 0; k < nv; k++) s += rj[k] * tmp[k];
			d->efc_AR[(size_t)j * ld + i] = s;
		}
	}
	/* symmetrise exactly (the two triangles differ by rounding) and add R */
	for (int i = 0; i < nefc; i++) {
		for (int j = 0; j < i; j++) {
			double s = 0.5 * (d->efc_AR[(size_t)i * ld + j] + d->efc_AR[(size_t)j * ld + i]);
			d->efc_AR[(size_t)i * ld + j] = d->efc_AR[(size_t)j * ld + i] = s;
		}
		d->efc_AR[(size_t)i * ld + i] += d->efc_R[i];
	}
}

/* A8: mj_referenceConstraint */
void mjo_reference_constraint(const mjb_model_desc *m, mjo_data *d)
{
	int nv = m->nv, nefc = d->nefc[0];
	for (int i = 0; i < nefc; i++) {
		double s = 0;
		const double *row = d->efc_J + (size_t)i * nv;
		for (int k = 0; k < nv; k++) s += row[k] * d->qvel[k];
		d->efc_vel[i] = s;
		const double *kb = d->efc_KBIP + 4 * i;
		d->efc_aref[i] = -kb[1] * s - kb[0] * kb[2] * (d->efc_pos[i] - d->efc_margin[i]);
	}
}

static void fwd_constraint_newton(const mjb_model_desc *m, mjo_data *d);
static double constraint_update(const mjb_model_desc *m, const mjo_data *d, int nefc, const double *jar, double *force,
                                double *hrow, double *hcone);

/* A13: mj_fwdConstraint with the PGS solver (mj_solPGS); A14 (Newton) below */
/* Friction part of an elliptic-cone block (the role of mju_QCQP2 / 3 / N in mj_solPGS):
 *   minimise 0.5 y'A y + y'b  subject to  sum_j (y_j / d_j)^2 <= r^2          (n <= 5, A symmetric positive definite).
 * In the scaled variable z_j = y_j / d_j the constraint is the ball |z| <= r; if the free minimiser violates it the
 * multiplier la >= 0 of (A_s + la I) z = -b_s, |z| = r is found by Newton's method on phi(la) = |z|^2 - r^2
 * (phi' = -2 z'(A_s + la I)^-1 z, at most 20 steps, 1e-10 stopping thresholds).  Returns 1 when the constraint is active. */
static int chol_solve_small(int n, const double *A, double la, const double *rhs, double *x)
{
	double Lc[25];
	for (int i = 0; i < n; i++)
		for (int j = 0; j <= i; j++) {
			double sum = A[i * n + j] + (i == j ? la : 0.0);
			for (int k = 0; k < j; k++) sum -= Lc[i * n + k] * Lc[j * n + k];
			if (i == j) {
				if (sum < MJO_MINVAL) return 0;
				Lc[i * n + i] = sqrt(sum);
			} else {
				Lc[i * n + j] = sum / Lc[j * n + j];
			}
		}
	for (int i = 0; i < n; i++) {
		double sum = rhs[i];
		for (int k = 0; k < i; k++) sum -= Lc[i * n + k] * x[k];
		x[i] = sum / Lc[i * n + i];
	}
	for (int i = n - 1; i >= 0; i--) {
		double sum = x[i];
		for (int k = i + 1; k < n; k++) sum -= Lc[k * n + i] * x[k];
		x[i] = sum / Lc[i * n + i];
	}
	return 1;
}

static int cone_qcqp(int n, double *y, const double *A, const double *b, const double *dsc, double r)
{
	double As[25], bs[5], z[5], w[5], nb[5];
	for (int i = 0; i < n; i++) {
		bs[i] = b[i] * dsc[i];
		nb[i] = -bs[i];
		for (int j = 0; j < n; j++) As[i * n + j] = A[i * n + j] * dsc[i] * dsc[j];
	}
	double la = 0;
	int ok = 1;
	for (int iter = 0; iter < 20; iter++) {
		ok = chol_solve_small(n, As, la, nb, z);
		if (!ok) break;
		double val = -r * r;
		for (int i = 0; i < n; i++) val += z[i] * z[i];
		if (val < 1e-10) break;
		chol_solve_small(n, As, la, z, w);
		double deriv = 0;
		for (int i = 0; i < n; i++) deriv -= 2 * z[i] * w[i];
		const double delta = -val / deriv;
		if (delta < 1e-10) break;
		la += delta;
	}
	if (!ok) {
		for (int i = 0; i < n; i++) y[i] = 0;
		return 0;
	}
	for (int i = 0; i < n; i++) y[i] = z[i] * dsc[i];
	return la != 0;
}

/* One Gauss-Seidel update of an elliptic contact block (rows i .. i+dim-1) of the dual problem, the structure of
 * mj_solPGS: (1) with (almost) no normal force, a plain update of the normal row and zero friction; otherwise an exact
 * step along the current force ray, stopped where the normal force would turn negative; (2) with the normal force
 * fixed, the friction forces minimise the block cost inside the cone section sum (f_j / mu_j)^2 <= f_n^2.
 * res = residual of the block rows at the old forces, Ac = the block of AR.  Forces are updated in place. */
static void pgs_cone_block(int dim, double *fc, const double *res, const double *Ac, const double *mu)
{
	double old[6];
	for (int a = 0; a < dim; a++) old[a] = fc[a];
	if (fc[0] < MJO_MINVAL) {
		fc[0] -= res[0] / Ac[0];
		if (fc[0] < 0) fc[0] = 0;
		for (int a = 1; a < dim; a++) fc[a] = 0;
	} else {
		double denom = 0, vr = 0;
		for (int a = 0; a < dim; a++) {
			double sa = 0;
			for (int c = 0; c < dim; c++) sa += Ac[a * dim + c] * old[c];
			denom += old[a] * sa;
			vr += old[a] * res[a];
		}
		if (denom >= MJO_MINVAL) {
			double x = -vr / denom;
			if (fc[0] + x * old[0] < 0) x = -fc[0] / old[0];
			for (int a = 0; a < dim; a++) fc[a] += x * old[a];
		}
	}
	if (fc[0] < MJO_MINVAL) {
		for (int a = 1; a < dim; a++) fc[a] = 0;
		return;
	}
	/* friction sub-problem: constant part of the residual bc = res - Ac old, then b_f = bc_f + Ac_f0 f_n */
	const int n = dim - 1;
	double Af[25], bf[5], y[5];
	for (int a = 1; a < dim; a++) {
		double bc = res[a];
		for (int c = 0; c < dim; c++) bc -= Ac[a * dim + c] * old[c];
		bf[a - 1] = bc + Ac[a * dim] * fc[0];
		for (int c = 1; c < dim; c++) Af[(a - 1) * n + (c - 1)] = Ac[a * dim + c];
	}
	const int active = cone_qcqp(n, y, Af, bf, mu, fc[0]);
	if (active) { /* put the result exactly on the cone */
		double sn = 0;
		for (int a = 0; a < n; a++) sn += (y[a] / mu[a]) * (y[a] / mu[a]);
		sn = sqrt(sn);
		if (sn > MJO_MINVAL)
			for (int a = 0; a < n; a++) y[a] *= fc[0] / sn;
	}
	for (int a = 0; a < n; a++) fc[1 + a] = y[a];
}

void mjo_fwd_constraint(const mjb_model_desc *m, mjo_data *d)
{
	if (d->nefc[0] > 0 && (m->solver == MJB_SOL_NEWTON || m->solver == MJB_SOL_CG)) { /* both primal (mj_solPrimal) */
		fwd_constraint_newton(m, d);
		return;
	}
	int nv = m->nv, nefc = d->nefc[0], ld = m->nefcmax;
	if (nefc == 0) {
		memcpy(d->qacc, d->qacc_smooth, sizeof(double) * (size_t)nv);
		memcpy(d->qacc_warmstart, d->qacc_smooth, sizeof(double) * (size_t)nv);
		memset(d->qfrc_constraint, 0, sizeof(double) * (size_t)nv);
		d->solver_iter[0] = 0;
		return;
	}
	double *f = d->efc_force, *b = d->efc_b;
	/* b = J qacc_smooth - aref */
	for (int i = 0; i < nefc; i++) {
		double s = 0;
		const double *row = d->efc_J + (size_t)i * nv;
		for (int k = 0; k < nv; k++) s += row[k] * d->qacc_smooth[k];
		b[i] = s - d->efc_aref[i];
	}
	/* warmstart (engine_forward.c warmstart()): forces implied by qacc_warmstart, kept only if their dual
	 * cost 0.5 f'ARf + f'b is negative */
	int warm = !(m->disableflags & MJB_DSBL_WARMSTART);
	if (warm) {
		for (int i = 0; i < nefc; i++) {
			double jar = -d->efc_aref[i];
			const double *row = d->efc_J + (size_t)i * nv;
			for (int k = 0; k < nv; k++) jar += row[k] * d->qacc_warmstart[k];
			f[i] = (jar < 0 || d->efc_type[i] == MJB_CNSTR_EQUALITY) ? -d->efc_D[i] * jar : 0.0; /* limit / contact rows are one-sided */
			if (d->efc_type[i] == MJB_CNSTR_FRICTION_DOF || d->efc_type[i] == MJB_CNSTR_FRICTION_TENDON) { /* two-sided, saturating at +-frictionloss (mj_constraintUpdate) */
				const double fl = d->efc_frictionloss[i];
				f[i] = -d->efc_D[i] * jar;
				if (f[i] > fl) f[i] = fl;
				if (f[i] < -fl) f[i] = -fl;
			}
		}
		if (m->cone == MJB_CONE_ELLIPTIC) { /* cone blocks: the primal update gives the forces of whole contacts */
			double jar[nefc];
			for (int i = 0; i < nefc; i++) {
				jar[i] = -d->efc_aref[i];
				const double *row = d->efc_J + (size_t)i * nv;
				for (int k = 0; k < nv; k++) jar[i] += row[k] * d->qacc_warmstart[k];
			}
			constraint_update(m, d, nefc, jar, f, NULL, NULL);
		}
		double cost = 0;
		for (int i = 0; i < nefc; i++) {
			double s = 0;
			for (int j = 0; j < nefc; j++) s += d->efc_AR[(size_t)i * ld + j] * f[j];
			cost += 0.5 * f[i] * s + f[i] * b[i];
		}
		if (cost > 0) warm = 0;
	}
	if (!warm) memset(f, 0, sizeof(double) * (size_t)nefc);

	double scale = 1.0 / (m->meaninertia[0] * (nv > 1 ? nv : 1));
	double ARinv[nefc]; /* 1 / diag(AR) (mj_solPGS precomputes it) */
	for (int i = 0; i < nefc; i++) ARinv[i] = 1.0 / d->efc_AR[(size_t)i * ld + i];
	int iter = 0;
	while (iter < m->iterations) {
		double improvement = 0;
		for (int i = 0; i < nefc; i++) {
			if (d->efc_type[i] == MJB_CNSTR_CONTACT_ELLIPTIC) {
				const int con = d->efc_id[i], dim = d->contact_dim[con];
				double resb[6], Ac[36], oldf[6], change = 0;
				for (int a = 0; a < dim; a++) {
					resb[a] = b[i + a];
					for (int j = 0; j < nefc; j++) resb[a] += d->efc_AR[(size_t)(i + a) * ld + j] * f[j];
					oldf[a] = f[i + a];
					for (int c = 0; c < dim; c++) Ac[a * dim + c] = d->efc_AR[(size_t)(i + a) * ld + i + c];
				}
				pgs_cone_block(dim, f + i, resb, Ac, d->contact_friction + 5 * con);
				for (int a = 0; a < dim; a++) {
					const double da = f[i + a] - oldf[a];
					double sa = 0;
					for (int c = 0; c < dim; c++) sa += Ac[a * dim + c] * (f[i + c] - oldf[c]);
					change += 0.5 * da * sa + da * resb[a];
				}
				if (change > 1e-10) {
					for (int a = 0; a < dim; a++) f[i + a] = oldf[a];
					change = 0;
				}
				improvement -= change;
				i += dim - 1;
				continue;
			}
			double res = b[i];
			for (int j = 0; j < nefc; j++) res += d->efc_AR[(size_t)i * ld + j] * f[j];
			double old = f[i];
			double Aii = d->efc_AR[(size_t)i * ld + i];
			f[i] -= res * ARinv[i];
			if (d->efc_type[i] == MJB_CNSTR_FRICTION_DOF || d->efc_type[i] == MJB_CNSTR_FRICTION_TENDON) { /* box: |f| <= frictionloss */
				const double fl = d->efc_frictionloss[i];
				if (f[i] < -fl) f[i] = -fl;
				if (f[i] > fl) f[i] = fl;
			} else if (f[i] < 0 && d->efc_type[i] != MJB_CNSTR_EQUALITY) f[i] = 0;
			double delta = f[i] - old;
			double change = 0.5 * delta * delta * Aii + delta * res;
			if (change > 1e-10) {
				f[i] = old;
				change = 0;
			}
			improvement -= change;
		}
		improvement *= scale;
		iter++;
		if (improvement < m->tolerance[0]) break;
	}
	d->solver_iter[0] = iter;
	/* qfrc_constraint = J' f ;  qacc = qacc_smooth + M^-1 qfrc_constraint */
	for (int k = 0; k < nv; k++) {
		double s = 0;
		for (int i = 0; i < nefc; i++) s += d->efc_J[(size_t)i * nv + k] * f[i];
		d->qfrc_constraint[k] = s;
		d->qacc[k] = s;
	}
	mjo_solve_m(m, d, d->qacc);
	for (int k = 0; k < nv; k++) {
		d->qacc[k] += d->qacc_smooth[k];
		d->qacc_warmstart[k] = d->qacc[k];
	}
}


/* ------------------------------------------------------------------ A14: Newton (mj_solPrimal, flg_Newton) */
/* Primal problem: minimise over qacc  0.5 (a - a0)' M (a - a0) + sum_i s_i(J_i a - aref_i),
 * s_i(x) = 0.5 D_i x^2 for x < 0 (limit / pyramidal contact rows), 0 otherwise.
 * Hessian H = M + J' diag(D_i [active]) J is rebuilt and Cholesky-factorised every iteration (MuJoCo updates
 * it incrementally with rank-1 up/down-dates when rows change state: same matrix up to rounding).
 * Ma = M qacc and jaref = J qacc - aref are computed once and then moved along the search direction, as mj_solPrimal does. */

/* res = M * vec with M in qM layout (mj_mulM) */
static void mul_m(const mjb_model_desc *m, const mjo_data *d, double *res, const double *vec)
{
	for (int i = 0; i < m->nv; i++) res[i] = 0;
	for (int i = 0; i < m->nv; i++) {
		int adr = m->dof_Madr[i];
		res[i] += d->qM[adr] * vec[i];
		int j = m->dof_parentid[i];
		adr++;
		while (j >= 0) {
			res[i] += d->qM[adr] * vec[j];
			res[j] += d->qM[adr] * vec[i];
			adr++;
			j = m->dof_parentid[j];
		}
	}
}

typedef struct {
	double alpha, cost, deriv[2];
} lspoint;

/* Elliptic cone in the scaled space U_0 = mu x_0, U_j = friction_j x_j  (x = jar of the contact's rows):
 * N = U_0, T = |U_1..|.  With R_j = R_0 mu^2 / friction_j^2 the dual problem is isotropic there and the primal
 * cost is  s(x) = 0.5 (D_0 / mu^2) |Proj_C(-U)|^2,  C = { |v_t| <= mu v_0 }:
 *   top    (N >= mu T)     : s = 0
 *   bottom (mu N + T <= 0) : s = 0.5 sum_j D_j x_j^2                        (all rows quadratic)
 *   middle                 : s = 0.5 Dm (N - mu T)^2,  Dm = D_0 / (mu^2 (1 + mu^2))
 * (C1 across both boundaries; tests/test_oracle_contact.py checks it numerically). */
enum { ZONE_TOP = 0, ZONE_MIDDLE = 1, ZONE_BOTTOM = 2 };

static int cone_zone(double N, double T, double mu)
{
	if (N >= mu * T) return ZONE_TOP;
	if (mu * N + T <= 0) return ZONE_BOTTOM;
	return ZONE_MIDDLE;
}

typedef struct {
	int nefc, nv;
	const int *type, *id;
	const double *D, *jaref, *jv;
	const mjo_data *d;
	double impratio;
	double quadGauss[3];
} lsctx;

/* PrimalEval: cost and first two derivatives along the search line at alpha */
static void ls_eval(const lsctx *c, lspoint *p)
{
	const double a = p->alpha;
	double cost = a * a * c->quadGauss[2] + a * c->quadGauss[1] + c->quadGauss[0];
	double d1 = 2 * a * c->quadGauss[2] + c->quadGauss[1], d2 = 2 * c->quadGauss[2];
	for (int i = 0; i < c->nefc; i++) {
		if (c->type[i] != MJB_CNSTR_CONTACT_ELLIPTIC) {
			double x = c->jaref[i] + a * c->jv[i];
			if (c->type[i] == MJB_CNSTR_FRICTION_DOF || c->type[i] == MJB_CNSTR_FRICTION_TENDON) { /* Huber: quadratic inside |x| < R f, linear outside */
				const double fl = c->d->efc_frictionloss[i], rf = fl / c->D[i];
				if (x <= -rf) {
					cost += fl * (-0.5 * rf - x);
					d1 += -fl * c->jv[i];
				} else if (x >= rf) {
					cost += fl * (-0.5 * rf + x);
					d1 += fl * c->jv[i];
				} else {
					cost += 0.5 * c->D[i] * x * x;
					d1 += c->D[i] * x * c->jv[i];
					d2 += c->D[i] * c->jv[i] * c->jv[i];
				}
				continue;
			}
			if (x < 0 || c->type[i] == MJB_CNSTR_EQUALITY) {
				cost += 0.5 * c->D[i] * x * x;
				d1 += c->D[i] * x * c->jv[i];
				d2 += c->D[i] * c->jv[i] * c->jv[i];
			}
			continue;
		}
		const int con = c->id[i], dim = c->d->contact_dim[con];
		const double *fri = c->d->contact_friction + 5 * con;
		const double mu = fri[0] / sqrt(fmax(MJO_MINVAL, c->impratio));
		double N = mu * (c->jaref[i] + a * c->jv[i]), N1 = mu * c->jv[i], TT = 0, UV = 0, VV = 0;
		for (int j = 1; j < dim; j++) {
			double U = fri[j - 1] * (c->jaref[i + j] + a * c->jv[i + j]), V = fri[j - 1] * c->jv[i + j];
			TT += U * U;
			UV += U * V;
			VV += V * V;
		}
		double T = sqrt(TT);
		int zone = cone_zone(N, T, mu);
		if (zone == ZONE_BOTTOM) {
			for (int j = 0; j < dim; j++) {
				double x = c->jaref[i + j] + a * c->jv[i + j];
				cost += 0.5 * c->D[i + j] * x * x;
				d1 += c->D[i + j] * x * c->jv[i + j];
				d2 += c->D[i + j] * c->jv[i + j] * c->jv[i + j];
			}
		} else if (zone == ZONE_MIDDLE) {
			double Dm = c->D[i] / (mu * mu * (1 + mu * mu)), NmT = N - mu * T;
			double T1 = UV / T, T2 = (VV - T1 * T1) / T;
			cost += 0.5 * Dm * NmT * NmT;
			d1 += Dm * NmT * (N1 - mu * T1);
			d2 += Dm * ((N1 - mu * T1) * (N1 - mu * T1) - NmT * mu * T2);
		}
		i += dim - 1;
	}
	p->cost = cost;
	p->deriv[0] = d1;
	p->deriv[1] = d2 <= 0 ? MJO_MINVAL : d2;
}

/* PrimalSearch: exact 1-D minimisation (Newton steps, then bracketing with midpoint / Newton candidates) */
static double primal_search(const lsctx *c, double gtol, int maxlsiter)
{
	lspoint p0, p1, p2, pmid, p1n, p2n;
	int iter = 0;
	p0.alpha = 0;
	ls_eval(c, &p0);
	p1.alpha = p0.alpha - p0.deriv[0] / p0.deriv[1];
	ls_eval(c, &p1);
	if (p0.cost < p1.cost) p1 = p0;
	if (fabs(p1.deriv[0]) < gtol) return p1.alpha;
	int dir = p1.deriv[0] < 0 ? 1 : -1;
	int p2update = 0;
	p2 = p1;
	while (p1.deriv[0] * dir <= -gtol && iter < maxlsiter) {
		p2 = p1;
		p2update = 1;
		p1.alpha = p1.alpha - p1.deriv[0] / p1.deriv[1];
		ls_eval(c, &p1);
		iter++;
		if (fabs(p1.deriv[0]) < gtol) return p1.alpha;
	}
	if (iter >= maxlsiter || !p2update) return p1.alpha;
	/* bracketed: p1 and p2 have derivatives of opposite sign */
	p1n.alpha = p1.alpha - p1.deriv[0] / p1.deriv[1];
	p2n.alpha = p2.alpha - p2.deriv[0] / p2.deriv[1];
	while (iter < maxlsiter) {
		pmid.alpha = 0.5 * (p1.alpha + p2.alpha);
		ls_eval(c, &pmid);
		ls_eval(c, &p1n);
		ls_eval(c, &p2n);
		iter++;
		lspoint *cand[3] = { &p1n, &p2n, &pmid };
		lspoint *best = NULL;
		for (int k = 0; k < 3; k++)
			if (fabs(cand[k]->deriv[0]) < gtol && (!best || cand[k]->cost < best->cost)) best = cand[k];
		if (best) return best->alpha;
		int updated = 0;
		for (int k = 0; k < 3; k++) {
			lspoint *q = cand[k];
			double lo = p1.alpha < p2.alpha ? p1.alpha : p2.alpha, hi = p1.alpha < p2.alpha ? p2.alpha : p1.alpha;
			if (!(q->alpha > lo && q->alpha < hi)) continue;
			if ((q->deriv[0] < 0) == (p1.deriv[0] < 0)) p1 = *q;
			else p2 = *q;
			updated = 1;
		}
		if (!updated) break;
		p1n.alpha = p1.alpha - p1.deriv[0] / p1.deriv[1];
		p2n.alpha = p2.alpha - p2.deriv[0] / p2.deriv[1];
	}
	return p1.cost < p2.cost ? p1.alpha : p2.alpha;
}

/* cost, force (= -ds/dx) and optionally the dim x dim Hessian block (row stride 6) of one elliptic contact */
static double cone_eval(double mu, const double *fri, const double *D, int dim, const double *jar, double *force, double *Hc)
{
	double U[6], TT = 0, cost = 0;
	U[0] = mu * jar[0];
	for (int j = 1; j < dim; j++) {
		U[j] = fri[j - 1] * jar[j];
		TT += U[j] * U[j];
	}
	const double N = U[0], T = sqrt(TT);
	const int zone = cone_zone(N, T, mu);
	if (Hc) memset(Hc, 0, 36 * sizeof(double));
	if (zone == ZONE_TOP) {
		for (int j = 0; j < dim; j++) force[j] = 0;
	} else if (zone == ZONE_BOTTOM) {
		for (int j = 0; j < dim; j++) {
			force[j] = -D[j] * jar[j];
			cost += 0.5 * D[j] * jar[j] * jar[j];
			if (Hc) Hc[j * 6 + j] = D[j];
		}
	} else {
		const double Dm = D[0] / (mu * mu * (1 + mu * mu)), NmT = N - mu * T;
		cost += 0.5 * Dm * NmT * NmT;
		force[0] = -Dm * NmT * mu;
		for (int j = 1; j < dim; j++) force[j] = -force[0] / T * U[j] * fri[j - 1];
		if (Hc) {
			/* g = d(NmT)/dx: g_0 = mu, g_j = -mu friction_j U_j / T;  H = Dm (g g' + NmT d g/dx) */
			double g[6];
			g[0] = mu;
			for (int j = 1; j < dim; j++) g[j] = -mu * fri[j - 1] * U[j] / T;
			for (int j = 0; j < dim; j++)
				for (int k = 0; k < dim; k++) Hc[j * 6 + k] = Dm * g[j] * g[k];
			for (int j = 1; j < dim; j++)
				for (int k = 1; k < dim; k++)
					Hc[j * 6 + k] += -Dm * NmT * mu * fri[j - 1] * fri[k - 1] * ((j == k ? 1.0 / T : 0.0) - U[j] * U[k] / (T * T * T));
		}
	}
	return cost;
}

/* test hook: the cone function by itself (D_j = D_0 friction_j^2 / mu^2 as make_constraint sets it) */
double mjo_debug_cone(const double *friction5, double impratio, double D0, int dim, const double *jar, double *force,
                      double *H36)
{
	double mu = friction5[0] / sqrt(fmax(MJO_MINVAL, impratio)), D[6];
	D[0] = D0;
	for (int j = 1; j < dim; j++) D[j] = D0 * friction5[j - 1] * friction5[j - 1] / (mu * mu);
	return cone_eval(mu, friction5, D, dim, jar, force, H36);
}

/* constraint update at jar: forces, constraint cost, and the per-row / per-cone Hessian weights:
 * hrow[i] = D_i for an active scalar row (else 0); for an elliptic contact starting at row i, hcone[36*con..]
 * holds the dim x dim block of d2s/dx2 (zero in the top zone, diag(D) in the bottom zone). */
static double constraint_update(const mjb_model_desc *m, const mjo_data *d, int nefc, const double *jar, double *force,
                                double *hrow, double *hcone)
{
	double cost = 0;
	for (int i = 0; i < nefc; i++) {
		if (d->efc_type[i] == MJB_CNSTR_FRICTION_DOF || d->efc_type[i] == MJB_CNSTR_FRICTION_TENDON) {
			const double fl = d->efc_frictionloss[i], rf = fl / d->efc_D[i], x = jar[i];
			if (x <= -rf) {
				force[i] = fl;
				cost += fl * (-0.5 * rf - x);
				if (hrow) hrow[i] = 0;
			} else if (x >= rf) {
				force[i] = -fl;
				cost += fl * (-0.5 * rf + x);
				if (hrow) hrow[i] = 0;
			} else {
				force[i] = -d->efc_D[i] * x;
				cost += 0.5 * d->efc_D[i] * x * x;
				if (hrow) hrow[i] = d->efc_D[i];
			}
			continue;
		}
		if (d->efc_type[i] != MJB_CNSTR_CONTACT_ELLIPTIC) {
			if (jar[i] < 0 || d->efc_type[i] == MJB_CNSTR_EQUALITY) {
				force[i] = -d->efc_D[i] * jar[i];
				cost += 0.5 * d->efc_D[i] * jar[i] * jar[i];
				if (hrow) hrow[i] = d->efc_D[i];
			} else {
				force[i] = 0;
				if (hrow) hrow[i] = 0;
			}
			continue;
		}
		const int con = d->efc_id[i], dim = d->contact_dim[con];
		const double *fri = d->contact_friction + 5 * con;
		const double mu = fri[0] / sqrt(fmax(MJO_MINVAL, m->impratio[0]));
		cost += cone_eval(mu, fri, d->efc_D + i, dim, jar + i, force + i, hcone ? hcone + 36 * con : NULL);
		if (hrow)
			for (int j = 0; j < dim; j++) hrow[i + j] = 0;
		i += dim - 1;
	}
	return cost;
}

static void fwd_constraint_newton(const mjb_model_desc *m, mjo_data *d)
{
	const int nv = m->nv, nefc = d->nefc[0], ncon = d->ncon[0];
	const double tol = m->tolerance[0], ls_tol = 0.01; /* mjOption.ls_tolerance default */
	const int ls_iter = 50;                               /* mjOption.ls_iterations default */
	const double scale = 1.0 / (m->meaninertia[0] * (nv > 1 ? nv : 1));
	double qacc[nv], Ma[nv], jaref[nefc], grad[nv], search[nv], Mv[nv], jv[nefc], hrow[nefc], H[nv * nv];
	double hcone[36 * (ncon > 0 ? ncon : 1)], tmpf[nefc];
	double *f = d->efc_force;
	const int cg = m->solver == MJB_SOL_CG; /* conjugate gradient: same cost, line search and stopping rules; the search
	                                         * direction is Polak-Ribiere with the preconditioner M^-1 instead of -H^-1 grad */
	double Mgrad[nv], gradold[nv], Mgradold[nv];

	/* warmstart: qacc_warmstart unless qacc_smooth has the lower cost (engine_forward.c warmstart()).  Ma = M qacc and
	 * jaref = J qacc - aref of the chosen start are kept: mj_solPrimal computes them once and from then on moves them along the
	 * search direction (Ma += alpha Mv, jaref += alpha jv below) instead of multiplying again every iteration */
	double best = 0;
	for (int pass = 0; pass < 2; pass++) {
		const double *q0 = pass == 0 ? d->qacc_warmstart : d->qacc_smooth;
		double tmp[nv], x0[nefc > 0 ? nefc : 1], cost = 0;
		mul_m(m, d, tmp, q0);
		for (int k = 0; k < nv; k++) cost += 0.5 * (tmp[k] - d->qfrc_smooth[k]) * (q0[k] - d->qacc_smooth[k]);
		for (int i = 0; i < nefc; i++) {
			double x = -d->efc_aref[i];
			for (int k = 0; k < nv; k++) x += d->efc_J[(size_t)i * nv + k] * q0[k];
			x0[i] = x;
		}
		cost += constraint_update(m, d, nefc, x0, tmpf, NULL, NULL);
		int take = 1;
		if (pass == 0) best = (m->disableflags & MJB_DSBL_WARMSTART) ? 1e300 : cost;
		else take = cost < best;
		if (take) {
			memcpy(qacc, q0, sizeof qacc);
			memcpy(Ma, tmp, sizeof Ma);
			memcpy(jaref, x0, nefc * sizeof(double));
		}
	}

	double cost = 0, prev_cost;
	int iter = 0;
	for (;;) {
		/* constraint update at the current Ma / jaref (forces, cost, Hessian weights), gradient */
		double gauss = 0;
		for (int k = 0; k < nv; k++) gauss += 0.5 * (Ma[k] - d->qfrc_smooth[k]) * (qacc[k] - d->qacc_smooth[k]);
		prev_cost = cost;
		cost = gauss + constraint_update(m, d, nefc, jaref, f, hrow, hcone);
		for (int k = 0; k < nv; k++) {
			double s = 0;
			for (int i = 0; i < nefc; i++) s += d->efc_J[(size_t)i * nv + k] * f[i];
			d->qfrc_constraint[k] = s;
			grad[k] = Ma[k] - d->qfrc_smooth[k] - s;
		}
		if (iter > 0) {
			double improvement = scale * (prev_cost - cost), gnorm = 0;
			for (int k = 0; k < nv; k++) gnorm += grad[k] * grad[k];
			gnorm = scale * sqrt(gnorm);
			if (improvement < tol || gnorm < tol || iter >= m->iterations) break;
		}
		if (cg) {
			memcpy(Mgrad, grad, sizeof Mgrad);
			mjo_solve_m(m, d, Mgrad);
			if (iter == 0) {
				for (int k = 0; k < nv; k++) search[k] = -Mgrad[k];
			} else {
				double num = 0, den = 0;
				for (int k = 0; k < nv; k++) {
					num += grad[k] * (Mgrad[k] - Mgradold[k]);
					den += gradold[k] * Mgradold[k];
				}
				double beta = num / fmax(MJO_MINVAL, den);
				if (beta < 0) beta = 0;
				for (int k = 0; k < nv; k++) search[k] = -Mgrad[k] + beta * search[k];
			}
			memcpy(gradold, grad, sizeof gradold);
			memcpy(Mgradold, Mgrad, sizeof Mgradold);
			goto linesearch;
		}
		/* Hessian H = M + J' W J  (W: D on active scalar rows, cone blocks on elliptic contacts) */
		memset(H, 0, sizeof H);
		for (int i = 0; i < nv; i++) {
			int adr = m->dof_Madr[i];
			for (int j = i; j >= 0; j = m->dof_parentid[j]) {
				H[i * nv + j] = H[j * nv + i] = d->qM[adr];
				adr++;
			}
		}
		for (int i = 0; i < nefc; i++) {
			const double *row = d->efc_J + (size_t)i * nv;
			if (d->efc_type[i] != MJB_CNSTR_CONTACT_ELLIPTIC) {
				if (hrow[i] == 0) continue;
				for (int r = 0; r < nv; r++) {
					double dr = hrow[i] * row[r];
					if (dr == 0) continue;
					for (int c2 = 0; c2 <= r; c2++) H[r * nv + c2] += dr * row[c2];
				}
				continue;
			}
			const int con = d->efc_id[i], dim = d->contact_dim[con];
			const double *Hc = hcone + 36 * con;
			for (int a = 0; a < dim; a++)
				for (int b2 = 0; b2 < dim; b2++) {
					double w = Hc[a * 6 + b2];
					if (w == 0) continue;
					const double *ra = d->efc_J + (size_t)(i + a) * nv, *rb = d->efc_J + (size_t)(i + b2) * nv;
					for (int r = 0; r < nv; r++) {
						double dr = w * ra[r];
						if (dr == 0) continue;
						for (int c2 = 0; c2 <= r; c2++) H[r * nv + c2] += dr * rb[c2];
					}
				}
			i += dim - 1;
		}
		for (int r = 0; r < nv; r++)
			for (int c2 = 0; c2 < r; c2++) H[c2 * nv + r] = H[r * nv + c2];
		/* mju_cholFactor (in place, lower) */
		for (int j = 0; j < nv; j++) {
			double s = H[j * nv + j];
			for (int k = 0; k < j; k++) s -= H[j * nv + k] * H[j * nv + k];
			if (s < MJO_MINVAL) s = MJO_MINVAL;
			double ljj = sqrt(s);
			H[j * nv + j] = ljj;
			for (int i = j + 1; i < nv; i++) {
				double t = H[i * nv + j];
				for (int k = 0; k < j; k++) t -= H[i * nv + k] * H[j * nv + k];
				H[i * nv + j] = t / ljj;
			}
		}
		for (int i = 0; i < nv; i++) {
			double t = grad[i];
			for (int k = 0; k < i; k++) t -= H[i * nv + k] * search[k];
			search[i] = t / H[i * nv + i];
		}
		for (int i = nv - 1; i >= 0; i--) {
			double t = search[i];
			for (int k = i + 1; k < nv; k++) t -= H[k * nv + i] * search[k];
			search[i] = t / H[i * nv + i];
		}
		for (int k = 0; k < nv; k++) search[k] = -search[k];
	linesearch:;
		/* line search */
		double snorm = 0;
		for (int k = 0; k < nv; k++) snorm += search[k] * search[k];
		snorm = sqrt(snorm);
		if (snorm < MJO_MINVAL) break;
		mul_m(m, d, Mv, search);
		lsctx c = { nefc, nv, d->efc_type, d->efc_id, d->efc_D, jaref, jv, d, m->impratio[0], { gauss, 0, 0 } };
		for (int k = 0; k < nv; k++) {
			c.quadGauss[1] += search[k] * (Ma[k] - d->qfrc_smooth[k]);
			c.quadGauss[2] += 0.5 * search[k] * Mv[k];
		}
		for (int i = 0; i < nefc; i++) {
			double s = 0;
			for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)i * nv + k] * search[k];
			jv[i] = s;
		}
		double gtol = tol * ls_tol * snorm / scale;
		double alpha = primal_search(&c, gtol, ls_iter);
		if (alpha == 0) break;
		for (int k = 0; k < nv; k++) {
			qacc[k] += alpha * search[k];
			Ma[k] += alpha * Mv[k];
		}
		for (int i = 0; i < nefc; i++) jaref[i] += alpha * jv[i];
		iter++;
	}
	d->solver_iter[0] = iter;
	memcpy(d->qacc, qacc, sizeof qacc);
	memcpy(d->qacc_warmstart, qacc, sizeof qacc);
}
