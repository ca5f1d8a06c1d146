// mjb_constraint.h — contact / limit constraint path of the step kernel (included by mjb_step.hip).
//
// SURVEY.md §8a rows A4-A7 and A13 for the primitive subset of include/mjb.h: plane / sphere / capsule /
// box geoms (pairs of collpair_geom), hinge / slide joint limits, frictionless + pyramidal contacts, PGS.
// ONE env per wavefront (G = 64): constraint rows map to lanes (nefcmax <= 64), the dual variables live in
// registers and are exchanged with v_readlane, so the Gauss-Seidel sweep touches LDS only for the two
// nv-long rows J_i and B_i = (M^-1 J')_i it needs per row.  The solver is AR-free: the residual
// A_i f + b_i is evaluated as  J_i . w + R_i f_i + b_i  with  w = M^-1 J' f  kept per lane (lane k holds
// w[k]) and updated by  w += B_i * delta  -- O(nv) per row instead of O(nefc), and no nefc^2 matrix in LDS.
#pragma once

struct RawCon {
	double dist, pos[3], frame[6];  // normal, first-tangent hint (zero: none)
};

DEVI void make_frame(double *fr)  // fr[9]: normal, tangent hint -> orthonormal frame (mju_makeFrame)
{
	normalize3(fr);
	if (sqrt(dot3(fr + 3, fr + 3)) < 0.5) {
		fr[3] = fr[4] = fr[5] = 0;
		if (fr[1] < 0.5 && fr[1] > -0.5) fr[4] = 1;
		else fr[5] = 1;
	}
	const double t = dot3(fr, fr + 3);
	fr[3] -= t * fr[0]; fr[4] -= t * fr[1]; fr[5] -= t * fr[2];
	normalize3(fr + 3);
	cross3(fr + 6, fr, fr + 3);
}

DEVI int raw_sphere_sphere(RawCon &c, const double *p1, double r1, const double *p2, double r2, double margin)
{
	double dif[3] = { p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2] };
	const double cdist = sqrt(dot3(dif, dif));
	if (cdist > margin + r1 + r2) return 0;
	c.dist = cdist - r1 - r2;
	c.frame[0] = dif[0]; c.frame[1] = dif[1]; c.frame[2] = dif[2];
	c.frame[3] = c.frame[4] = c.frame[5] = 0;
	normalize3(c.frame);
	for (int k = 0; k < 3; k++) c.pos[k] = p1[k] + c.frame[k] * (r1 + 0.5 * c.dist);
	return 1;
}

// Ties (oracle/mjo_constraint.c, MJO_TIE: the same offset at the same comparisons).  A scene built on a grid puts the narrow phase's comparisons on their knife
// edge -- the least penetrated of two equal face axes, a vertex ON a side plane, a surface at distance == margin -- and the last bit decides them differently in
// two implementations of the same steps.  Equal candidates keep the first in order, a point within MJB_TIE of a plane is on it, a distance equal to the margin is
// inside it.
#define MJB_TIE 1e-12
DEVI int raw_plane_sphere(RawCon &c, const double *pos1, const double *n, const double *p, double r, double margin)
{
	double tmp[3] = { p[0] - pos1[0], p[1] - pos1[1], p[2] - pos1[2] };
	const double cdist = dot3(tmp, n);
	if (cdist > margin + r) return 0;
	c.dist = cdist - r;
	c.frame[0] = n[0]; c.frame[1] = n[1]; c.frame[2] = n[2];
	c.frame[3] = c.frame[4] = c.frame[5] = 0;
	for (int k = 0; k < 3; k++) c.pos[k] = p[k] - n[k] * (r + 0.5 * c.dist);
	return 1;
}

DEVI double clipd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

// append `c` to a pair's contact list that holds `n` (0 or 1) entries -- written as selects on both slots: `rc[n] = c`, or a branch
// the optimiser turns into it, is a dynamically indexed store, which moves ALL of rc[] from registers to scratch memory
DEVI void rc_append(RawCon *rc, int n, const RawCon &c)
{
	const bool first = n == 0;
	rc[0].dist = first ? c.dist : rc[0].dist;
	rc[1].dist = first ? rc[1].dist : c.dist;
#pragma unroll
	for (int k = 0; k < 3; k++) {
		rc[0].pos[k] = first ? c.pos[k] : rc[0].pos[k];
		rc[1].pos[k] = first ? rc[1].pos[k] : c.pos[k];
	}
#pragma unroll
	for (int k = 0; k < 6; k++) {
		rc[0].frame[k] = first ? c.frame[k] : rc[0].frame[k];
		rc[1].frame[k] = first ? rc[1].frame[k] : c.frame[k];
	}
}

DEVI int raw_sphere_box(RawCon &c, const double *pos1, double r1, const double *pos2, const double *mat2,
                        const double *size2, double margin)
{
	const double tmp[3] = { pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2] };
	double center[3], clamped[3], dv[3];
	matTvec3(center, mat2, tmp);
	for (int i = 0; i < 3; i++) {
		clamped[i] = clipd(center[i], -size2[i], size2[i]);
		dv[i] = clamped[i] - center[i];
	}
	const double dist = sqrt(dot3(dv, dv));
	if (dist - r1 > margin) return 0;
	double nloc[3] = { 0, 0, 0 }, ploc[3];
	if (dist <= MJB_MINVAL) {
		double closest = 2 * fmax(size2[0], fmax(size2[1], size2[2]));
		int kk = 0;
		for (int i = 0; i < 6; i++) {
			const double fd = fabs(((i % 2) ? 1 : -1) * size2[i / 2] - center[i / 2]);
			if (closest > fd + MJB_TIE) {
				closest = fd;
				kk = i;
			}
		}
		const double sgn = (kk % 2) ? -1.0 : 1.0;
		nloc[0] = (kk / 2 == 0) ? sgn : 0.0; nloc[1] = (kk / 2 == 1) ? sgn : 0.0; nloc[2] = (kk / 2 == 2) ? sgn : 0.0;
		for (int i = 0; i < 3; i++) ploc[i] = center[i] + nloc[i] * (r1 - closest) / 2;
		c.dist = -closest - r1;
	} else {
		for (int i = 0; i < 3; i++) {
			const double deepest = center[i] + dv[i] * (r1 / dist);
			ploc[i] = 0.5 * (clamped[i] + deepest);
			nloc[i] = dv[i] / dist;
		}
		c.dist = dist - r1;
	}
	matvec3(c.frame, mat2, nloc);
	c.frame[3] = c.frame[4] = c.frame[5] = 0;
	matvec3(c.pos, mat2, ploc);
	c.pos[0] += pos2[0]; c.pos[1] += pos2[1]; c.pos[2] += pos2[2];
	return 1;
}

// capsule - box: same steps as oracle/mjo_constraint.c capsule_box (see the derivation there): minimiser set of
// the convex axis-to-box distance (the oracle: by bisection on its slope; here: exactly, from the slope's breakpoints), candidate axis points from the closest feature
// (face: ends of the stretch over the face; inside: ends of the inside stretch; edge / vertex: the minimiser
// set), each candidate through the sphere-box contact
#define MJB_CAPBOX_PAR 1e-12  // a direction component below this counts as parallel to that face pair (oracle: MJO_CAPBOX_PAR)
DEVI double capbox_slope(const double *p0, const double *d, const double *s, double t)
{
	double g = 0;
	for (int i = 0; i < 3; i++) {
		const double p = p0[i] + t * d[i];
		double r = p - clipd(p, -s[i], s[i]);
		// (t may BE the crossing of this face plane, computed by a division: p then misses +-s by a rounding, and on an axis parallel to the other
		//  two face pairs that 1e-17 is the whole slope -- its sign decided between the start, the middle and the end of a flat minimiser set: one
		//  contact where the oracle's bisection finds the set's two ends.  A residual of a few ulps of the half size is the plane itself.)
		if (fabs(r) <= 4e-15 * s[i]) r = 0;
		g += r * (fabs(d[i]) <= MJB_CAPBOX_PAR ? 0.0 : d[i]);
	}
	return g;
}

DEVI int capsule_box(RawCon *rc, const double *pos1, const double *mat1, const double *size1, const double *pos2,
                     const double *mat2, const double *size2, double margin)
{
	const double r = size1[0], h = size1[1];
	const double axis[3] = { mat1[2], mat1[5], mat1[8] };
	const double tmp[3] = { pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2] };
	double p0[3], d[3];
	matTvec3(p0, mat2, tmp);
	matTvec3(d, mat2, axis);
	// The slope of the (convex) squared axis-to-box distance is piecewise linear and non-decreasing in t, with breakpoints where a
	// coordinate of the axis point crosses a face plane: its zero set [tlo, thi] = [inf{slope >= 0}, sup{slope <= 0}] follows from
	// the slope at the six sorted breakpoints and the two ends -- eight INDEPENDENT evaluations and one interpolation -- where the
	// oracle bisects twice (120 dependent evaluations, ~8 k cycles on every capsule near the cube; same set up to rounding).
	double tk[8], gk[8];
	tk[0] = -h;
	tk[7] = h;
#pragma unroll
	for (int i = 0; i < 3; i++) {
		const bool par = fabs(d[i]) <= MJB_CAPBOX_PAR;  // (axis parallel to the face pair: no crossing)
		const double dd = par ? 1.0 : d[i];
		const double a = (-size2[i] - p0[i]) / dd, b = (size2[i] - p0[i]) / dd;
		tk[1 + 2 * i] = par ? h : clipd(a, -h, h);
		tk[2 + 2 * i] = par ? h : clipd(b, -h, h);
	}
	{
#define MJB_CE(i, j) { const double lo_ = fmin(tk[i], tk[j]), hi_ = fmax(tk[i], tk[j]); tk[i] = lo_; tk[j] = hi_; }
		MJB_CE(1, 6) MJB_CE(2, 4) MJB_CE(3, 5) MJB_CE(2, 3) MJB_CE(4, 5) MJB_CE(1, 4) MJB_CE(3, 6) MJB_CE(1, 2) MJB_CE(3, 4) MJB_CE(5, 6) MJB_CE(2, 3) MJB_CE(4, 5)
#undef MJB_CE
	}
#pragma unroll
	for (int k = 0; k < 8; k++) gk[k] = capbox_slope(p0, d, size2, tk[k]);
	const double glo = gk[0], ghi = gk[7];
	double tlo = h, thi = -h;
	{
		bool found = false;
#pragma unroll
		for (int k = 1; k < 8; k++) {  // first k with slope >= 0 (slope < 0 before it)
			const bool hit = !found && gk[k] >= 0;
			const double den = gk[k] - gk[k - 1];
			const double tx = gk[k] > 0 ? tk[k - 1] + (tk[k] - tk[k - 1]) * (-gk[k - 1] / (den > 0 ? den : 1.0)) : tk[k];
			tlo = hit ? tx : tlo;
			found = found || hit;
		}
		if (glo >= 0) tlo = -h;
		else if (ghi < 0) tlo = h;
		found = false;
#pragma unroll
		for (int k = 6; k >= 0; k--) {  // last k with slope <= 0 (slope > 0 after it)
			const bool hit = !found && gk[k] <= 0;
			const double den = gk[k + 1] - gk[k];
			const double tx = gk[k] < 0 ? tk[k] + (tk[k + 1] - tk[k]) * (-gk[k] / (den > 0 ? den : 1.0)) : tk[k];
			thi = hit ? tx : thi;
			found = found || hit;
		}
		if (ghi <= 0) thi = h;
		else if (glo > 0) thi = -h;
	}
	if (thi < tlo) thi = tlo;
	const double ts = 0.5 * (tlo + thi);
	int nout = 0, face = 0;
	for (int i = 0; i < 3; i++)
		if (fabs(p0[i] + ts * d[i]) > size2[i] + MJB_TIE) {
			nout++;
			face = i;
		}
	double ta = tlo, tb = thi;
	if (nout <= 1) {
		ta = -h;
		tb = h;
		for (int j = 0; j < 3; j++) {
			if (nout == 1 && j == face) continue;
			if (fabs(d[j]) <= MJB_MINVAL) continue;
			double t1 = (-size2[j] - p0[j]) / d[j], t2 = (size2[j] - p0[j]) / d[j];
			if (t1 > t2) {
				const double sw = t1;
				t1 = t2;
				t2 = sw;
			}
			if (t1 > ta) ta = t1;
			if (t2 < tb) tb = t2;
		}
		if (ta > tb) ta = tb = ts;
	}
	int n = 0;
	double ctr[3] = { pos1[0] + axis[0] * ta, pos1[1] + axis[1] * ta, pos1[2] + axis[2] * ta };
	n += raw_sphere_box(rc[0], ctr, r, pos2, mat2, size2, margin);
	if (tb - ta > 1e-6 * h) {
		ctr[0] = pos1[0] + axis[0] * tb; ctr[1] = pos1[1] + axis[1] * tb; ctr[2] = pos1[2] + axis[2] * tb;
		RawCon c2;
		if (raw_sphere_box(c2, ctr, r, pos2, mat2, size2, margin)) {
			rc_append(rc, n, c2);
			n++;
		}
	}
	return n;
}

// box - box: same steps as oracle/mjo_constraint.c box_box (separating axes, then either the incident face clipped
// against the reference face -- every clipped vertex within the margin is a contact, up to 8 -- or one edge-edge contact)
// The clipping polygons, the axis tables and the half sizes are indexed dynamically (separating-axis code, reference face): they
// (and the up to 8 output contacts, 10 doubles each: dist, pos, normal + tangent hint) live in a MJB_BBSCR-double LDS scratch `scr`
// that ONE lane at a time owns (collision() serialises the box - box lanes) -- private arrays indexed that way sit in scratch
// memory, where every access is a trip to L2 / HBM (config 5 runs this every step: the cube lies on the palm plate).
#define MJB_BBSCR 216
DEVI int box_box(double *scr, const double *pos1, const double *mat1, const double *size1_in, const double *pos2,
                 const double *mat2, const double *size2_in, double margin)
{
	double (*poly)[3] = reinterpret_cast<double (*)[3]>(scr), (*tmp)[3] = reinterpret_cast<double (*)[3]>(scr + 24);
	double *out = scr + 48;
	double (*A)[3] = reinterpret_cast<double (*)[3]>(scr + 128), (*B)[3] = reinterpret_cast<double (*)[3]>(scr + 137);
	double (*C)[3] = reinterpret_cast<double (*)[3]>(scr + 146), (*Q)[3] = reinterpret_cast<double (*)[3]>(scr + 155);
	double *tA = scr + 164, *tB = scr + 167, *size1 = scr + 203, *size2 = scr + 206;
	for (int i = 0; i < 3; i++) {
		size1[i] = size1_in[i];
		size2[i] = size2_in[i];
	}
	for (int i = 0; i < 3; i++)
		for (int k = 0; k < 3; k++) {
			A[i][k] = mat1[3 * k + i];
			B[i][k] = mat2[3 * k + i];
		}
	const double t[3] = { pos2[0] - pos1[0], pos2[1] - pos1[1], pos2[2] - pos1[2] };
	for (int i = 0; i < 3; i++) {
		tA[i] = dot3(t, A[i]);
		tB[i] = dot3(t, B[i]);
		for (int j = 0; j < 3; j++) {
			C[i][j] = dot3(A[i], B[j]);
			Q[i][j] = fabs(C[i][j]) + 1e-12;
		}
	}
	double best = -1e300;
	int code = -1;
	for (int i = 0; i < 3; i++) {
		const double s = fabs(tA[i]) - (size1[i] + size2[0] * Q[i][0] + size2[1] * Q[i][1] + size2[2] * Q[i][2]);
		if (s > margin) return 0;
		if (s > best + MJB_TIE) { best = s; code = i; }
	}
	for (int j = 0; j < 3; j++) {
		const double s = fabs(tB[j]) - (size2[j] + size1[0] * Q[0][j] + size1[1] * Q[1][j] + size1[2] * Q[2][j]);
		if (s > margin) return 0;
		if (s > best + MJB_TIE) { best = s; code = 3 + j; }
	}
	double ebest = -1e300;
	int ecode = -1;
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++) {
			const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
			const double l = sqrt(fmax(0.0, 1.0 - C[i][j] * C[i][j]));
			if (l < 1e-6) continue;
			double s = fabs(tA[i2] * C[i1][j] - tA[i1] * C[i2][j]) -
			           (size1[i1] * Q[i2][j] + size1[i2] * Q[i1][j] + size2[j1] * Q[i][j2] + size2[j2] * Q[i][j1]);
			s /= l;
			if (s > margin) return 0;
			if (s > ebest + MJB_TIE) { ebest = s; ecode = 6 + 3 * i + j; }
		}
	if (ecode >= 0 && ebest > best + 0.05 * fabs(best) + 1e-9) {
		const int i = (ecode - 6) / 3, j = (ecode - 6) % 3;
		double n[3];
		cross3(n, A[i], B[j]);
		normalize3(n);
		if (dot3(n, t) < 0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
		double pa[3] = { pos1[0], pos1[1], pos1[2] }, pb[3] = { pos2[0], pos2[1], pos2[2] };
		for (int k = 0; k < 3; k++) {
			if (k != i) {
				const double sg = dot3(n, A[k]) > 0 ? 1.0 : -1.0;
				for (int q = 0; q < 3; q++) pa[q] += sg * size1[k] * A[k][q];
			}
			if (k != j) {
				const double sg = dot3(n, B[k]) > 0 ? -1.0 : 1.0;
				for (int q = 0; q < 3; q++) pb[q] += sg * size2[k] * B[k][q];
			}
		}
		const double d[3] = { pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2] };
		const double uaub = C[i][j], q1 = dot3(A[i], d), q2 = -dot3(B[j], d), den = 1 - uaub * uaub;
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
		const double dv[3] = { xb[0] - xa[0], xb[1] - xa[1], xb[2] - xa[2] };
		const double dist = dot3(dv, n);
		if (dist > margin) return 0;
		out[0] = dist;
		for (int q = 0; q < 3; q++) {
			out[1 + q] = 0.5 * (xa[q] + xb[q]);
			out[4 + q] = n[q];
			out[7 + q] = 0;
		}
		return 1;
	}
	const bool ref1 = code < 3;
	const int ax = ref1 ? code : code - 3;
	double (*R)[3] = reinterpret_cast<double (*)[3]>(scr + 170), (*O)[3] = reinterpret_cast<double (*)[3]>(scr + 179);
	double *pr = scr + 188, *po = scr + 191, *hr = scr + 194, *ho = scr + 197;
	for (int i = 0; i < 3; i++) {
		pr[i] = ref1 ? pos1[i] : pos2[i];
		po[i] = ref1 ? pos2[i] : pos1[i];
		hr[i] = ref1 ? size1[i] : size2[i];
		ho[i] = ref1 ? size2[i] : size1[i];
		for (int k = 0; k < 3; k++) {
			R[i][k] = ref1 ? A[i][k] : B[i][k];
			O[i][k] = ref1 ? B[i][k] : A[i][k];
		}
	}
	double nref[3];
	{
		const double sg = (ref1 ? tA[ax] : -tB[ax]) >= -MJB_TIE ? 1.0 : -1.0;  // (centres level along the axis: the + face, whatever the last bit says)
		for (int q = 0; q < 3; q++) nref[q] = sg * R[ax][q];
	}
	int k = 0;
	double kbest = -1;
	for (int q = 0; q < 3; q++) {
		const double a = fabs(dot3(O[q], nref));
		if (a > kbest + MJB_TIE) { kbest = a; k = q; }
	}
	const double fs = dot3(O[k], nref) > 0 ? -1.0 : 1.0;
	const int u = (k + 1) % 3, v = (k + 2) % 3, sx = (ax + 1) % 3, sy = (ax + 2) % 3;
	int np = 4;
	for (int w = 0; w < 4; w++) {
		const double su = (w == 0 || w == 3) ? 1.0 : -1.0, sv = (w < 2) ? 1.0 : -1.0;
		double d[3];
		for (int q = 0; q < 3; q++) d[q] = po[q] + fs * ho[k] * O[k][q] + su * ho[u] * O[u][q] + sv * ho[v] * O[v][q] - pr[q];
		poly[w][0] = dot3(d, R[sx]);
		poly[w][1] = dot3(d, R[sy]);
		poly[w][2] = dot3(d, nref) - hr[ax];
	}
	for (int side = 0; side < 4; side++) {
		const int cax = side >> 1;
		const double sgn = (side & 1) ? -1.0 : 1.0, lim = cax == 0 ? hr[sx] : hr[sy];
		int nn = 0;
		for (int w = 0; w < np; w++) {
			const int w1 = (w + 1 == np) ? 0 : w + 1;
			const double d0 = sgn * poly[w][cax] - lim, d1 = sgn * poly[w1][cax] - lim;
			if (d0 <= MJB_TIE && nn < 8) {  // (within MJB_TIE of the side plane: ON it -- kept, and no crossing next to it)
				for (int q = 0; q < 3; q++) tmp[nn][q] = poly[w][q];
				nn++;
			}
			if (((d0 < -MJB_TIE && d1 > MJB_TIE) || (d0 > MJB_TIE && d1 < -MJB_TIE)) && nn < 8) {
				const double fr = d0 / (d0 - d1);
				for (int q = 0; q < 3; q++) tmp[nn][q] = poly[w][q] + fr * (poly[w1][q] - poly[w][q]);
				nn++;
			}
		}
		np = nn;
		for (int w = 0; w < 8; w++)
			for (int q = 0; q < 3; q++) poly[w][q] = tmp[w][q];
		if (np == 0) return 0;
	}
	int nk = 0;
	for (int w = 0; w < np; w++)
		if (poly[w][2] < margin) {
			for (int q = 0; q < 3; q++) tmp[nk][q] = poly[w][q];
			nk++;
		}
	if (nk == 0) return 0;
	// every clipped vertex within the margin is a contact: up to 8, as mjc_BoxBox returns
	const int npick = nk < 8 ? nk : 8;
	for (int w = 0; w < npick; w++) {
		const double pv[3] = { tmp[w][0], tmp[w][1], tmp[w][2] };
		double *o = out + 10 * w;
		o[0] = pv[2];
		const double hgt = hr[ax] + 0.5 * pv[2];
		for (int q = 0; q < 3; q++) {
			o[1 + q] = pr[q] + pv[0] * R[sx][q] + pv[1] * R[sy][q] + hgt * nref[q];
			o[4 + q] = ref1 ? nref[q] : -nref[q];
			o[7 + q] = 0;
		}
	}
	return npick;
}

// narrow phase of one candidate pair; returns the number of raw contacts (<= 4; box - box: <= 8)
// (plane - box: up to four contacts that share the plane's normal -- only the ids of the qualifying corners come back, packed three
//  bits each in `pbc`; collision() rebuilds distance and point of each where it stores them.  Every other pair yields <= 2 contacts
//  in rc[]: two contact records less to keep in registers across the narrow phase of the 256-register PGS kernel.)
DEVI void plane_box_contact(int i, const double *pos1, const double *nrm, const double *pos2, const double *mat2, const double *size2,
                            double &cdist, double *cpos)
{
	const double dif[3] = { pos2[0] - pos1[0], pos2[1] - pos1[1], pos2[2] - pos1[2] };
	const double dist = dot3(dif, nrm);
	const double vec[3] = { (i & 1) ? size2[0] : -size2[0], (i & 2) ? size2[1] : -size2[1], (i & 4) ? size2[2] : -size2[2] };
	double corner[3];
	matvec3(corner, mat2, vec);
	const double ldist = dot3(nrm, corner);
	cdist = dist + ldist;
	for (int k = 0; k < 3; k++) cpos[k] = corner[k] + pos2[k] - nrm[k] * cdist * 0.5;
}

DEVI int narrowphase(int t1, int t2, const double *pos1, const double *mat1, const double *size1, const double *pos2,
                     const double *mat2, const double *size2, double margin, RawCon *rc, int &pbc)
{
	int n = 0;
	if (t1 == MJB_GEOM_PLANE) {
		const double nrm[3] = { mat1[2], mat1[5], mat1[8] };
		if (t2 == MJB_GEOM_SPHERE) {
			n = raw_plane_sphere(rc[0], pos1, nrm, pos2, size2[0], margin);
		} else if (t2 == MJB_GEOM_CAPSULE) {
			const double axis[3] = { mat2[2], mat2[5], mat2[8] };
			double p[3] = { pos2[0] + axis[0] * size2[1], pos2[1] + axis[1] * size2[1], pos2[2] + axis[2] * size2[1] };
			n = raw_plane_sphere(rc[0], pos1, nrm, p, size2[0], margin);
			p[0] = pos2[0] - axis[0] * size2[1]; p[1] = pos2[1] - axis[1] * size2[1]; p[2] = pos2[2] - axis[2] * size2[1];
			RawCon c2;
			if (raw_plane_sphere(c2, pos1, nrm, p, size2[0], margin)) {
				rc_append(rc, n, c2);
				n++;
			}
			for (int i = 0; i < 2; i++)
				if (i < n) { rc[i].frame[3] = axis[0]; rc[i].frame[4] = axis[1]; rc[i].frame[5] = axis[2]; }
		} else if (t2 == MJB_GEOM_BOX) {
			const double dif[3] = { pos2[0] - pos1[0], pos2[1] - pos1[1], pos2[2] - pos1[2] };
			const double dist = dot3(dif, nrm);
			// corners in mjc_PlaneBox's order (bit k of i: sign of axis k); the first four that qualify become contacts
			unsigned int okmask = 0;
#pragma unroll
			for (int i = 0; i < 8; i++) {
				const double vec[3] = { (i & 1) ? size2[0] : -size2[0], (i & 2) ? size2[1] : -size2[1],
					                    (i & 4) ? size2[2] : -size2[2] };
				double corner[3];
				matvec3(corner, mat2, vec);
				const double ldist = dot3(nrm, corner);
				if (!(dist + ldist > margin || ldist > 0)) okmask |= 1u << i;
			}
#pragma unroll
			for (int slot = 0; slot < 4; slot++) {
				if (okmask) {
					const int i = __builtin_ctz(okmask);
					okmask &= okmask - 1;
					pbc |= i << (3 * slot);
					n = slot + 1;
				}
			}
		}
	} else if (t1 == MJB_GEOM_SPHERE) {
		if (t2 == MJB_GEOM_SPHERE) {
			n = raw_sphere_sphere(rc[0], pos1, size1[0], pos2, size2[0], margin);
		} else if (t2 == MJB_GEOM_CAPSULE) {
			const double axis[3] = { mat2[2], mat2[5], mat2[8] };
			const double vec[3] = { pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2] };
			const double x = clipd(dot3(axis, vec), -size2[1], size2[1]);
			const double p[3] = { pos2[0] + axis[0] * x, pos2[1] + axis[1] * x, pos2[2] + axis[2] * x };
			n = raw_sphere_sphere(rc[0], pos1, size1[0], p, size2[0], margin);
		} else if (t2 == MJB_GEOM_BOX) {
			n = raw_sphere_box(rc[0], pos1, size1[0], pos2, mat2, size2, margin);
		}
	} else if (t1 == MJB_GEOM_CAPSULE && t2 == MJB_GEOM_BOX) {
		n = capsule_box(rc, pos1, mat1, size1, pos2, mat2, size2, margin);
	} else if (t1 == MJB_GEOM_CAPSULE && t2 == MJB_GEOM_CAPSULE) {
		const double a1[3] = { mat1[2], mat1[5], mat1[8] }, a2[3] = { mat2[2], mat2[5], mat2[8] };
		const double dif[3] = { pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2] };
		const double ma = dot3(a1, a1), mb = -dot3(a1, a2), mc = dot3(a2, a2);
		const double u = -dot3(a1, dif), v = dot3(a2, dif);
		const double det = ma * mc - mb * mb;
		double p1[3], p2[3];
		if (fabs(det) >= MJB_MINVAL) {
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
			n = raw_sphere_sphere(rc[0], p1, size1[0], p2, size2[0], margin);
		} else {
			for (int s = 1; s >= -1; s -= 2) {
				if (n >= 2) break;
				for (int k = 0; k < 3; k++) p1[k] = pos1[k] + a1[k] * s * size1[1];
				const double d2[3] = { p1[0] - pos2[0], p1[1] - pos2[1], p1[2] - pos2[2] };
				const double x2 = clipd(dot3(a2, d2) / mc, -size2[1], size2[1]);
				for (int k = 0; k < 3; k++) p2[k] = pos2[k] + a2[k] * x2;
				RawCon c;
				if (raw_sphere_sphere(c, p1, size1[0], p2, size2[0], margin)) {
					rc_append(rc, n, c);
					n++;
				}
			}
			for (int s = 1; s >= -1; s -= 2) {
				if (n >= 2) break;
				for (int k = 0; k < 3; k++) p2[k] = pos2[k] + a2[k] * s * size2[1];
				const double d1[3] = { p2[0] - pos1[0], p2[1] - pos1[1], p2[2] - pos1[2] };
				const double x1 = clipd(dot3(a1, d1) / ma, -size1[1], size1[1]);
				if (fabs(fabs(x1) - size1[1]) < MJB_MINVAL) continue;
				for (int k = 0; k < 3; k++) p1[k] = pos1[k] + a1[k] * x1;
				RawCon c;
				if (raw_sphere_sphere(c, p1, size1[0], p2, size2[0], margin)) {
					rc_append(rc, n, c);
					n++;
				}
			}
		}
	}
	return n;
}

// ------------------------------------------------------------------------------------------------
// A4 + A5  collision: one candidate pair per lane, contacts compacted in pair order
// ------------------------------------------------------------------------------------------------
// number of set bits of a wave ballot in the lanes below this one
DEVI int lanes_below(unsigned long long mask)
{
	return (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mask, 0u));
}

// Everything the two geoms' constants decide (types, sizes, margin, bounding radii, mj_contactParam's mixing of condim /
// solref / solimp) comes from the host-built per-pair record: one level of memory latency per step instead of the
// pair -> geom -> attribute chain.  Contact slots are assigned without LDS: a pair yields <= 8 contacts, so its offset is
// sum_k popcount(ballot(n >= k) below this lane).
template <int G> STAGE void collision(CModel m, CLayout L, CState s, const EnvLite &e)
{
	static_assert(G == 64, "one env per wavefront: pair offsets come from wave ballots");
	double *f = e.f;
	int *fi = e.fi;
	const int lane = e.lane;
	if (m.nconmax <= 0 || (m.disableflags & (MJB_DSBL_CONSTRAINT | MJB_DSBL_CONTACT))) {
		if (lane == 0) fi[L.ncon] = 0;
		gsync<G>();
		return;
	}
#if defined(MJB_PROFILE_SUB) || defined(MJB_PROFILE_COL)
	EPROF_BEGIN();
#endif
	// More candidate pairs than lanes: the bounding-sphere / plane cull of ALL pairs first (a record's head, two positions), the
	// survivors -- a wave mask per round of 64 pairs -- packed into as few narrow-phase rounds as they need, in pair order (so the
	// contacts come out in the order the uncompacted rounds give).  The hand model's 115 pairs leave ~20: one round of the narrow
	// phase, slot counting and stores instead of two.  (Per-env geom types can reorder a pair and change what the cull looks at:
	// those batches keep the plain rounds.)
	const bool compact = m.ncollpair > G && m.ncollpair <= 4 * G && s.env_geom_type == nullptr;
	unsigned long long cm0 = 0, cm1 = 0, cm2 = 0, cm3 = 0;
	int nsurv = 0, cc1 = 0, cc2 = 0, cc3 = 0;  // survivors before round 1 / 2 / 3's
	if (compact) {
		MJB_KEEP_BRANCH();
		auto cull_round = [&](int q) -> unsigned long long {
			const int p = G * q + lane;
			const bool valid = p < m.ncollpair;
			const int pc = valid ? p : 0;
			const mjb_ciptr pi = m.pair_i + 8 * pc;
			const mjb_cdptr pd = m.pair_d + 24 * pc;
			const int g1 = pi[0], g2 = pi[1], t1 = pi[2];
			const double margin = pd[6] + MJB_TIE, rb1 = pd[8], rb2 = pd[9];  // (the culls and the pair functions test against margin + MJB_TIE)
			double pos1[3], pos2[3];
			ld3(pos1, f + L.geom_xpos + 3 * g1);
			ld3(pos2, f + L.geom_xpos + 3 * g2);
			const double nrm[3] = { f[L.geom_xmat + 9 * g1 + 2], f[L.geom_xmat + 9 * g1 + 5], f[L.geom_xmat + 9 * g1 + 8] };
			const double dv[3] = { pos2[0] - pos1[0], pos2[1] - pos1[1], pos2[2] - pos1[2] };
			bool cull = false;
			if (rb1 > 0 && rb2 > 0) {
				const double bound = margin + rb1 + rb2;
				cull = dot3(dv, dv) > bound * bound;
			} else if (t1 == MJB_GEOM_PLANE && rb2 > 0) {
				cull = dot3(dv, nrm) > margin + rb2;
			}
			return __ballot(valid && !cull);
		};
		cm0 = cull_round(0);
		cm1 = cull_round(1);
		if (m.ncollpair > 2 * G) {
			MJB_KEEP_BRANCH();
			cm2 = cull_round(2);
			if (m.ncollpair > 3 * G) {
				MJB_KEEP_BRANCH();
				cm3 = cull_round(3);
			}
		}
		cc1 = __popcll(cm0);
		cc2 = cc1 + __popcll(cm1);
		cc3 = cc2 + __popcll(cm2);
		nsurv = cc3 + __popcll(cm3);
	}
	const int nitems = compact ? nsurv : m.ncollpair;
	int base = 0;  // contacts of the earlier rounds (wave-uniform)
	for (int p0 = 0; p0 < nitems; p0 += G) {
		int p = p0 + lane;
		if (compact) {
			MJB_KEEP_BRANCH();
			// the lane's survivor: the (p - cc_q)-th set bit of the round q that holds it (binary search by population counts)
			const int q = (p >= cc1) + (p >= cc2) + (p >= cc3);
			unsigned long long mk = q == 0 ? cm0 : (q == 1 ? cm1 : (q == 2 ? cm2 : cm3));
			int k = p - (q == 0 ? 0 : (q == 1 ? cc1 : (q == 2 ? cc2 : cc3))), pos = 0;
#pragma unroll
			for (int w = 32; w >= 1; w >>= 1) {
				const int c = __popcll((mk >> pos) & ((1ull << w) - 1ull));
				if (k >= c) {
					k -= c;
					pos += w;
				}
			}
			p = p < nsurv ? G * q + pos : m.ncollpair;
		}
		RawCon rc[2];  // (<= 2 contacts in registers; plane - box: the corner ids in pbc; box - box: from the LDS scratch straight to the frame)
		int pbc = 0;
		bool planebox = false;
		int n = 0, g1 = 0, g2 = 0, condim = 1, frisel = 0;
		double margin = 0, incl = 0;
		const mjb_cdptr pd = m.pair_d + 24 * (p < m.ncollpair ? p : 0);
		bool boxbox = false;  // an un-culled box - box pair: narrow phase below, one lane at a time
		double pos1[3], pos2[3], mat1[9], mat2[9];
		double size1[3] = { pd[0], pd[1], pd[2] }, size2[3] = { pd[3], pd[4], pd[5] };
		if (p < m.ncollpair) {
			const mjb_ciptr pi = m.pair_i + 8 * p;
			g1 = pi[0];
			g2 = pi[1];
			int t1 = pi[2], t2 = pi[3];
			condim = pi[4];
			frisel = pi[5];
			margin = pd[6] + MJB_TIE;  // (MJB_TIE: a surface at distance == margin is inside it; incl below is the exact margin - gap)
			incl = pd[17];
			double rb1 = pd[8], rb2 = pd[9];
			// per-env geom sizes / types (setGeomProperties per env, mjb_set_env_geom_size / _type): bounding radii stay the
			// model's, as in the reference; a type change may reverse the pair's (type1 <= type2) order
			if (s.env_geom_size) {
				const double *es = s.env_geom_size + (size_t)e.env * 3 * m.ngeom;
				for (int k = 0; k < 3; k++) {
					size1[k] = es[3 * g1 + k];
					size2[k] = es[3 * g2 + k];
				}
			}
			if (s.env_geom_type) {
				const int *et = s.env_geom_type + (size_t)e.env * m.ngeom;
				t1 = et[g1];
				t2 = et[g2];
				if (t1 > t2) {
					const int tg = g1; g1 = g2; g2 = tg;
					const int tt = t1; t1 = t2; t2 = tt;
					const double tr = rb1; rb1 = rb2; rb2 = tr;
					for (int k = 0; k < 3; k++) { const double ts = size1[k]; size1[k] = size2[k]; size2[k] = ts; }
					if (frisel == 1 || frisel == 2) frisel = 3 - frisel;
				}
			}
			ld3(pos1, f + L.geom_xpos + 3 * g1);
			ld3(pos2, f + L.geom_xpos + 3 * g2);
			ld9(mat1, f + L.geom_xmat + 9 * g1);
			ld9(mat2, f + L.geom_xmat + 9 * g2);
			bool cull = false;
			const double dv[3] = { pos2[0] - pos1[0], pos2[1] - pos1[1], pos2[2] - pos1[2] };
			if (rb1 > 0 && rb2 > 0) {
				const double bound = margin + rb1 + rb2;
				cull = dot3(dv, dv) > bound * bound;
			} else if (t1 == MJB_GEOM_PLANE && rb2 > 0) {
				const double nrm[3] = { mat1[2], mat1[5], mat1[8] };
				cull = dot3(dv, nrm) > margin + rb2;
			}
			// (every narrow-phase routine returns only contacts with dist <= margin, and mj_collideGeoms adds what its
			//  collision function returns: no second distance filter)
			if (!cull) {
				// MujocoEnv::registerCollisionFunction's override of the pair type (mjb_register_collision): the pair record carries the one of the
				// MODEL's types; with per-env geom types the table is read by the geoms' CURRENT types, as mjCOLLISIONFUNC is (mujoco_env.cpp:163-176)
				const int cfun = s.env_geom_type ? s.colfunc[8 * (t1 & 7) + (t2 & 7)] : pi[6];
				if (cfun == MJB_COLFUNC_DEFAULT) {
					if (t1 == MJB_GEOM_BOX && t2 == MJB_GEOM_BOX) boxbox = true;
					else {
						n = narrowphase(t1, t2, pos1, mat1, size1, pos2, mat2, size2, margin, rc, pbc);
						planebox = t1 == MJB_GEOM_PLANE && t2 == MJB_GEOM_BOX;
					}
				} else if (cfun == MJB_COLFUNC_SPHERES) {
					if (t1 == MJB_GEOM_PLANE) {
						const double nrm[3] = { mat1[2], mat1[5], mat1[8] };
						n = raw_plane_sphere(rc[0], pos1, nrm, pos2, rb2, margin);
					} else {
						n = raw_sphere_sphere(rc[0], pos1, rb1, pos2, rb2, margin);
					}
				}  // MJB_COLFUNC_NONE: no contacts
			}
		}
		// this pair's friction (mj_contactParam) and the store of one contact -- shared by the register path and the box - box path
		auto pair_friction = [&](double (&fri)[3]) {
			if (frisel == 4) {  // <contact><pair friction=...>: the pair's own numbers, a constant of the model (tangent 1, spin, roll 1 here; tangent 2 / roll 2: put_contact)
				for (int k = 0; k < 3; k++) fri[k] = pd[18 + k];
			} else if (L.gfriction >= 0) {
				for (int k = 0; k < 3; k++) {
					const double a = f[L.gfriction + 3 * g1 + k], b = f[L.gfriction + 3 * g2 + k];
					fri[k] = frisel == 0 ? fmax(a, b) : (frisel == 1 ? a : b);
				}
			} else if (s.env_geom_friction) {  // lean frame, per-env override: straight from HBM
				const double *gf = s.env_geom_friction + (size_t)e.env * 3 * m.ngeom;
				for (int k = 0; k < 3; k++) {
					const double a = gf[3 * g1 + k], b = gf[3 * g2 + k];
					fri[k] = frisel == 0 ? fmax(a, b) : (frisel == 1 ? a : b);
				}
			} else {  // lean frame: the model's geoms, mixed on the host (pair record)
				for (int k = 0; k < 3; k++) fri[k] = pd[18 + k];
			}
		};
		auto put_contact = [&](int c, double dist, const double *cpos, const double *frame6, const double (&fri)[3]) {
			double fr[9];
			for (int k = 0; k < 6; k++) fr[k] = frame6[k];
			make_frame(fr);
			f[L.contact_dist + c] = dist;
			st3(f + L.contact_pos + 3 * c, cpos);
			st9(f + L.contact_frame + 9 * c, fr);
			f[L.contact_includemargin + c] = incl;
			double *f5 = f + L.contact_friction + 5 * c;
			f5[0] = fri[0]; f5[1] = frisel == 4 ? pd[22] : fri[0]; f5[2] = fri[1]; f5[3] = fri[2]; f5[4] = frisel == 4 ? pd[23] : fri[2];
			if (L.contact_solref >= 0) {
				f[L.contact_solref + 2 * c] = pd[10];
				f[L.contact_solref + 2 * c + 1] = pd[11];
				for (int k = 0; k < 5; k++) f[L.contact_solimp + 5 * c + k] = pd[12 + k];
			}
			fi[L.contact_geom + 2 * c] = g1;
			fi[L.contact_geom + 2 * c + 1] = g2;
			fi[L.contact_dim + c] = condim;
			// no rows yet (< 0).  The lean frame keeps no per-contact solref / solimp: -2 - pair tells make_constraint
			// which pair record to read them from
			fi[L.contact_efc_address + c] = L.contact_solref >= 0 ? -1 : -2 - p;
		};
		// contact slot of every lane's pair from the counts known so far: a pair yields <= 8 contacts, so its offset is
		// sum_k popcount(ballot(n >= k) below this lane)
		auto slot_of = [&](int &total) {
			int o = base;
			total = 0;
#pragma unroll
			for (int k = 1; k <= 8; k++) {
				const unsigned long long mk = __ballot(n >= k);
				o += lanes_below(mk);
				total += __popcll(mk);
			}
			return o;
		};
#ifdef MJB_PROFILE_COL
		EPROF(20);
#endif
		{
			// box - box (up to 8 contacts, dynamically indexed clipping polygons): one lane at a time through the LDS scratch, and
			// from there straight into the frame.  A lane's slot only depends on the pairs before it: the register pairs below it
			// are counted already, the box - box pairs below it have had their turn, the ones above it still count 0.
			unsigned long long bbmask = __ballot(boxbox);
			while (bbmask) {  // (wave-uniform)
				const int l = __builtin_ctzll(bbmask);
				bbmask &= bbmask - 1;
				double *scr = f + L.bbscr;  // (transient scratch: nothing else of the frame's shared region is alive during collision)
				if (lane == l) {
					// (the geoms' poses are read again, through laundered pointers: otherwise the 24 doubles loaded for the cull stay
					//  live across every pair's narrow phase just for this rare path)
					const double *gx = f + L.geom_xpos, *gm = f + L.geom_xmat;
					asm volatile("" : "+v"(gx), "+v"(gm));
					double p1[3], p2[3], m1[9], m2[9];
					ld3(p1, gx + 3 * g1);
					ld3(p2, gx + 3 * g2);
					ld9(m1, gm + 9 * g1);
					ld9(m2, gm + 9 * g2);
					n = box_box(scr, p1, m1, size1, p2, m2, size2, margin);
				}
				int unused_total;
				const int offl = slot_of(unused_total);
				if (lane == l && n > 0) {
					double fri[3];
					pair_friction(fri);
					for (int i = 0; i < n; i++) {
						const int c = offl + i;
						if (c >= m.nconmax) break;
						const double *o = scr + 48 + 10 * i;
						put_contact(c, o[0], o + 1, o + 4, fri);
					}
				}
			}
		}
#ifdef MJB_PROFILE_SUB
		EPROF(24);
#endif
#ifdef MJB_PROFILE_COL
		EPROF(21);
#endif
		int total;
		const int off = slot_of(total);
		if (n > 0 && !boxbox) {
			double fri[3];
			pair_friction(fri);
			if (planebox) {
				// (the plane's and the box's poses are read again, like the box - box path does: kept in registers they would stay live
				//  across every pair's narrow phase for this one)
				const double *gx = f + L.geom_xpos, *gm = f + L.geom_xmat;
				asm volatile("" : "+v"(gx), "+v"(gm));
				double p1[3], p2[3], m2[9];
				ld3(p1, gx + 3 * g1);
				ld3(p2, gx + 3 * g2);
				ld9(m2, gm + 9 * g2);
				const double fr6[6] = { gm[9 * g1 + 2], gm[9 * g1 + 5], gm[9 * g1 + 8], 0, 0, 0 };
#pragma unroll
				for (int i = 0; i < 4; i++) {
					const int c = off + i;
					if (i >= n || c >= m.nconmax) continue;
					double cd, cp[3];
					plane_box_contact((pbc >> (3 * i)) & 7, p1, fr6, p2, m2, size2, cd, cp);
					put_contact(c, cd, cp, fr6, fri);
				}
			} else {
#pragma unroll
			for (int i = 0; i < 2; i++) {  // (fully unrolled, no early exit: rc[] stays in statically indexed registers)
				const int c = off + i;
				if (i >= n || c >= m.nconmax) continue;
				put_contact(c, rc[i].dist, rc[i].pos, rc[i].frame, fri);
			}
			}
		}
		base += total;
#ifdef MJB_PROFILE_SUB
		EPROF(25);
#endif
#ifdef MJB_PROFILE_COL
		EPROF(22);
#endif
	}
	if (lane == 0) {
		fi[L.ncon] = base < m.nconmax ? base : m.nconmax;
		if (base > m.nconmax) atomicAdd(s.nwarn + MJB_WARN_CONTACTFULL, 1ull);  // mjWARN_CONTACTFULL: the contacts past nconmax were dropped
	}
	gsync<G>();
}

// getimpedance
DEVI void impedance(const double *solimp, double pos, double margin, double &imp, double &impP)
{
	if (solimp[0] == solimp[1] || solimp[2] <= MJB_MINVAL) {
		imp = 0.5 * (solimp[0] + solimp[1]);
		impP = 0;
		return;
	}
	double x = (pos - margin) / solimp[2], sgn = 1;
	if (x < 0) {
		x = -x;
		sgn = -1;
	}
	if (x >= 1 || x <= 0) {
		imp = x >= 1 ? solimp[1] : solimp[0];
		impP = 0;
		return;
	}
	double y, yP;
	if (solimp[4] == 1) {
		y = x;
		yP = 1;
	} else if (solimp[4] == 2) {  // MuJoCo's default power: the same expressions with pow(., 2) / pow(., 1) written out
		if (x <= solimp[3]) {
			const double a = 1 / solimp[3];
			y = a * (x * x);
			yP = 2 * a * x;
		} else {
			const double b = 1 / (1 - solimp[3]);
			y = 1 - b * ((1 - x) * (1 - x));
			yP = 2 * b * (1 - x);
		}
	} else if (x <= solimp[3]) {
		const double a = 1 / pow(solimp[3], solimp[4] - 1);
		y = a * pow(x, solimp[4]);
		yP = solimp[4] * a * pow(x, solimp[4] - 1);
	} else {
		const double b = 1 / pow(1 - solimp[3], solimp[4] - 1);
		y = 1 - b * pow(1 - x, solimp[4]);
		yP = solimp[4] * b * pow(1 - x, solimp[4] - 1);
	}
	imp = solimp[0] + y * (solimp[1] - solimp[0]);
	impP = yP * sgn * (solimp[1] - solimp[0]) / solimp[2];
}

// K, B, imp, imp' of one row (getsolparam + getimpedance + the reference-acceleration gains of mj_makeImpedance)
struct RowGain {
	double K, B, imp, impP;
};
DEVI RowGain row_gain(CModel m, const double *solref_in, const double *solimp_in, double imp_pos, double margin)
{
	// getsolparam: mixed-sign solref falls back to the default (0.02, 1); refsafe; solimp clamped to its legal ranges
	double sr0 = solref_in[0], sr1 = solref_in[1];
	if ((sr0 > 0) != (sr1 > 0)) {
		sr0 = 0.02;
		sr1 = 1.0;
	}
	if (!(m.disableflags & MJB_DSBL_REFSAFE) && sr0 > 0) sr0 = fmax(sr0, 2 * m.timestep[0]);
	const double solimp[5] = { fmin(MJB_MAXIMP, fmax(MJB_MINIMP, solimp_in[0])), fmin(MJB_MAXIMP, fmax(MJB_MINIMP, solimp_in[1])),
		                       fmax(0.0, solimp_in[2]), fmin(MJB_MAXIMP, fmax(MJB_MINIMP, solimp_in[3])), fmax(1.0, solimp_in[4]) };
	RowGain g;
	impedance(solimp, imp_pos, margin, g.imp, g.impP);
	const double dmax = solimp[1];
	if (sr0 > 0) {
		g.K = 1 / fmax(MJB_MINVAL, dmax * dmax * sr0 * sr0 * sr1 * sr1);
		g.B = 2 / fmax(MJB_MINVAL, dmax * sr0);
	} else {
		g.K = -sr0 / fmax(MJB_MINVAL, dmax * dmax);
		g.B = -sr1 / fmax(MJB_MINVAL, dmax);
	}
	return g;
}
DEVI double row_R(const RowGain &g, double diag_approx) { return fmax(MJB_MINVAL, (1 - g.imp) * diag_approx / g.imp); }
// (lean frame of the fused step -- no efc_KBIP / efc_pos / efc_margin: efc_aref takes K imp (pos - margin) and efc_b the damping
//  gain B until reference_constraint folds them into aref; D or R is absent when the model's solver does not read it)
DEVI void row_store(CLayout L, double *f, int i, double pos, double margin, const RowGain &g, double R)
{
	if (L.efc_R >= 0) f[L.efc_R + i] = R;
	if (L.efc_D >= 0) f[L.efc_D + i] = 1.0 / R;
	if (L.efc_KBIP >= 0) {
		f[L.efc_KBIP + 4 * i] = g.K;
		f[L.efc_KBIP + 4 * i + 1] = g.B;
		f[L.efc_KBIP + 4 * i + 2] = g.imp;
		f[L.efc_KBIP + 4 * i + 3] = g.impP;
		f[L.efc_pos + i] = pos;
		f[L.efc_margin + i] = margin;
	} else {
		f[L.efc_aref + i] = g.K * g.imp * (pos - margin);
		f[L.efc_b + i] = g.B;
	}
}
// imp_pos: the position the impedance is evaluated at (rows of a connect / weld share the norm of their residual)
DEVI void row_params_x(CModel m, CLayout L, double *f, int i, double pos, double margin, const double *solref_in,
                       const double *solimp_in, double diag_approx, double imp_pos)
{
	const RowGain g = row_gain(m, solref_in, solimp_in, imp_pos, margin);
	row_store(L, f, i, pos, margin, g, row_R(g, diag_approx));
}
DEVI void row_params(CModel m, CLayout L, double *f, int i, double pos, double margin, const double *solref_in,
                     const double *solimp, double diag_approx)
{
	row_params_x(m, L, f, i, pos, margin, solref_in, solimp, diag_approx, pos);
}

// does dof `i` move body `b`?  (bit i of the body's ancestor-dof mask, built on the host; nv <= 64)
DEVI bool dof_moves_body(CModel m, int b, int i)
{
	const unsigned int w = (unsigned int)m.body_dofmask[2 * b + (i >> 5)];
	return (w >> (i & 31)) & 1u;
}

// connect / weld geometry shared by the row and the Jacobian passes (oracle/mjo_constraint.c make_equality):
// world anchor points of both bodies, and for a weld quat = q1 * relpose, quat1 = neg(q2)
struct EqGeom {
	double pos0[3], pos1[3], quat[4], quat1[4];
};
DEVI void eq_geometry(CModel m, CLayout L, const double *f, int e, EqGeom &g)
{
	const int type = m.eq_type[e], id0 = m.eq_obj1id[e], id1 = m.eq_obj2id[e];
	double a0[3], a1[3], M0[9], M1[9];
	for (int k = 0; k < 3; k++) {
		a0[k] = f[L.eqparam + 19 * e + 1 + (type == MJB_EQ_CONNECT ? 0 : 3) + k];
		a1[k] = f[L.eqparam + 19 * e + 1 + (type == MJB_EQ_CONNECT ? 3 : 0) + k];
	}
	ld9(M0, f + L.xmat + 9 * id0);
	ld9(M1, f + L.xmat + 9 * id1);
	matvec3(g.pos0, M0, a0);
	matvec3(g.pos1, M1, a1);
	for (int k = 0; k < 3; k++) {
		g.pos0[k] += f[L.xpos + 3 * id0 + k];
		g.pos1[k] += f[L.xpos + 3 * id1 + k];
	}
	if (type == MJB_EQ_WELD) {
		double q0[4], rel[4];
		for (int k = 0; k < 4; k++) {
			q0[k] = f[L.xquat + 4 * id0 + k];
			rel[k] = f[L.eqparam + 19 * e + 1 + 6 + k];
		}
		qmul(g.quat, q0, rel);
		g.quat1[0] = f[L.xquat + 4 * id1];
		for (int k = 1; k < 4; k++) g.quat1[k] = -f[L.xquat + 4 * id1 + k];
	}
}

// ------------------------------------------------------------------------------------------------
// A6  make_constraint: rows for equalities (connect / weld / joint), joint limits, then contacts
// ------------------------------------------------------------------------------------------------
// inclusive prefix sum over the 64 lanes of a wavefront (Hillis-Steele with ds_bpermute)
DEVI int wave_incl_scan(int v, int lane)
{
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const int y = __shfl_up(v, d, 64);
		if (lane >= d) v += y;
	}
	return v;
}

// Wave-uniform values read back from LDS (row / contact counts): telling the compiler so makes the loops on them scalar (fewer
// spills, and none of ROCm 7.2's spill-before-exec-restore patterns in the plain PGS kernel).  (While loop invariants were still
// hoisted to the top of the kernel this cost the 256-register PGS variant 400 spills and was switched off for it; with machine
// LICM off it is neutral there.)
template <int TAG> DEVI int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// A limited BALL joint (mj_instantiateLimit): the rotation angle of the joint quaternion -- mju_quat2Vel(., 1), the angle in (-pi, pi] times the axis, then
// mju_normalize3 -- against max(range); one row, Jacobian = minus the unit axis on the joint's three dofs.  Out of line: the hinge / slide path of
// make_constraint keeps its registers, and the trigonometry is only in the instruction stream of the wavefronts that meet such a joint.
__device__ __attribute__((noinline)) double ball_limit_angle(const double *q, double *aa)
{
	double ax[3] = { q[1], q[2], q[3] };
	const double sn = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
	if (sn < MJB_MINVAL) {
		ax[0] = 1; ax[1] = 0; ax[2] = 0;
	} else {
		const double r = 1 / sn;
		ax[0] *= r; ax[1] *= r; ax[2] *= r;
	}
	double speed = 2 * atan2(sn, q[0]);
	if (speed > 3.14159265358979323846) speed -= 2 * 3.14159265358979323846;
	for (int k = 0; k < 3; k++) aa[k] = ax[k] * speed;
	const double n = sqrt(aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2]);
	if (n < MJB_MINVAL) {
		aa[0] = 1; aa[1] = 0; aa[2] = 0;
	} else {
		const double r = 1 / n;
		aa[0] *= r; aa[1] *= r; aa[2] *= r;
	}
	return n;
}

// TAG 4 (Newton, up to 256 rows, fused step): the frame holds the first L.jrows rows of efc_J only -- what lets two envs of
// config 5 share a CU's LDS.  Rows beyond go to the env's block of s.efc_Jg in HBM; an env-step that ends up with more than
// L.jrows rows (none of config 5's do) copies the leading rows there as well and the solver reads ALL of J from HBM.
template <int G, int TAG> STAGE void make_constraint(CModel m, CLayout L, CState s, const EnvLite &e)
{
	static_assert(G == 64, "one env per wavefront: row offsets come from a wave prefix sum");
	double *f = e.f;
	int *fi = e.fi;
	const int lane = e.lane, nv = m.nv;
	[[maybe_unused]] RowBlock gb{};
	if constexpr (TAG == 4)
		if (s.efc_Jg) gb = mjb_rowblock(s.efc_Jg + (size_t)e.env * s.efc_Jg_stride, m.nefcmax, nv, m.nconmax, L.hcs);
	[[maybe_unused]] double *Jg = gb.J;
	auto jrow = [&](int r) -> double * {  // row r of efc_J
		if constexpr (TAG == 4) return r < L.jrows ? f + L.efc_J + r * nv : Jg + (size_t)r * nv;
		else return f + L.efc_J + r * nv;
	};
	// the other per-row arrays: rows >= L.rcap (X-layout fused frame of variant 4 only) go to the env's block in HBM
	auto rstore = [&](int i, double pos, double margin, const RowGain &g, double R) {
		if constexpr (TAG == 4) {
			if (i >= L.rcap) {  // (a lean primal frame: D, and aref / b as row_store parks them)
				gb.D[i] = 1.0 / R;
				gb.aref[i] = g.K * g.imp * (pos - margin);
				gb.b[i] = g.B;
				return;
			}
		}
		row_store(L, f, i, pos, margin, g, R);
	};
	auto rtag = [&](int i, int type, int id) {
		if constexpr (TAG == 4) {
			if (i >= L.rcap) {
				gb.type[i] = type;
				gb.id[i] = id;
				return;
			}
		}
		fi[L.efc_id + i] = id;
		fi[L.efc_type + i] = type;
	};
	auto rfl = [&](int i, double v) {
		if constexpr (TAG == 4) {
			if (i >= L.rcap) {
				gb.fl[i] = v;
				return;
			}
		}
		f[L.efc_frictionloss + i] = v;
	};
	if (lane == 0) fi[L.nefc] = 0;
	if (m.nefcmax <= 0 || (m.disableflags & MJB_DSBL_CONSTRAINT)) {
		gsync<G>();
		return;
	}
	const int ncon = wave_uniform<TAG>(fi[L.ncon]);
	const bool do_lim = !(m.disableflags & MJB_DSBL_LIMIT), do_con = !(m.disableflags & MJB_DSBL_CONTACT);
	const int neq = (m.disableflags & MJB_DSBL_EQUALITY) ? 0 : m.neq;
	const int nten = do_lim ? m.ntendon : 0;
	const int nfd = m.nfriction > 0 ? nv : 0;  // dof friction items, then tendon friction items (models without dry friction keep the shorter list)
	const int nfr = m.nfriction > 0 ? nv + m.ntendon : 0;
	const int nitem = neq + nfr + m.njnt + nten + ncon;  // item = equality, dof / tendon friction, joint limit, tendon limit or contact -- MuJoCo's row order
	int *adr = fi + L.iscratch;              // first row of every item (-1: no rows / dropped); read by the Jacobian passes
#ifdef MJB_PROFILE_SUB
	EPROF_BEGIN();
#endif
#ifdef MJB_PROFILE_MK
	EPROF_BEGIN();
#endif
	if (nfr)
		for (int r = lane; r < m.nefcmax; r += G) rfl(r, 0.0);
	// Items in MuJoCo's row order, one per lane, 64 at a time: rows per item -> wave prefix sum -> row parameters.
	// Row budget: the first item that does not fit, and everything after it, is dropped (mjWARN_CNSTRFULL).
	int nefc = 0;       // rows so far (wave-uniform)
	bool full = false;  // an item did not fit: the rest is dropped
	for (int it0 = 0; it0 < nitem; it0 += G) {
		const int it = it0 + lane;
		int n = 0;
		// The lane's item record -- limit (host table lim_d / lim_i) or contact (the pair record its contact came from) -- in ONE batch
		// of unconditional loads before the kinds diverge: round 3 walked the model tables inside the divergent branches of both
		// passes (jnt_limited -> jnt_range / margin / solref / solimp -> dof_invweight0;  contact -> geom -> body -> invweight),
		// a chain of global-memory round trips per kind, one kind after the other: 10 k cycles for config 3's 19 rows.
		const int it_lim0 = neq + nfr, it_con0 = neq + nfr + m.njnt + nten;
		const bool is_lim = it >= it_lim0 && it < it_con0, is_con = it >= it_con0 && it < nitem;
		const int li = is_lim ? it - it_lim0 : 0;
		const int lean_pair = (is_con && L.contact_solref < 0) ? -2 - fi[L.contact_efc_address + (it - it_con0)] : 0;
		const mjb_cdptr rec = is_con ? m.pair_d + 24 * (lean_pair >= 0 && lean_pair < m.ncollpair ? lean_pair : 0) : m.lim_d + 24 * li;
		const double rc0 = rec[0], rc1 = rec[1], rc6 = rec[6], rc21 = rec[21];
		double rcs[7];
#pragma unroll
		for (int k = 0; k < 7; k++) rcs[k] = rec[10 + k];  // solref[2] | solimp[5]
		const int lim_on = m.lim_i[4 * li], lim_q = m.lim_i[4 * li + 1], lim_d = m.lim_i[4 * li + 2];
		if (it < nitem && !full) {
			if (it < neq) {
				if (f[L.eqparam + 19 * it] != 0) n = m.eq_type[it] == MJB_EQ_CONNECT ? 3 : (m.eq_type[it] == MJB_EQ_WELD ? 6 : 1);  // joint, tendon: 1
			} else if (it < neq + nfr) {
				n = (it < neq + nfd ? m.dof_frictionloss[it - neq] : m.tendon_frictionloss[it - neq - nfd]) > 0 ? 1 : 0;
			} else if (it < neq + nfr + m.njnt + nten) {
				if (do_lim && lim_on == 2) {  // (a limited ball joint: one row when the rotation angle comes within the margin of max(range))
					MJB_KEEP_BRANCH();
					double aa[3];
					const double value = ball_limit_angle(f + L.qpos + lim_q, aa);
					if (fmax(rc0, rc1) - value < rc6) n = 1;
				} else if (do_lim && lim_on) {  // (a limited slide / hinge joint, or a limited tendon)
					const double value = it < neq + nfr + m.njnt ? f[L.qpos + lim_q] : f[L.ten_length + lim_q], margin = rc6;
					if (value - rc0 < margin) n++;
					if (rc1 - value < margin) n++;
				}
			} else if (do_con) {
				const int c = it - neq - nfr - m.njnt - nten;
				if (f[L.contact_dist + c] < f[L.contact_includemargin + c]) {
					const int dim = fi[L.contact_dim + c];
					n = dim == 1 ? 1 : (m.cone == MJB_CONE_ELLIPTIC ? dim : 2 * (dim - 1));
				}
			}
		}
		const int incl = wave_incl_scan(n, lane);
		int off = nefc + incl - n;
		const unsigned long long over = __ballot(n > 0 && off + n > m.nefcmax);
		int round_rows = __builtin_amdgcn_readlane(incl, 63);
		if (over) {  // (wave-uniform) first lane that does not fit: it and everything after it is dropped
			const int cutlane = __builtin_ctzll(over);
			round_rows = __builtin_amdgcn_readlane(incl - n, cutlane);
			if (lane >= cutlane) n = 0;
			if (!full && lane == 0) atomicAdd(s.nwarn + MJB_WARN_CNSTRFULL, 1ull);  // mjWARN_CNSTRFULL (rule: include/mjb.h, mjb_warning)
			full = true;
		}
		nefc += round_rows;
		if (it < nitem) adr[it] = n > 0 ? off : -1;
#ifdef MJB_PROFILE_SUB
		EPROF(26);
#endif
#ifdef MJB_PROFILE_MK
		EPROF(20);
#endif
		if (n == 0) continue;
		// The lanes of a round hold items of different kinds, and a wave runs divergent branches one after the other: the branches
		// only gather what differs per kind (J row, solver parameters, the position the impedance is evaluated at); the gain
		// evaluation itself -- getsolparam + getimpedance, a dozen fp64 divisions -- runs ONCE for all kinds, then the stores.
		double solref[2] = { 0, 0 }, solimp[5] = { 0, 0, 0, 0, 0 }, ipos = 0, imarg = 0;
		double cpos[6] = { 0, 0, 0, 0, 0, 0 }, diag[6] = { 0, 0, 0, 0, 0, 0 };  // equality rows: residuals, diagApprox
		double dist2 = 0;   // second side of a limit (both sides active: degenerate range)
		int idv = 0;        // efc_id of the item's rows
		if (it < neq) {
			const int eq = it, type = m.eq_type[eq], id0 = m.eq_obj1id[eq], id1 = m.eq_obj2id[eq];
			idv = eq;
			if (type == MJB_EQ_JOINT) {
				const int a1 = m.jnt_qposadr[id0], d1 = m.jnt_dofadr[id0];
				double c5[5];
				for (int k = 0; k < 5; k++) c5[k] = f[L.eqparam + 19 * eq + 1 + k];
				double poly = c5[0], deriv = 0;
				double *row = jrow(off);
				for (int k = 0; k < nv; k++) row[k] = 0;
				diag[0] = MP_DOF_INVW(m, e, d1);
				if (id1 >= 0) {
					const int a2 = m.jnt_qposadr[id1], d2 = m.jnt_dofadr[id1];
					const double x = f[L.qpos + a2] - m.qpos0[a2];
					poly = c5[0] + x * (c5[1] + x * (c5[2] + x * (c5[3] + x * c5[4])));
					deriv = c5[1] + x * (2 * c5[2] + x * (3 * c5[3] + x * 4 * c5[4]));
					row[d2] = -deriv;
					diag[0] += MP_DOF_INVW(m, e, d2);
				}
				row[d1] += 1;
				cpos[0] = f[L.qpos + a1] - m.qpos0[a1] - poly;
			} else if (type == MJB_EQ_TENDON) {
				double c5[5];
				for (int k = 0; k < 5; k++) c5[k] = f[L.eqparam + 19 * eq + 1 + k];
				double poly = c5[0], deriv = 0;
				double *row = jrow(off);
				for (int k = 0; k < nv; k++) row[k] = 0;
				diag[0] = MP_TEN_INVW(m, e, id0);
				if (id1 >= 0) {
					const double x = f[L.ten_length + id1] - m.tendon_length0[id1];
					poly = c5[0] + x * (c5[1] + x * (c5[2] + x * (c5[3] + x * c5[4])));
					deriv = c5[1] + x * (2 * c5[2] + x * (3 * c5[3] + x * 4 * c5[4]));
					for (int w = m.tendon_adr[id1]; w < m.tendon_adr[id1] + m.tendon_num[id1]; w++)
						row[m.jnt_dofadr[m.wrap_objid[w]]] -= deriv * m.wrap_prm[w];
					diag[0] += MP_TEN_INVW(m, e, id1);
				}
				for (int w = m.tendon_adr[id0]; w < m.tendon_adr[id0] + m.tendon_num[id0]; w++)
					row[m.jnt_dofadr[m.wrap_objid[w]]] += m.wrap_prm[w];
				cpos[0] = f[L.ten_length + id0] - m.tendon_length0[id0] - poly;
			} else {
				EqGeom g;
				eq_geometry(m, L, f, eq, g);
				for (int k = 0; k < 3; k++) cpos[k] = g.pos0[k] - g.pos1[k];
				const double tran = MP_BODY_INVW(m, e, 2 * id0) + MP_BODY_INVW(m, e, 2 * id1);
				diag[0] = diag[1] = diag[2] = tran;
				if (type == MJB_EQ_WELD) {
					const double ts = f[L.eqparam + 19 * eq + 1 + 10];
					double q2[4];
					qmul(q2, g.quat1, g.quat);
					for (int k = 0; k < 3; k++) cpos[3 + k] = ts * q2[1 + k];
					diag[3] = diag[4] = diag[5] = MP_BODY_INVW(m, e, 2 * id0 + 1) + MP_BODY_INVW(m, e, 2 * id1 + 1);
				}
			}
			double nrm = 0;
			for (int k = 0; k < 6; k++) nrm += cpos[k] * cpos[k];
			nrm = sqrt(nrm);
			solref[0] = f[L.eqparam + 19 * eq + 12];
			solref[1] = f[L.eqparam + 19 * eq + 13];
			for (int k = 0; k < 5; k++) solimp[k] = f[L.eqparam + 19 * eq + 14 + k];
			ipos = n > 1 ? nrm : cpos[0];  // (the rows of a connect / weld share the norm of their residual)
		} else if (it < neq + nfr) {
			// dry friction (mj_instantiateFriction): J = e_dof or the tendon's moment arms, pos = margin = 0, |force| <= frictionloss
			const bool isdof = it < neq + nfd;
			const int i = isdof ? it - neq : it - neq - nfd;
			idv = i;
			double *row = jrow(off);
			for (int k = 0; k < nv; k++) row[k] = 0;
			if (isdof) {
				row[i] = 1;
			} else {
				for (int w = m.tendon_adr[i]; w < m.tendon_adr[i] + m.tendon_num[i]; w++)
					row[m.jnt_dofadr[m.wrap_objid[w]]] += m.wrap_prm[w];
			}
			const mjb_cdptr sr = isdof ? m.dof_solref + 2 * i : m.tendon_solref_fri + 2 * i;
			const mjb_cdptr si = isdof ? m.dof_solimp + 5 * i : m.tendon_solimp_fri + 5 * i;
			solref[0] = sr[0];
			solref[1] = sr[1];
			for (int k = 0; k < 5; k++) solimp[k] = si[k];
			diag[0] = isdof ? MP_DOF_INVW(m, e, i) : MP_TEN_INVW(m, e, i);
		} else if (it < neq + nfr + m.njnt + nten) {
			// joint / tendon limit: one row per active side, lower side first
			const bool isj = it < neq + nfr + m.njnt;
			const int j = isj ? it - neq - nfr : it - neq - nfr - m.njnt;
			idv = j;
			if (lim_on == 2) {  // ball joint: row = -axis on its three dofs, pos = max(range) - angle
				MJB_KEEP_BRANCH();
				double aa[3];
				const double ang = ball_limit_angle(f + L.qpos + lim_q, aa);
				double *row = jrow(off);
				for (int k = 0; k < nv; k++) row[k] = 0;
				for (int k = 0; k < 3; k++) row[lim_d + k] = -aa[k];
				imarg = rc6;
				solref[0] = rcs[0];
				solref[1] = rcs[1];
				for (int k = 0; k < 5; k++) solimp[k] = rcs[2 + k];
				diag[0] = e.mp ? MP_DOF_INVW(m, e, lim_d) : rc21;
				ipos = fmax(rc0, rc1) - ang;
			} else {
			const double value = isj ? f[L.qpos + lim_q] : f[L.ten_length + j];
			imarg = rc6;
			const double rng[2] = { rc0, rc1 };
			solref[0] = rcs[0];
			solref[1] = rcs[1];
			for (int k = 0; k < 5; k++) solimp[k] = rcs[2 + k];
			diag[0] = e.mp ? (isj ? MP_DOF_INVW(m, e, lim_d) : MP_TEN_INVW(m, e, j)) : rc21;
			const double dlo = -1 * (rng[0] - value), dhi = 1 * (rng[1] - value);
			int r = off;
			for (int side = -1; side <= 1; side += 2) {
				if ((side < 0 ? dlo : dhi) < imarg) {
					double *row = jrow(r);
					for (int k = 0; k < nv; k++) row[k] = 0;
					if (isj) row[lim_d] = -side;
					else
						for (int w = m.tendon_adr[j]; w < m.tendon_adr[j] + m.tendon_num[j]; w++)
							row[m.jnt_dofadr[m.wrap_objid[w]]] += -side * m.wrap_prm[w];
					r++;
				}
			}
			ipos = dlo < imarg ? dlo : dhi;
			dist2 = dhi;
			}
		} else {
			const int c = it - neq - nfr - m.njnt - nten;
			idv = c;
			if (L.contact_solref >= 0) {
				solref[0] = f[L.contact_solref + 2 * c];
				solref[1] = f[L.contact_solref + 2 * c + 1];
				for (int k = 0; k < 5; k++) solimp[k] = f[L.contact_solimp + 5 * c + k];
			} else {  // lean frame: the mixed parameters of the pair record this contact came from (collision left -2 - pair), fetched above
				solref[0] = rcs[0];
				solref[1] = rcs[1];
				for (int k = 0; k < 5; k++) solimp[k] = rcs[2 + k];
			}
			fi[L.contact_efc_address + c] = off;
			// ONE impedance evaluation per contact: its rows share solref / solimp, and either all of them sit at
			// (pos, margin) = (dist, includemargin) [frictionless, pyramidal] or the friction rows sit at (0, 0) [elliptic]
			ipos = f[L.contact_dist + c];
			imarg = f[L.contact_includemargin + c];
		}
		const RowGain g0 = row_gain(m, solref, solimp, ipos, imarg);
		if (it < neq) {
			for (int k = 0; k < 6; k++) {
				if (k >= n) break;
				rstore(off + k, cpos[k], 0.0, g0, row_R(g0, diag[k]));
				rtag(off + k, MJB_CNSTR_EQUALITY, idv);
			}
		} else if (it < neq + nfr) {
			const bool isdof = it < neq + nfd;
			rstore(off, 0.0, 0.0, g0, row_R(g0, diag[0]));
			rfl(off, isdof ? m.dof_frictionloss[idv] : m.tendon_frictionloss[idv]);
			rtag(off, isdof ? MJB_CNSTR_FRICTION_DOF : MJB_CNSTR_FRICTION_TENDON, idv);
		} else if (it < neq + nfr + m.njnt + nten) {
			const int type = it < neq + nfr + m.njnt ? MJB_CNSTR_LIMIT_JOINT : MJB_CNSTR_LIMIT_TENDON;
			rstore(off, ipos, imarg, g0, row_R(g0, diag[0]));
			rtag(off, type, idv);
			if (n == 2) {  // both sides within the margin
				const RowGain g1 = row_gain(m, solref, solimp, dist2, imarg);
				rstore(off + 1, dist2, imarg, g1, row_R(g1, diag[0]));
				rtag(off + 1, type, idv);
			}
		} else {
			const int c = idv;
			const int dim = fi[L.contact_dim + c];
			const double dist = ipos, cm = imarg;
			double tran = rc21;  // (lean frame, the model's masses: the pair record carries the sum)
			if (e.mp || L.contact_solref >= 0) {
				const int b1 = m.geom_bodyid[fi[L.contact_geom + 2 * c]], b2 = m.geom_bodyid[fi[L.contact_geom + 2 * c + 1]];
				tran = MP_BODY_INVW(m, e, 2 * b1) + MP_BODY_INVW(m, e, 2 * b2);
			}
			double fri[5];
			for (int k = 0; k < 5; k++) fri[k] = f[L.contact_friction + 5 * c + k];
			if (dim == 1) {
				rstore(off, dist, cm, g0, row_R(g0, tran));
				rtag(off, MJB_CNSTR_CONTACT_FRICTIONLESS, c);
			} else if (m.cone == MJB_CONE_ELLIPTIC) {
				// row 0 = normal (pos = dist), rows 1.. = friction directions (pos = margin = 0);
				// R_j = R_0 mu^2 / friction_j^2 with mu = friction_0 / sqrt(impratio)
				const RowGain gf = row_gain(m, solref, solimp, 0.0, 0.0);
				const double mu = fri[0] / sqrt(fmax(MJB_MINVAL, m.impratio[0]));
				const double R0 = row_R(g0, tran);
				for (int k = 0; k < 6; k++) {
					if (k >= dim) break;
					if (k == 0) rstore(off, dist, cm, g0, R0);
					else rstore(off + k, 0.0, 0.0, gf, fmax(MJB_MINVAL, R0 * mu * mu / (fri[k - 1] * fri[k - 1])));
					rtag(off + k, MJB_CNSTR_CONTACT_ELLIPTIC, c);
				}
			} else {
				// pyramidal: every row gets Rpy = 2 mu^2 R(first row), R(first row) from diagApprox = tran + friction_0^2 tran
				const double mu = fri[0] / sqrt(fmax(MJB_MINVAL, m.impratio[0]));
				const double Rpy = fmax(MJB_MINVAL, 2 * mu * mu * row_R(g0, tran + fri[0] * fri[0] * tran));
				for (int k = 0; k < 10; k++) {
					if (k >= 2 * (dim - 1)) break;
					rstore(off + k, dist, cm, g0, Rpy);
					rtag(off + k, MJB_CNSTR_CONTACT_PYRAMIDAL, c);
				}
			}
		}
	}
	gsync<G>();
#ifdef MJB_PROFILE_SUB
	EPROF(27);
#endif
#ifdef MJB_PROFILE_MK
	EPROF(21);
#endif
	for (int r = lane; r < nefc; r += G) {
		if constexpr (TAG == 4) {
			if (r >= L.rcap) {
				gb.force[r] = 0;
				continue;
			}
		}
		f[L.efc_force + r] = 0;
	}
	if (lane == 0) fi[L.nefc] = nefc;
	gsync<G>();
	// connect / weld Jacobian columns: one (equality, dof) pair per lane.  J = body1 - body2 (points differ, so a
	// dof moving both bodies does not cancel); weld rotation rows 0.5 ts imag(neg(q2) (0, w1 - w2) q1 relpose)
	for (int t = lane; t < neq * nv; t += G) {
		const int eq = t / nv, i = t - eq * nv;
		const int type = m.eq_type[eq];
		const int adr = fi[L.iscratch + eq];  // first row of the equality (-1: inactive / dropped)
		if (adr < 0 || type == MJB_EQ_JOINT || type == MJB_EQ_TENDON) continue;
		const int id0 = m.eq_obj1id[eq], id1 = m.eq_obj2id[eq];
		const bool in0 = dof_moves_body(m, id0, i), in1 = dof_moves_body(m, id1, i);
		double jp[3] = { 0, 0, 0 }, jr[3] = { 0, 0, 0 };
		EqGeom g;
		eq_geometry(m, L, f, eq, g);
		if (in0 || in1) {
			double cd[6], root[3];
			ld6(cd, f + L.cdof + 6 * i);
			ld3(root, f + L.subtree_com + 3 * m.body_rootid[m.dof_bodyid[i]]);
			if (in0) {
				const double off3[3] = { g.pos0[0] - root[0], g.pos0[1] - root[1], g.pos0[2] - root[2] };
				double c3[3];
				cross3(c3, cd, off3);
				for (int k = 0; k < 3; k++) { jp[k] += c3[k] + cd[3 + k]; jr[k] += cd[k]; }
			}
			if (in1) {
				const double off3[3] = { g.pos1[0] - root[0], g.pos1[1] - root[1], g.pos1[2] - root[2] };
				double c3[3];
				cross3(c3, cd, off3);
				for (int k = 0; k < 3; k++) { jp[k] -= c3[k] + cd[3 + k]; jr[k] -= cd[k]; }
			}
		}
		for (int k = 0; k < 3; k++) jrow(adr + k)[i] = jp[k];
		if (type == MJB_EQ_WELD) {
			const double ts = f[L.eqparam + 19 * eq + 1 + 10];
			const double *q = g.quat1;
			const double qa[4] = { -q[1] * jr[0] - q[2] * jr[1] - q[3] * jr[2], q[0] * jr[0] + q[2] * jr[2] - q[3] * jr[1],
				                   q[0] * jr[1] + q[3] * jr[0] - q[1] * jr[2], q[0] * jr[2] + q[1] * jr[1] - q[2] * jr[0] };
			double q3[4];
			qmul(q3, qa, g.quat);
			for (int k = 0; k < 3; k++) jrow(adr + 3 + k)[i] = 0.5 * ts * q3[1 + k];
		}
	}
#ifdef MJB_PROFILE_SUB
	EPROF(28);
#endif
#ifdef MJB_PROFILE_MK
	EPROF(22);
#endif
	// contact Jacobian rows: one (contact, dof) pair per lane
	const int npair = ncon * nv;
	for (int t = lane; t < npair; t += G) {
		const int c = t / nv, i = t - c * nv;
		const int adr = fi[L.contact_efc_address + c];
		if (adr < 0) continue;
		const int dim = fi[L.contact_dim + c];
		const int nrow = dim == 1 ? 1 : (m.cone == MJB_CONE_ELLIPTIC ? dim : 2 * (dim - 1));
		if (adr + nrow > nefc) continue;
		const int b1 = m.geom_bodyid[fi[L.contact_geom + 2 * c]], b2 = m.geom_bodyid[fi[L.contact_geom + 2 * c + 1]];
		const bool in2 = dof_moves_body(m, b2, i), in1 = dof_moves_body(m, b1, i);
		double jdp[3] = { 0, 0, 0 }, jdr[3] = { 0, 0, 0 };
		if (in1 != in2) {  // a dof moving both bodies cancels exactly in the reference too (same point, same dof)
			double cd[6], pos[3], root[3];
			ld6(cd, f + L.cdof + 6 * i);
			ld3(pos, f + L.contact_pos + 3 * c);
			ld3(root, f + L.subtree_com + 3 * m.body_rootid[m.dof_bodyid[i]]);
			const double off[3] = { pos[0] - root[0], pos[1] - root[1], pos[2] - root[2] };
			cross3(jdp, cd, off);
			const double sg = in2 ? 1.0 : -1.0;
			jdp[0] = sg * (jdp[0] + cd[3]); jdp[1] = sg * (jdp[1] + cd[4]); jdp[2] = sg * (jdp[2] + cd[5]);
			jdr[0] = sg * cd[0]; jdr[1] = sg * cd[1]; jdr[2] = sg * cd[2];
		}
		double fr[9];
		ld9(fr, f + L.contact_frame + 9 * c);
		const double j0 = dot3(fr, jdp);
		if (dim == 1) {
			jrow(adr)[i] = j0;
		} else if (m.cone == MJB_CONE_ELLIPTIC) {
			jrow(adr)[i] = j0;
			for (int k = 1; k < dim; k++)
				jrow(adr + k)[i] = k < 3 ? dot3(fr + 3 * k, jdp) : dot3(fr + 3 * (k - 3), jdr);
		} else {
			int r = adr;
			for (int k = 1; k < dim; k++) {
				const double jk = k < 3 ? dot3(fr + 3 * k, jdp) : dot3(fr + 3 * (k - 3), jdr);
				const double mu = f[L.contact_friction + 5 * c + (k - 1)];
				jrow(r)[i] = j0 + mu * jk;
				jrow(r + 1)[i] = j0 - mu * jk;
				r += 2;
			}
		}
	}
	gsync<G>();
	if constexpr (TAG == 4) {
		if (Jg && nefc > L.jrows) {  // (wave-uniform) more rows than the frame holds: the solver reads all of J from HBM
			MJB_KEEP_BRANCH();
			for (int t = lane; t < L.jrows * nv; t += G) Jg[t] = f[L.efc_J + t];
			if (L.rcap < m.nefcmax) {  // ... and the rest of the row data with it
				for (int r = lane; r < L.rcap; r += G) {
					gb.D[r] = f[L.efc_D + r];
					gb.aref[r] = f[L.efc_aref + r];
					gb.b[r] = f[L.efc_b + r];
					gb.force[r] = 0;
					if (nfr) gb.fl[r] = f[L.efc_frictionloss + r];
					gb.type[r] = fi[L.efc_type + r];
					gb.id[r] = fi[L.efc_id + r];
				}
			}
			__threadfence();  // the rows are read back by other lanes through the vector cache
			gsync<G>();
		}
	}
#ifdef MJB_PROFILE_SUB
	EPROF(29);
#endif
#ifdef MJB_PROFILE_MK
	EPROF(23);
#endif
}

// ------------------------------------------------------------------------------------------------
// A7  project: B = (M^-1 J')' row by row -- one ROW per lane, each lane runs the sparse solve serially on
// its own row (loop structure is wave-uniform: the scalar table loads are shared by all rows)
// ------------------------------------------------------------------------------------------------
template <int G, int TAG> STAGE void project_constraint(CModel m, CLayout L, const EnvLite &e)
{
	double *f = e.f;
	const int nefc = wave_uniform<TAG>(e.fi[L.nefc]), nv = m.nv;
	if (nefc == 0) return;
	const double *LD = f + L.qLD, *di = f + L.qLDiagInv;
	for (int r0 = 0; r0 < nefc; r0 += G) {
		const int r = r0 + e.lane;
		const bool act = r < nefc;
		double *x = f + L.efc_B + (act ? r : 0) * nv;
		if (act)
			for (int k = 0; k < nv; k++) x[k] = f[L.efc_J + r * nv + k];
#pragma nounroll
		for (int i = nv - 1; i >= 0; i--) {
			const int na = m.dof_rec[4 * i + 1], ii = m.dof_rec[4 * i];
			if (na <= 0 || !act) continue;
			const double xi = x[i];
			for (int a = 0; a < na; a++) x[m.M_coldof[ii + 1 + a]] -= LD[ii + 1 + a] * xi;
		}
		if (act)
			for (int k = 0; k < nv; k++) x[k] *= di[k];
#pragma nounroll
		for (int i = 0; i < nv; i++) {
			const int na = m.dof_rec[4 * i + 1], ii = m.dof_rec[4 * i];
			if (na <= 0 || !act) continue;
			double acc = x[i];
			for (int a = 0; a < na; a++) acc -= LD[ii + 1 + a] * x[m.M_coldof[ii + 1 + a]];
			x[i] = acc;
		}
	}
	gsync<G>();
}

// A7 for nv <= 16: the solve of row r runs in lane r's REGISTERS.  The sparse L'DL factor is first spread into a
// packed dense strictly-lower triangle (entry (i, j) at i (i - 1) / 2 + j, zeros where j is no ancestor of i) parked in
// the efc_B region itself, so that the fully unrolled substitution reads every L entry at a compile-time offset and a
// wave-uniform address (one LDS broadcast per entry) and x never leaves the registers: 240 fma per row with no
// load-modify-store chain, against two dependent LDS round trips per factor entry in the generic version.
// Rows / columns >= nv of the triangle are zero and x[k >= nv] = 0, so the unrolled sweeps need no guards.
template <int G, int TAG> STAGE void project_constraint_dense16(CModel m, CLayout L, const EnvLite &e)
{
	static_assert(G == 64, "one constraint row per lane of the wavefront");
	double *f = e.f;
	const int nefc = wave_uniform<TAG>(e.fi[L.nefc]), nv = m.nv, lane = e.lane;
	if (nefc == 0) return;
	double *Ld = f + L.tri;  // (compact layout: inside the region kinematics / crb / rne share -- nobody else is alive here)
	for (int t = lane; t < 120; t += G) Ld[t] = 0;
	gsync<G>();
	for (int en = lane; en < m.nM; en += G) {
		const int i = m.M_rowdof[en], j = m.M_coldof[en];
		if (i != j) Ld[i * (i - 1) / 2 + j] = f[L.qLD + en];
	}
	gsync<G>();
	const double *di = f + L.qLDiagInv;
	// one 64-row block per trip (PGS: nefc <= 128); one copy of the substitution in the instruction stream
#pragma nounroll
	for (int r0 = 0; r0 < nefc; r0 += 64) {
		const int r = r0 + lane;
		const bool act = r < nefc;
		const double *Jr = f + L.efc_J + (act ? r : 0) * nv;
		double x[16];
#pragma unroll
		for (int k = 0; k < 16; k++) {
			const double v = Jr[k < nv ? k : 0];
			x[k] = (k < nv && act) ? v : 0.0;
		}
		// x <- L^-T x: once x[i] is final, every x[j < i] takes its share (independent fma)
#pragma unroll
		for (int i = 15; i >= 1; i--) {
#pragma unroll
			for (int j = 0; j < i; j++) x[j] -= Ld[i * (i - 1) / 2 + j] * x[i];
		}
#pragma unroll
		for (int k = 0; k < 16; k++) x[k] *= di[k < nv ? k : 0];
		// x <- L^-1 x, column by column (same summation order as the row form)
#pragma unroll
		for (int j = 0; j < 15; j++) {
#pragma unroll
			for (int i = j + 1; i < 16; i++) x[i] -= Ld[i * (i - 1) / 2 + j] * x[j];
		}
		if (act) {
			double *Br = f + L.efc_B + r * nv;
#pragma unroll
			for (int k = 0; k < 16; k++)
				if (k < nv) Br[k] = x[k];
		}
	}
	gsync<G>();
}

// A8  reference accelerations: efc_vel = J qvel, aref = -B vel - K imp (pos - margin)
template <int G, int TAG> STAGE void reference_constraint(CModel m, CLayout L, CState st, const EnvLite &e)
{
	double *f = e.f;
	const int nefc = wave_uniform<TAG>(e.fi[L.nefc]), nv = m.nv;
	// (aref / bg: where row_store parked K imp (pos - margin) and B on a lean frame)
	auto rows = [&](const double *Jb, double *aref, const double *bg) {
	for (int r = e.lane; r < nefc; r += G) {
		double s = 0;
		for (int k = 0; k < nv; k++) s += Jb[r * nv + k] * f[L.qvel + k];
		if (L.efc_KBIP >= 0) {
			f[L.efc_vel + r] = s;
			const double *kb = f + L.efc_KBIP + 4 * r;
			f[L.efc_aref + r] = -kb[1] * s - kb[0] * kb[2] * (f[L.efc_pos + r] - f[L.efc_margin + r]);
		} else {
			aref[r] = -bg[r] * s - aref[r];  // (row_store left B and K imp (pos - margin) here)
		}
	}
	};
	if constexpr (TAG == 4) {  // (see make_constraint: all of J -- on a row-capped frame all row data -- sits in HBM when the rows outnumber the frame's share)
		if (st.efc_Jg && nefc > L.jrows) {
			MJB_KEEP_BRANCH();
			const RowBlock gb = mjb_rowblock(st.efc_Jg + (size_t)e.env * st.efc_Jg_stride, m.nefcmax, nv, m.nconmax, L.hcs);
			if (L.rcap < m.nefcmax) rows(gb.J, gb.aref, gb.b);
			else rows(gb.J, f + L.efc_aref, f + L.efc_b);
			__threadfence();
		} else {
			MJB_KEEP_BRANCH();
			rows(f + L.efc_J, f + L.efc_aref, f + L.efc_b);
		}
	} else
		rows(f + L.efc_J, f + L.efc_aref, f + L.efc_b);
	gsync<G>();
}

DEVI double wave_bcast_c(double v, int srclane)
{
	return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), srclane), __builtin_amdgcn_readlane(__double2loint(v), srclane));
}
// wave-wide sum (G == 64: the env owns the whole wavefront)
DEVI double wave_sum(double v)
{
	v = row_sum<16>(v);
	// combine the four rows: readlane the row totals (lanes 0,16,32,48 hold them after the butterfly)
	double t = 0;
	for (int r = 0; r < 4; r++)
		t += __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 16 * r),
		                      __builtin_amdgcn_readlane(__double2loint(v), 16 * r));
	return t;
}
// three wave-wide sums side by side: the butterfly steps of the three values are pinned level by level (left to itself the scheduler
// runs the three reductions one after the other, each a chain of dpp moves and dependent adds: ~250 cycles apiece in the line search's
// trial points)
DEVI void wave_sum3(double &a, double &b, double &c)
{
#define MJB_DPP3(ctrl)                            \
	MJB_DPP_STEP(a, ctrl);                        \
	MJB_DPP_STEP(b, ctrl);                        \
	MJB_DPP_STEP(c, ctrl);                        \
	asm volatile("" : "+v"(a), "+v"(b), "+v"(c))
	MJB_DPP3(0xB1);
	MJB_DPP3(0x4E);
	MJB_DPP3(0x141);
	MJB_DPP3(0x140);
#undef MJB_DPP3
	double ta = 0, tb = 0, tc = 0;
#pragma unroll
	for (int r = 0; r < 4; r++) {
		ta += wave_bcast_c(a, 16 * r);
		tb += wave_bcast_c(b, 16 * r);
		tc += wave_bcast_c(c, 16 * r);
	}
	a = ta;
	b = tb;
	c = tc;
}
DEVI double wave_bcast(double v, int srclane)  // srclane must be wave-uniform
{
	return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), srclane),
	                        __builtin_amdgcn_readlane(__double2loint(v), srclane));
}

// One Gauss-Seidel sweep over the rows: row i lives in lane i together with its row of AR (registers).  Every lane
// proposes the update of ITS row from its running residual (only lane i's proposal at step i is ever used, so the
// lanes can work from the force they had at the start of the sweep); lane i's delta is broadcast with v_readlane,
// filed into lane i of `dvec` with v_writelane, and every lane does res += AR[.][i] * delta.  Dependent chain per row:
// fma, max, add, readlane, fma.
// The reference's guard "undo the update if it RAISES the cost by more than 1e-10" is dead code here (a 1-D minimiser
// clipped to an interval never raises a convex cost; in numbers:) with
// Aii = J M^-1 J' + R > 0 the unclipped step changes the cost by -0.5 res^2 / Aii and the clipped one by
// -f (res - 0.5 f Aii) with res >= f Aii, both <= 0 beyond any rounding, so it is not evaluated.
template <bool BOX, int NR>  // BOX: some row has an upper force limit too (dry friction: |f| <= frictionloss); NR: rows held in registers
DEVI void pgs_sweep(const double (&AR)[NR], const int nefc, const double lo, const double hi, const double ARinv, double &res,
                    const double frc, double &dvec)
{
#pragma unroll
	for (int i = 0; i < NR; i++) {
		if ((i & 3) == 0) {
			if (i >= nefc) break;  // rows beyond nefc in a group of four propose delta == 0
			MJB_KEEP_BRANCH();
		}
		double fn = __builtin_fmax(frc - res * ARinv, lo);
		if (BOX) fn = __builtin_fmin(fn, hi);
		const double delta = fn - frc;
		const int dh = __builtin_amdgcn_readlane(__double2hiint(delta), i), dl = __builtin_amdgcn_readlane(__double2loint(delta), i);
		res += AR[i] * __hiloint2double(dh, dl);
		int vh = __double2hiint(dvec), vl = __double2loint(dvec);  // (clang has no writelane builtin)
		asm("v_writelane_b32 %0, %1, %2" : "+v"(vh) : "s"(dh), "n"(i));
		asm("v_writelane_b32 %0, %1, %2" : "+v"(vl) : "s"(dl), "n"(i));
		dvec = __hiloint2double(vh, vl);
	}
}

// ------------------------------------------------------------------------------------------------
// PGS with elliptic cones: block update of one contact (oracle/mjo_constraint.c pgs_cone_block, cone_qcqp; the role
// of mj_solPGS's ray step and mju_QCQP2 / 3 / N).  All values are wave-uniform; DIM = condim (3, 4 or 6) is a template
// parameter so that every small array stays in statically indexed registers.
// ------------------------------------------------------------------------------------------------
template <int N> DEVI bool chol_solve_small(const double (&A)[N * N], const double la, const double (&rhs)[N], double (&x)[N])
{
	double Lc[N * N];
#pragma unroll
	for (int i = 0; i < N; i++) {
#pragma unroll
		for (int j = 0; j <= i; j++) {
			double sum = A[i * N + j] + (i == j ? la : 0.0);
#pragma unroll
			for (int k = 0; k < j; k++) sum -= Lc[i * N + k] * Lc[j * N + k];
			if (i == j) {
				if (sum < MJB_MINVAL) return false;
				Lc[i * N + i] = sqrt(sum);
			} else {
				Lc[i * N + j] = sum / Lc[j * N + j];
			}
		}
	}
#pragma unroll
	for (int i = 0; i < N; i++) {
		double sum = rhs[i];
#pragma unroll
		for (int k = 0; k < i; k++) sum -= Lc[i * N + k] * x[k];
		x[i] = sum / Lc[i * N + i];
	}
#pragma unroll
	for (int i = N - 1; i >= 0; i--) {
		double sum = x[i];
#pragma unroll
		for (int k = i + 1; k < N; k++) sum -= Lc[k * N + i] * x[k];
		x[i] = sum / Lc[i * N + i];
	}
	return true;
}

// minimise 0.5 y'Ay + y'b  s.t.  sum (y_j / d_j)^2 <= r^2 : Newton's method on the multiplier in the scaled variable
template <int N> DEVI bool cone_qcqp(double (&y)[N], const double (&A)[N * N], const double (&b)[N], const double (&dsc)[N], const double r)
{
	double As[N * N], nb[N], z[N], w[N];
#pragma unroll
	for (int i = 0; i < N; i++) {
		nb[i] = -(b[i] * dsc[i]);
		z[i] = 0;
		w[i] = 0;
#pragma unroll
		for (int j = 0; j < N; j++) As[i * N + j] = A[i * N + j] * dsc[i] * dsc[j];
	}
	double la = 0;
	bool ok = true;
#pragma nounroll
	for (int iter = 0; iter < 20; iter++) {
		ok = chol_solve_small<N>(As, la, nb, z);
		if (!ok) break;
		double val = -r * r;
#pragma unroll
		for (int i = 0; i < N; i++) val += z[i] * z[i];
		if (val < 1e-10) break;
		chol_solve_small<N>(As, la, z, w);
		double deriv = 0;
#pragma unroll
		for (int i = 0; i < N; i++) deriv -= 2 * z[i] * w[i];
		const double delta = -val / deriv;
		if (delta < 1e-10) break;
		la += delta;
	}
#pragma unroll
	for (int i = 0; i < N; i++) y[i] = ok ? z[i] * dsc[i] : 0.0;
	return ok && la != 0;
}

// forces fc[0..DIM) updated in place from the block residual res (at the old forces) and the block Ac of AR (row stride 6)
template <int DIM> DEVI void pgs_cone_block(double (&fc)[6], const double (&res)[6], const double *Ac, const double *mu5)
{
	double old[DIM], A[DIM * DIM];
#pragma unroll
	for (int a = 0; a < DIM; a++) {
		old[a] = fc[a];
#pragma unroll
		for (int c = 0; c < DIM; c++) A[a * DIM + c] = Ac[6 * a + c];
	}
	if (fc[0] < MJB_MINVAL) {
		fc[0] -= res[0] / A[0];
		if (fc[0] < 0) fc[0] = 0;
#pragma unroll
		for (int a = 1; a < DIM; a++) fc[a] = 0;
	} else {
		double denom = 0, vr = 0;
#pragma unroll
		for (int a = 0; a < DIM; a++) {
			double sa = 0;
#pragma unroll
			for (int c = 0; c < DIM; c++) sa += A[a * DIM + c] * old[c];
			denom += old[a] * sa;
			vr += old[a] * res[a];
		}
		if (denom >= MJB_MINVAL) {
			double x = -vr / denom;
			if (fc[0] + x * old[0] < 0) x = -fc[0] / old[0];
#pragma unroll
			for (int a = 0; a < DIM; a++) fc[a] += x * old[a];
		}
	}
	if (fc[0] < MJB_MINVAL) {
#pragma unroll
		for (int a = 1; a < DIM; a++) fc[a] = 0;
		return;
	}
	constexpr int N = DIM - 1;
	double Af[N * N], bf[N], y[N], mu[N];
#pragma unroll
	for (int a = 1; a < DIM; a++) {
		double bc = res[a];
#pragma unroll
		for (int c = 0; c < DIM; c++) bc -= A[a * DIM + c] * old[c];
		bf[a - 1] = bc + A[a * DIM] * fc[0];
		mu[a - 1] = mu5[a - 1];
#pragma unroll
		for (int c = 1; c < DIM; c++) Af[(a - 1) * N + (c - 1)] = A[a * DIM + c];
	}
	const bool active = cone_qcqp<N>(y, Af, bf, mu, fc[0]);
	if (active) {  // put the result exactly on the cone
		double sn = 0;
#pragma unroll
		for (int a = 0; a < N; a++) sn += (y[a] / mu[a]) * (y[a] / mu[a]);
		sn = sqrt(sn);
		if (sn > MJB_MINVAL) {
#pragma unroll
			for (int a = 0; a < N; a++) y[a] *= fc[0] / sn;
		}
	}
#pragma unroll
	for (int a = 0; a < N; a++) fc[1 + a] = y[a];
}

// primal cone update of one contact (forces only; oracle cone_eval): used by the PGS warmstart
DEVI void cone_force(const double mu, const double *fri, const double *D, const int dim, const double *jar, double *force)
{
	double U[6], TT = 0;
	U[0] = mu * jar[0];
	for (int j = 1; j < 6; j++) {
		U[j] = 0;
		if (j < dim) {
			U[j] = fri[j - 1] * jar[j];
			TT += U[j] * U[j];
		}
	}
	const double N = U[0], T = sqrt(TT);
	if (N >= mu * T) {
		for (int j = 0; j < 6; j++)
			if (j < dim) force[j] = 0;
	} else if (mu * N + T <= 0) {
		for (int j = 0; j < 6; j++)
			if (j < dim) force[j] = -D[j] * jar[j];
	} else {
		const double Dm = D[0] / (mu * mu * (1 + mu * mu)), NmT = N - mu * T;
		force[0] = -Dm * NmT * mu;
		for (int j = 1; j < 6; j++)
			if (j < dim) force[j] = -force[0] / T * U[j] * fri[j - 1];
	}
}

// res += sum_a AR[i + a] * d[a] with a statically indexed AR for a wave-uniform, dynamic i (a switch over the 64 rows)
#define MJB_APPLY_CASE(I)                                                                      \
	case I:                                                                                    \
		res += AR[I] * d[0];                                                                   \
		if (multi) {                                                                           \
			res += AR[(I) + 1 < 64 ? (I) + 1 : 63] * ((I) + 1 < 64 ? d[1] : 0.0);              \
			res += AR[(I) + 2 < 64 ? (I) + 2 : 63] * ((I) + 2 < 64 ? d[2] : 0.0);              \
			res += AR[(I) + 3 < 64 ? (I) + 3 : 63] * ((I) + 3 < 64 ? d[3] : 0.0);              \
			res += AR[(I) + 4 < 64 ? (I) + 4 : 63] * ((I) + 4 < 64 ? d[4] : 0.0);              \
			res += AR[(I) + 5 < 64 ? (I) + 5 : 63] * ((I) + 5 < 64 ? d[5] : 0.0);              \
		}                                                                                      \
		break;
#define MJB_APPLY_CASE4(I) MJB_APPLY_CASE(I) MJB_APPLY_CASE((I) + 1) MJB_APPLY_CASE((I) + 2) MJB_APPLY_CASE((I) + 3)
#define MJB_APPLY_CASE16(I) MJB_APPLY_CASE4(I) MJB_APPLY_CASE4((I) + 4) MJB_APPLY_CASE4((I) + 8) MJB_APPLY_CASE4((I) + 12)
DEVI void pgs_apply_rows(const double (&AR)[64], const int i, const double (&d)[6], const bool multi, double &res)
{
	switch (i) {
		MJB_APPLY_CASE16(0) MJB_APPLY_CASE16(16) MJB_APPLY_CASE16(32) MJB_APPLY_CASE16(48)
	default: break;
	}
}

// one sweep over a model with elliptic contacts: scalar rows as in pgs_sweep, contacts as blocks; forces updated in place
DEVI void pgs_sweep_elliptic(const double (&AR)[64], const int nefc, const int lane, const double lo, const double hi,
                             const double ARinv, const int rtype, const int rcon, const int rdim, double &res, double &frc,
                             const double *Hc, const double *cfriction)
{
	int i = 0;
#pragma nounroll
	while (i < nefc) {
		const int ti = __builtin_amdgcn_readlane(rtype, i);
		double d[6] = { 0, 0, 0, 0, 0, 0 };
		if (ti != MJB_CNSTR_CONTACT_ELLIPTIC) {
			MJB_KEEP_BRANCH();
			const double fn = __builtin_fmin(__builtin_fmax(frc - res * ARinv, lo), hi);
			const double delta = fn - frc;
			d[0] = wave_bcast(delta, i);
			pgs_apply_rows(AR, i, d, false, res);
			if (lane == i) frc += delta;
			i++;
		} else {
			MJB_KEEP_BRANCH();
			const int con = __builtin_amdgcn_readlane(rcon, i), dim = __builtin_amdgcn_readlane(rdim, i);
			double fc[6], rb[6], old[6];
#pragma unroll
			for (int a = 0; a < 6; a++) {
				const int src = i + a < 64 ? i + a : 63;
				rb[a] = a < dim ? wave_bcast(res, src) : 0.0;
				fc[a] = a < dim ? wave_bcast(frc, src) : 0.0;
				old[a] = fc[a];
			}
			const double *Ac = Hc + 36 * con, *mu5 = cfriction + 5 * con;
			if (dim == 3) pgs_cone_block<3>(fc, rb, Ac, mu5);
			else if (dim == 4) pgs_cone_block<4>(fc, rb, Ac, mu5);
			else pgs_cone_block<6>(fc, rb, Ac, mu5);
			// the reference's guard: an update that raises the block cost 0.5 d'Ac d + d'res by more than 1e-10 is dropped
			double change = 0;
#pragma unroll
			for (int a = 0; a < 6; a++) {
				d[a] = a < dim ? fc[a] - old[a] : 0.0;
			}
#pragma unroll
			for (int a = 0; a < 6; a++) {
				double sa = 0;
#pragma unroll
				for (int c = 0; c < 6; c++) sa += (a < dim && c < dim) ? Ac[6 * a + c] * d[c] : 0.0;
				change += 0.5 * d[a] * sa + d[a] * rb[a];
			}
			if (!(change > 1e-10)) {
				MJB_KEEP_BRANCH();
				pgs_apply_rows(AR, i, d, true, res);
#pragma unroll
				for (int a = 0; a < 6; a++)
					if (lane == i + a) frc += d[a];
			}
			i += dim;
		}
	}
}

// x <- M^-1 x for nv <= 16 with x in 16 statically indexed registers: the sparse L'DL factor spread into a packed dense
// strictly-lower triangle Ld (entry (i, j) at i (i - 1) / 2 + j, zeros where j is no ancestor of i; rows / columns >= nv are
// zero and x[k >= nv] = 0, so the unrolled sweeps need no guards).  Every L entry is read at a compile-time offset and a
// wave-uniform address (one LDS broadcast per entry): 240 fma per vector with no load-modify-store chain.
DEVI void tri_build(CModel m, CLayout L, double *f, double *Ld, int lane)
{
	for (int t = lane; t < 120; t += 64) Ld[t] = 0;
	gsync<64>();
	for (int en = lane; en < m.nM; en += 64) {
		const int i = m.M_rowdof[en], j = m.M_coldof[en];
		if (i != j) Ld[i * (i - 1) / 2 + j] = f[L.qLD + en];
	}
	gsync<64>();
}
// (the triangle's base address as seen by one stage of the substitution: Ld plus a zero the compiler cannot see through, computed
//  from an x that is final MJB_TRI_LAG stages earlier.  Left alone, the scheduler issues the loads of ALL stages up front -- 240 registers of
//  triangle -- and the 256-register PGS kernel spilled and reloaded eight 16-byte loads at both call sites: 16 KB of scratch traffic
//  per env-step, essentially all of the kernel's write-backs.  With the address dependency MJB_TRI_LAG stages of loads are in flight.)
#ifndef MJB_TRI_LAG
#define MJB_TRI_LAG 8  // stages of triangle loads in flight (MI355X, config 3: lag 4 21.7 M, 6 22.0 M, 8 22.3 M, 12 22.0 M env-steps/s; 8 spills each)
#endif
DEVI const double *tri_base(const double *Ld, double dep)
{
	int z;
	asm volatile("v_and_b32 %0, 0, %1" : "=v"(z) : "v"(__double2loint(dep)));
	return Ld + z;
}
DEVI void tri_solve(double (&x)[16], const double *Ld, const double *di, int nv)
{
	// x <- L^-T x: once x[i] is final, every x[j < i] takes its share (independent fma)
#pragma unroll
	for (int i = 15; i >= 1; i--) {
		const double *Li = i <= 15 - MJB_TRI_LAG ? tri_base(Ld, x[i + MJB_TRI_LAG - 1]) : Ld;  // (x[i + LAG - 1] is final once row i + LAG has been applied)
#pragma unroll
		for (int j = 0; j < i; j++) x[j] -= Li[i * (i - 1) / 2 + j] * x[i];
	}
#pragma unroll
	for (int k = 0; k < 16; k++) x[k] *= di[k < nv ? k : 0];
	// x <- L^-1 x, column by column (same summation order as the row form)
#pragma unroll
	for (int j = 0; j < 15; j++) {
		const double *Lj = j >= MJB_TRI_LAG ? tri_base(Ld, x[j - MJB_TRI_LAG + 1]) : Ld;  // (x[j - LAG + 1] is final once column j - LAG has been applied)
#pragma unroll
		for (int i = j + 1; i < 16; i++) x[i] -= Lj[i * (i - 1) / 2 + j] * x[j];
	}
}

// ------------------------------------------------------------------------------------------------
// A13 PGS beyond 64 rows (64 < nefc <= 128, scalar rows: equality / friction / limit / frictionless / pyramidal).
// A row of AR no longer fits the lane's registers, so this path is AR-free: row r lives in lane r % 64, slot r / 64, with
// its b, R, 1 / A_rr, bounds and force in registers; the residual of row i is  b_i + R_i f_i + J_i . w  with
// w = M^-1 J' f kept one element per lane (lanes 0 .. nv-1) and updated by  w += B_i delta  after every row.  Same
// fixed point and same sweep order as the register path; ~4x its cost per row update, which is why it only runs for the
// env-steps whose row count actually exceeds 64 (wave-uniform branch in fwd_constraint_pgs; out of line so that the
// common path pays neither registers nor instruction-cache lines for it).
// ------------------------------------------------------------------------------------------------
// (TAG: the kernel variant this copy belongs to -- out-of-line functions shared by kernels with different register budgets are
//  compiled for the loosest one, which would cost the capped kernel its second wave per SIMD)
template <int G, int TAG> __device__ __attribute__((noinline)) void fwd_constraint_pgs_large(CModel m, CLayout L, const EnvLite e, const double *Brows)
{
	static_assert(G == 64, "the constraint solver maps rows to the 64 lanes of one wavefront");
	double *f = e.f;
	int *fi = e.fi;
	const int lane = e.lane, nv = m.nv, nefc = __builtin_amdgcn_readfirstlane(fi[L.nefc]);
	constexpr int S = 2;
	const bool warm = !(m.disableflags & MJB_DSBL_WARMSTART);
	const bool small = nv <= 16;
	bool act[S];
	double b[S], Rr[S], ARinv[S], lo[S], hi[S], frc[S];
	auto b_entry = [&](int i) -> double { return lane < nv ? Brows[i * nv + lane] : 0.0; };  // B_i[lane]
#pragma unroll
	for (int s = 0; s < S; s++) {
		const int r = lane + 64 * s;
		act[s] = r < nefc;
		const int rr = act[s] ? r : 0;
		const double *Jr = f + L.efc_J + rr * nv, *Br = Brows + rr * nv;
		double jq = 0, jb = 0, jw = 0;
		for (int k = 0; k < nv; k++) {
			const double j = Jr[k];
			jq += j * f[L.qacc_smooth + k];
			jb += j * Br[k];
			jw += j * f[L.qacc_warmstart + k];
		}
		const double aref = f[L.efc_aref + rr];
		const bool bilateral = act[s] && fi[L.efc_type + rr] == MJB_CNSTR_EQUALITY;
		const double floss = (act[s] && m.nfriction > 0) ? f[L.efc_frictionloss + rr] : 0.0;
		const bool friction = floss > 0;
		Rr[s] = act[s] ? f[L.efc_R + rr] : 0.0;
		b[s] = act[s] ? jq - aref : 0.0;
		ARinv[s] = act[s] ? 1.0 / (jb + Rr[s]) : 0.0;
		lo[s] = bilateral ? -__builtin_huge_val() : (friction ? -floss : 0.0);
		hi[s] = friction ? floss : __builtin_huge_val();
		double fr = 0;
		if (act[s] && warm) {
			const double jar = jw - aref;
			fr = (jar < 0 || bilateral || friction) ? -(L.efc_D >= 0 ? f[L.efc_D + rr] : 1.0 / Rr[s]) * jar : 0.0;  // (lean frame: no efc_D)
			if (friction) fr = __builtin_fmin(__builtin_fmax(fr, -floss), floss);
		}
		frc[s] = fr;
		if (act[s]) {
			f[L.efc_b + rr] = b[s];
			f[L.efc_force + rr] = fr;
		}
	}
	gsync<G>();
	double *w = f + L.qacc;  // w = M^-1 J' f = B' f parked in the qacc slot between sweeps (qacc = qacc_smooth + w at the end)
	double wreg = 0;
#pragma nounroll
	for (int i = 0; i < nefc; i++) wreg += b_entry(i) * f[L.efc_force + i];
	// cost = 0.5 f'ARf + f'b = sum_r 0.5 f_r (res_r + b_r), res_r = b_r + R_r f_r + J_r . w
	auto cost_of = [&]() -> double {
		if (lane < nv) w[lane] = wreg;
		gsync<G>();
		double c = 0;
#pragma unroll
		for (int s = 0; s < S; s++) {
			const double *Jr = f + L.efc_J + (act[s] ? lane + 64 * s : 0) * nv;
			double dot = 0;
			for (int k = 0; k < nv; k++) dot += Jr[k] * w[k];
			const double res = b[s] + Rr[s] * frc[s] + dot;
			c += act[s] ? 0.5 * frc[s] * (res + b[s]) : 0.0;
		}
		gsync<G>();
		return wave_sum(c);
	};
	double cost = cost_of();
	if (cost > 0 || !warm) {
		frc[0] = frc[1] = 0;
		wreg = 0;
		cost = 0;
	}
	const double scale = 1.0 / (MP_MEANINERTIA(m, e) * (nv > 1 ? nv : 1));
	const double tol = m.tolerance[0];
	int iter = 0;
	while (iter < m.iterations) {
#pragma unroll
		for (int s = 0; s < S; s++) {
			const int i0 = 64 * s, i1 = nefc < i0 + 64 ? nefc : i0 + 64;
			double jn = (lane < nv && i0 < i1) ? f[L.efc_J + i0 * nv + lane] : 0.0;
			double bn = i0 < i1 ? b_entry(i0) : 0.0;
#pragma nounroll
			for (int i = i0; i < i1; i++) {
				const double ji = jn, bi = bn;
				const int nx = i + 1 < nefc ? i + 1 : i;  // the next row's entries (off the dependent chain)
				if (lane < nv) jn = f[L.efc_J + nx * nv + lane];
				bn = b_entry(nx);
				const double p = ji * wreg;
				const double dot = small ? wave_bcast(row_sum<16>(p), 0) : wave_sum(p);
				const double res = b[s] + Rr[s] * frc[s] + dot;
				const double fn = __builtin_fmin(__builtin_fmax(frc[s] - res * ARinv[s], lo[s]), hi[s]);
				const double delta = fn - frc[s];
				const double du = wave_bcast(delta, i - i0);
				if (lane == i - i0) frc[s] = fn;
				wreg += bi * du;
			}
		}
		const double cost1 = cost_of();
		// (mj_solPGS adds up the rows' cost changes, each <= 0: its improvement is never negative.  The difference of the two sums is the same number up to
		//  cancellation noise -- clamped, so that tolerance = 0 means what it means there: every sweep is run)
		const double improvement = fmax((cost - cost1) * scale, 0.0);
		cost = cost1;
		iter++;
		if (improvement < tol) break;
	}
	if (lane == 0) fi[L.solver_iter] = iter;
#pragma unroll
	for (int s = 0; s < S; s++)
		if (act[s]) f[L.efc_force + lane + 64 * s] = frc[s];
	gsync<G>();
	if (lane < nv) {
		double sacc = 0;
		for (int i = 0; i < nefc; i++) sacc += f[L.efc_J + i * nv + lane] * f[L.efc_force + i];
		f[L.qfrc_constraint + lane] = sacc;
		const double a = f[L.qacc_smooth + lane] + wreg;
		f[L.qacc + lane] = a;
		f[L.qacc_warmstart + lane] = a;
	}
	gsync<G>();
}

// ------------------------------------------------------------------------------------------------
// A13 constraint solve: warmstart + projected Gauss-Seidel (dual), one env per wavefront
// ------------------------------------------------------------------------------------------------
// MJB_PGS_PRESOLVE (round 5): nv <= 16 -- the stage ends at qfrc_constraint = J' f; the CALLER (forward_rest) then gets
// qacc = qacc_smooth + M^-1 J' f and Euler's (M + h B)^-1 (qfrc_smooth + qfrc_constraint) from ONE two-right-hand-side run of the
// DPP-row substitution on lanes 0 - 15 (solve_dense16, dual) in place of this stage's register substitution over the dense triangle
// (9.1 k cycles per step on config 3) plus Euler's own solve (~5 k).
#ifndef MJB_EXTRA_SWEEPS
#define MJB_EXTRA_SWEEPS 8
#endif
#ifndef MJB_PGS_PRESOLVE
#define MJB_PGS_PRESOLVE 1
#endif
template <int G, bool ELL, bool REGB, int TAG> STAGE void fwd_constraint_pgs(CModel m, CLayout L, CState s, const EnvLite &e)
{
	static_assert(G == 64, "the constraint solver maps rows to the 64 lanes of one wavefront");
	double *f = e.f;
	int *fi = e.fi;
	const int lane = e.lane, nv = m.nv;
	const int nefc = __builtin_amdgcn_readfirstlane(fi[L.nefc]);  // (wave-uniform by construction: make the compiler see it, so that the branches on it are scalar)
	if (nefc == 0) {
		for (int d = lane; d < nv; d += G) {
			const double a = f[L.qacc_smooth + d];
			f[L.qacc + d] = a;
			f[L.qacc_warmstart + d] = a;
			f[L.qfrc_constraint + d] = 0;
		}
		if (lane == 0) fi[L.solver_iter] = 0;
		gsync<G>();
		return;
	}
#ifdef MJB_PROFILE_SUB
	EPROF_BEGIN();
#endif
	// Row r lives in lane r together with ITS ROW OF AR = J M^-1 J' + diag(R) in registers (64 doubles): a
	// Gauss-Seidel update of row i is then "every lane proposes the update of its own row from its running residual,
	// lane i's proposal is broadcast with v_readlane, every lane does res += AR[.][i] * delta" -- no reduction and no
	// LDS access inside the sweep (the dependent chain per row is ~9 fp64 ops + one readlane pair).
	const bool rowact = lane < nefc;
	const int r = rowact ? lane : 0;
	const double *Jr = f + L.efc_J + r * nv;
	// REGB (nv <= 16): the row B_r = (M^-1 J_r')' of B = J M^-1 is solved for in this lane's registers (A7,
	// mj_projectConstraint) and never stored to LDS: AR_ri = B_r . J_i reads J, which has to be in LDS anyway.  Otherwise the
	// rows of B come from LDS (project_constraint).  Beyond 64 rows (rare; pyramidal / scalar rows only) the rows of B go to
	// the env's scratch in HBM and the AR-free path takes over.
	double x[16];
	// NR: rows the register path holds (= entries of AR per lane).  (32 for the capped kernel variant was tried: fewer spills, but
	// the 1 - 6 % of config 3's env-steps with 33 - 64 rows then take the AR-free path and throughput drops 10.3 -> 4.3 M.)
	constexpr int NR = 64;
	const bool large = !ELL && nefc > NR;
	if constexpr (REGB) MJB_REP(20) {
		double *Ld = f + L.tri;  // (compact layout: inside the region kinematics / crb / rne share -- nobody else is alive here)
		tri_build(m, L, f, Ld, lane);
		double *Bg = s.pgs_B ? s.pgs_B + (size_t)e.env * m.nefcmax * nv : nullptr;
#pragma nounroll
		for (int r0 = large ? ((nefc - 1) >> 6) << 6 : 0; r0 >= 0; r0 -= 64) {  // (block 0 last: its row stays in the registers)
			const int rb = r0 + lane;
			const bool act = rb < nefc;
			const double *Jb = f + L.efc_J + (act ? rb : 0) * nv;
#pragma unroll
			for (int k = 0; k < 16; k++) {
				const double v = Jb[k < nv ? k : 0];
				x[k] = (k < nv && act) ? v : 0.0;
			}
			tri_solve(x, Ld, f + L.qLDiagInv, nv);
			if (L.efc_B >= 0 && act) {  // (full layout only: the field mjb_get serves)
#pragma unroll
				for (int k = 0; k < 16; k++)
					if (k < nv) f[L.efc_B + rb * nv + k] = x[k];
			} else if (large && act && Bg) {
#pragma unroll
				for (int k = 0; k < 16; k++)
					if (k < nv) Bg[rb * nv + k] = x[k];
			}
		}
		if (large) {
			MJB_KEEP_BRANCH();
			__threadfence();  // the rows just written to HBM are read back through the vector cache
			fwd_constraint_pgs_large<G, TAG>(m, L, e, L.efc_B >= 0 ? f + L.efc_B : Bg);
			return;
		}
	} else if (large) {
		MJB_KEEP_BRANCH();
		fwd_constraint_pgs_large<G, TAG>(m, L, e, f + L.efc_B);
		return;
	}
	const double *Br = f + (REGB ? L.efc_J : L.efc_B) + r * nv;  // (REGB: unused)
	const bool bilateral = rowact && fi[L.efc_type + r] == MJB_CNSTR_EQUALITY;
	const double floss = (rowact && m.nfriction > 0) ? f[L.efc_frictionloss + r] : 0.0;  // > 0: dry-friction row (dof or tendon)
	const bool friction = floss > 0;
	// elliptic cones (uniform per model): a contact's rows are consecutive lanes starting at its leader row i0
	const bool ellmodel = ELL;  // (own kernel variant: the plain PGS kernel carries none of the block code)
	const int rtype = rowact ? fi[L.efc_type + r] : -1;
	const bool ell = ellmodel && rtype == MJB_CNSTR_CONTACT_ELLIPTIC;
	const int rcon = ell ? fi[L.efc_id + r] : 0;
	const int rdim = ell ? fi[L.contact_dim + rcon] : 0, ri0 = ell ? fi[L.contact_efc_address + rcon] : 0;
	double *Hc = f + L.nwt_hc;  // [36 nconmax] the contacts' diagonal blocks of AR (allocated for PGS + elliptic)
	double b = 0, Aii = 1, ARinv = 0, frc = 0;
	MJB_REP(21) {
		double jq = 0, jb = 0, jw = 0;
		if constexpr (REGB) {
#pragma unroll
			for (int k = 0; k < 16; k++) {
				const int kc = k < nv ? k : 0;
				const double j = k < nv ? Jr[kc] : 0.0;
				jq += j * f[L.qacc_smooth + kc];
				jb += j * x[k];
				jw += j * f[L.qacc_warmstart + kc];
			}
		} else {
			for (int k = 0; k < nv; k++) {
				const double j = Jr[k];
				jq += j * f[L.qacc_smooth + k];
				jb += j * Br[k];
				jw += j * f[L.qacc_warmstart + k];
			}
		}
		const double aref = f[L.efc_aref + r], R = f[L.efc_R + r];
		if (rowact) {
			b = jq - aref;
			Aii = jb + R;
			ARinv = 1.0 / Aii;
			f[L.efc_b + r] = b;
			if (!(m.disableflags & MJB_DSBL_WARMSTART)) {
				const double jar = jw - aref;
				frc = (jar < 0 || bilateral || friction) ? -(L.efc_D >= 0 ? f[L.efc_D + r] : 1.0 / R) * jar : 0.0;  // (lean frame: no efc_D)
				if (friction) frc = __builtin_fmin(__builtin_fmax(frc, -floss), floss);
				if (ellmodel) f[L.efc_force + r] = ell ? jar : frc;  // (cone rows: jar parked for the contact's leader)
			}
		}
	}
	if (ellmodel && !(m.disableflags & MJB_DSBL_WARMSTART)) {
		// warmstart forces of whole contacts from the primal cone update (mj_constraintUpdate), by the leader lanes
		MJB_KEEP_BRANCH();
		gsync<G>();
		if (ell && ri0 == r) {
			double jar6[6], D6[6], fo[6];
			for (int a = 0; a < 6; a++) {
				jar6[a] = a < rdim ? f[L.efc_force + r + a] : 0.0;
				D6[a] = a < rdim ? f[L.efc_D + r + a] : 0.0;
				fo[a] = 0;
			}
			const double *cfri = f + L.contact_friction + 5 * rcon;
			cone_force(cfri[0] / sqrt(fmax(MJB_MINVAL, m.impratio[0])), cfri, D6, rdim, jar6, fo);
			for (int a = 0; a < 6; a++)
				if (a < rdim) f[L.efc_force + r + a] = fo[a];
		}
		gsync<G>();
		if (ell) frc = f[L.efc_force + r];
	}
	// AR[i] = J_r . B_i: k outermost so that the 64 accumulators are independent (every load of one k is in flight
	// together); row groups beyond nefc accumulate stale rows and are zeroed afterwards
	double AR[NR];
	MJB_REP(22) {
#pragma unroll
	for (int i = 0; i < NR; i++) AR[i] = 0;
#ifdef MJB_PROFILE_SUB
	EPROF(19);
#endif
	if constexpr (REGB) {
		// AR_ri = B_r . J_i (AR is symmetric): B_r from the registers, row J_i as 16 LDS broadcasts at immediate offsets from
		// one address; four rows per wave-uniform guard
#pragma unroll
		for (int i = 0; i < NR; i += 4) {
			if (i < nefc) {
				MJB_KEEP_BRANCH();
				const double *J0 = f + L.efc_J + i * nv;
#pragma unroll
				for (int q = 0; q < 4; q++) {
					const double *Ji = J0 + q * nv;
					double acc0 = 0, acc1 = 0;
#pragma unroll
					for (int k = 0; k < 16; k += 2) {
						// (x[k >= nv] == 0: the clamped address keeps whatever lies behind a short row -- possibly NaN bits -- out of
						//  the sum without a vector select per element)
						const double j0 = Ji[k < nv ? k : 0], j1 = Ji[k + 1 < nv ? k + 1 : 0];
						acc0 += x[k] * j0;
						acc1 += x[k + 1] * j1;
					}
					AR[i + q] = acc0 + acc1;
				}
			}
		}
	} else {
#pragma nounroll
		for (int k = 0; k < nv; k++) {
			const double jk = Jr[k];
			const double *Bk = f + L.efc_B + k;
#pragma unroll
			for (int i = 0; i < NR; i++) {
				if ((i & 3) == 0) {
					if (i >= nefc) break;
					MJB_KEEP_BRANCH();
				}
				AR[i] += jk * Bk[i * nv];
			}
		}
	}
#pragma unroll
	for (int i = 0; i < NR; i++) AR[i] = (rowact && i < nefc) ? (lane == i ? Aii : AR[i]) : 0.0;
	}
	if constexpr (ELL) {  // the contacts' diagonal blocks, for the block updates: row r of contact c -> Hc[36 c + 6 (r - i0) + .]
		MJB_KEEP_BRANCH();
#pragma unroll
		for (int i = 0; i < 64; i++)
			if (ell && i >= ri0 && i < ri0 + rdim) Hc[36 * rcon + 6 * (r - ri0) + (i - ri0)] = AR[i];
		gsync<G>();
	}
#ifdef MJB_PROFILE_SUB
	EPROF(30);
#endif
	// residual of the warmstart forces, res = b + AR f, and their cost 0.5 f'ARf + f'b = sum_i 0.5 f_i (res_i + b_i)
	double res = b;
#pragma unroll
	for (int i = 0; i < NR; i++)
		if (i < nefc) res += AR[i] * wave_bcast(frc, i);
	{
		const double cost = wave_sum(rowact ? 0.5 * frc * (res + b) : 0.0);
		if (cost > 0 || (m.disableflags & MJB_DSBL_WARMSTART)) {
			frc = 0;
			res = b;
		}
	}
	// Gauss-Seidel sweeps; the cost decrease of a sweep (the reference sums the per-row changes, which telescope to
	// exactly this) is cost(before) - cost(after) with cost = 0.5 f'ARf + f'b = sum_i 0.5 f_i (res_i + b_i)
	const double scale = 1.0 / (MP_MEANINERTIA(m, e) * (nv > 1 ? nv : 1));
	const double tol = m.tolerance[0];
	// equality rows are two-sided, dry-friction rows live in [-frictionloss, frictionloss], the rest in [0, inf)
	const double lo = bilateral ? -__builtin_huge_val() : (friction ? -floss : 0.0);
	const double hi = friction ? floss : __builtin_huge_val();
	double cost = wave_sum(0.5 * frc * (res + b));
	int iter = 0;
	while (iter < m.iterations) {
		double dvec = 0;
		if constexpr (ELL) {
			MJB_KEEP_BRANCH();
			pgs_sweep_elliptic(AR, nefc, lane, lo, hi, ARinv, rtype, rcon, rdim, res, frc, Hc, f + L.contact_friction);
		} else if (m.nfriction > 0) {
			MJB_KEEP_BRANCH();
			pgs_sweep<true, NR>(AR, nefc, lo, hi, ARinv, res, frc, dvec);
		} else {
			MJB_KEEP_BRANCH();
			pgs_sweep<false, NR>(AR, nefc, lo, hi, ARinv, res, frc, dvec);
		}
		frc += dvec;
		const double cost1 = wave_sum(0.5 * frc * (res + b));
		// (mj_solPGS adds up the rows' cost changes, each <= 0: its improvement is never negative.  The difference of the two sums is the same number up to
		//  cancellation noise -- clamped, so that tolerance = 0 means what it means there: every sweep is run)
		const double improvement = fmax((cost - cost1) * scale, 0.0);
		cost = cost1;
		iter++;
		if (improvement < tol) break;
	}
#if MJB_DOUBLE_STAGE == 23  // (measurement builds: MJB_EXTRA_SWEEPS more Gauss-Seidel sweeps behind the stop test: the launch-time difference / that = one sweep)
	if constexpr (!ELL) {
#pragma nounroll
		for (int xs = 0; xs < MJB_EXTRA_SWEEPS; xs++) {
			double dvec = 0;
			pgs_sweep<false, NR>(AR, nefc, lo, hi, ARinv, res, frc, dvec);
			frc += dvec;
			cost = wave_sum(0.5 * frc * (res + b));
		}
	}
#endif
	if (lane == 0) fi[L.solver_iter] = iter;
	if (rowact) f[L.efc_force + r] = frc;
	gsync<G>();
#ifdef MJB_PROFILE_SUB
	EPROF(31);
	prof_rec(e.env, lane, 21, (unsigned long long)iter); prof_rec(e.env, lane, 22, (unsigned long long)nefc);
#endif
	// qfrc_constraint = J' f,  qacc = qacc_smooth + M^-1 J' f = qacc_smooth + B' f
	MJB_REP(24) if (lane < nv) {
		double sj = 0, w = 0;
		for (int i = 0; i < nefc; i++) {
			const double fr = f[L.efc_force + i];
			sj += f[L.efc_J + i * nv + lane] * fr;
			if constexpr (!REGB) w += f[L.efc_B + i * nv + lane] * fr;
		}
		f[L.qfrc_constraint + lane] = sj;
		if constexpr (!REGB) {
			const double a = f[L.qacc_smooth + lane] + w;
			f[L.qacc + lane] = a;
			f[L.qacc_warmstart + lane] = a;
		}
	}
	gsync<G>();
#ifdef MJB_PROFILE_SUB
	EPROF(20);
#endif
	if constexpr (REGB && !MJB_PGS_PRESOLVE) {
		// M^-1 (J' f) by the same dense substitution, the whole nv-vector in the registers of every lane (wave-uniform
		// addresses: one LDS broadcast per entry; lane k keeps element k)
		double w[16];
#pragma unroll
		for (int k = 0; k < 16; k++) {
			const double v = f[L.qfrc_constraint + (k < nv ? k : 0)];
			w[k] = k < nv ? v : 0.0;
		}
		tri_solve(w, f + L.tri, f + L.qLDiagInv, nv);
		double wk = 0;
#pragma unroll
		for (int k = 0; k < 16; k++) wk = lane == k ? w[k] : wk;
		if (lane < nv) {
			const double a = f[L.qacc_smooth + lane] + wk;
			f[L.qacc + lane] = a;
			f[L.qacc_warmstart + lane] = a;
		}
		gsync<G>();
	}
#ifdef MJB_PROFILE_SUB
	EPROF(23);
#endif
}

// the LDS-B variants (nv > 16) stay out of line: models that small never pay their registers or instruction-cache lines
template <int G, bool ELL, int TAG> __device__ __attribute__((noinline)) void fwd_constraint_pgs_ldsB(CModel m, CLayout L, CState s, const EnvLite e)
{
	fwd_constraint_pgs<G, ELL, false, TAG>(m, L, s, e);
}

// packed lower triangle of the nv <= 32 Hessian (FrameLayout::nwt_H holds nv (nv + 1) / 2 + 1 doubles then, at least the 272 of chol_schur16's scratch)
#define MJB_HPACK(r, c) ((r) * ((r) + 1) / 2 + (c))

// ------------------------------------------------------------------------------------------------
// A14 Newton solver (primal), one env per wavefront.
//   minimise over a:  0.5 (a - a0)' M (a - a0) + sum_i s_i(J_i a - aref_i),  s_i(x) = 0.5 D_i x^2 for x < 0
// Rows live in lanes (jaref, jv, D in registers), dof vectors in LDS; the nv x nv Hessian
// H = M + J' diag(D[active]) J is rebuilt each iteration (entry per lane), factorised by a column Cholesky
// whose pivot is exchanged with v_readlane, and the exact line search evaluates the piecewise-quadratic cost
// with three wave-wide sums per trial point.
// ------------------------------------------------------------------------------------------------
struct LsPoint {
	double alpha, cost, d0, d1;
};

// Elliptic cone of one contact in the scaled space U_0 = mu x_0, U_j = friction_j x_j (x = jar of its rows),
// N = U_0, T = |U_1..|; with R_j = R_0 mu^2 / friction_j^2 the primal cost is (oracle/mjo_constraint.c):
//   top (N >= mu T): 0 | bottom (mu N + T <= 0): 0.5 sum D_j x_j^2 | middle: 0.5 Dm (N - mu T)^2, Dm = D_0 / (mu^2 (1 + mu^2))
struct ConeLine {  // per-contact constants of the line search, held by the contact's leader lane
	double N0, N1, TT, UV, VV, q0b, q1b, q2b, mu, Dm;
};

// one lane's share of the line-search cost and its first two derivatives at step `a`
// 1 / sqrt(x) for a positive NORMAL x (the pivots are clamped to mjMINVAL): the hardware seed and the refinement of the library's
// rsqrt, term for term -- same bits -- without its select on the operand's class (zero / infinity / NaN), three dependent
// instructions on a chain the factorisation walks 32 times
DEVI double rsqrt_pos(double x)
{
	const double y0 = __builtin_amdgcn_rsq(x);
	const double e = fma(y0 * -x, y0, 1.0);
	return fma(y0 * e, fma(e, 0.375, 0.5), y0);
}
// One row's (scalar rows) or one contact's (cone leaders) share of the line-search cost and its two derivatives at step a.
// Straight-line: every lane evaluates the scalar form and the cone form and the row's kind picks by selects -- as nested branches the
// kinds of a wave ran one after the other (scalar rows, then the leaders' chain of a square root and a division, ~1.2 k cycles per
// trial point); |T| and 1 / |T| come from ONE reciprocal square root (the oracle takes sqrt, then divides: equal to rounding).
DEVI void ls_row(double a, bool scalar_row, bool bilateral, bool leader, double jaref, double jv, double D, double fl,
                 const ConeLine &cl, double &c0, double &c1, double &c2)
{
	// scalar rows: quadratic when active (x < 0, or an equality), Huber for dry friction (quadratic for |x| < R fl, slope -+fl outside)
	const double x = jaref + a * jv;
	const bool fric = fl > 0;
	const double rf = fric ? fl / D : 0.0;
	const bool quad = scalar_row && (fric ? (x > -rf && x < rf) : (x < 0 || bilateral));
	const bool lin = scalar_row && fric && !quad;
	const double Dx = D * x;
	const double q0 = 0.5 * Dx * x, q1 = Dx * jv, q2 = D * jv * jv;
	const double l0 = fl * (-0.5 * rf + fabs(x)), l1 = (x < 0 ? -fl : fl) * jv;
	// elliptic cone of a contact (evaluated by its first row's lane)
	const double N = cl.N0 + a * cl.N1;
	const double T2r = cl.TT + a * (2 * cl.UV + a * cl.VV);
	const bool tpos = T2r > 1e-290;
	const double iT = tpos ? rsqrt_pos(tpos ? T2r : 1.0) : 0.0, T = tpos ? T2r * iT : 0.0;
	const bool top = N >= cl.mu * T, bottom = !top && cl.mu * N + T <= 0;
	const bool mid = leader && !top && !bottom, bot = leader && bottom;
	const double b0 = cl.q0b + a * (cl.q1b + a * cl.q2b), b1 = cl.q1b + 2 * a * cl.q2b, b2 = 2 * cl.q2b;
	const double NmT = N - cl.mu * T, T1 = (cl.UV + a * cl.VV) * iT, T2d = (cl.VV - T1 * T1) * iT;
	const double s1 = cl.N1 - cl.mu * T1, DN = cl.Dm * NmT;
	const double m0 = 0.5 * DN * NmT, m1 = DN * s1, m2 = cl.Dm * (s1 * s1 - NmT * cl.mu * T2d);
	c0 += quad ? q0 : (lin ? l0 : (bot ? b0 : (mid ? m0 : 0.0)));
	c1 += quad ? q1 : (lin ? l1 : (bot ? b1 : (mid ? m1 : 0.0)));
	c2 += quad ? q2 : (bot ? b2 : (mid ? m2 : 0.0));
}

typedef double mjb_d4 __attribute__((ext_vector_type(4)));

// columns J0 .. J0+15 of the right-looking Cholesky with one row of H per lane in registers (see fwd_constraint_newton).
// A column's update stays INSIDE the 16-column block: the first block's contribution to the second (rows / columns 16 .. 31) is
// one 16 x 16 x 16 product, chol_schur16 below.  Round 3 ran all 32 columns right-looking: every entry l_cj a lane multiplies into
// its row arrives by a v_readlane pair, 1444 v_readlane against 633 v_fma in the stage, issue bound at ~6 cycles each.
// (`lid`: the self-laundering lane index, read afresh for each column's `lane == j`: as plain loop invariants of the Newton
//  iteration the 32 + 96 lane masks of the factorisation and the two substitutions were hoisted out of it, spilled -- SGPR pairs
//  in lanes of a VGPR that was itself parked in an AGPR -- and fetched back at every use: s_or_saveexec, v_accvgpr_read, two
//  v_readlane, s_nop, where three VALU instructions rebuild the mask)
template <int J0> DEVI void chol_cols16(double (&Hr)[32], const int nv, const LaneId lid, double &myrinv)
{
	const int lane = lid;  // (once per block of columns)
#pragma unroll
	for (int j = J0; j < J0 + 16; j++) {
		// (no guards on nv anywhere in the nest: rows / columns >= nv are zero, so their steps are no-ops, and one straight-line
		//  block lets the scheduler run a column's rsqrt chain under the previous column's trailing update -- config 5: +7 %)
		double sj = wave_bcast(Hr[j], j);
		if (sj < MJB_MINVAL) sj = MJB_MINVAL;
		const double rinv = rsqrt_pos(sj);
		const double lkj = (lane == j) ? sj * rinv : Hr[j] * rinv;
		Hr[j] = lkj;
		if (lane == j) myrinv = rinv;
		// groups of four columns (constant loop bounds -- the nest only unrolls fully that way; entries c >= nv of a group belong to
		// idle lanes: l_cj == 0)
#pragma unroll
		for (int c0 = J0; c0 < J0 + 16; c0 += 4) {
			if (c0 + 3 <= j) continue;
			double l4[4];
#pragma unroll
			for (int q = 0; q < 4; q++) l4[q] = wave_bcast(lkj, c0 + q);
#pragma unroll
			for (int q = 0; q < 4; q++)
				if (c0 + q > j) Hr[c0 + q] -= lkj * l4[q];
		}
	}
}

// H22 -= L21 L21' on the matrix cores (16 < nv <= 32; rows 16 .. 31 sit in lanes 16 .. 31, L21 = their registers 0 .. 15 once the
// first 16 columns are done): the lanes publish L21 column-major in `S` (272 doubles of scratch: the Hessian's LDS copy, dead while
// its rows are in registers), lane l feeds A[l & 15][l >> 4] = B[l >> 4][l & 15] = L21[l & 15][4 s + (l >> 4)] to four
// v_mfma_f64_16x16x4_f64, the product goes back through S (row stride 17: every lane its own bank) and each of the lanes 16 .. 31
// subtracts its row.  Replaces 376 fma + 752 v_readlane per lane by ~60 instructions and two LDS round trips; the entries of H22
// now receive the block's sixteen products as one sum instead of sixteen subtractions (rounding only).
template <typename SYNC> DEVI void chol_schur16(double (&Hr)[32], const int lane, double *S, SYNC &&sync)
{
	const bool mine = lane >= 16 && lane < 32;
	const int i = mine ? lane - 16 : 0;
	if (mine) {
#pragma unroll
		for (int c = 0; c < 16; c++) S[c * 16 + i] = Hr[c];
	}
	sync();
	const int li = lane & 15, lk = lane >> 4;
	double v[4];
#pragma unroll
	for (int q = 0; q < 4; q++) v[q] = S[(4 * q + lk) * 16 + li];
	mjb_d4 acc = mjb_d4{ 0, 0, 0, 0 };
#pragma unroll
	for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v[q], v[q], acc, 0, 0, 0);
	sync();
#pragma unroll
	for (int q = 0; q < 4; q++) S[(lk + 4 * q) * 17 + li] = acc[q];
	sync();
	if (mine) {
		double p[16];
#pragma unroll
		for (int c = 0; c < 16; c++) p[c] = S[i * 17 + c];
#pragma unroll
		for (int c = 0; c < 16; c++) Hr[16 + c] -= p[c];
	}
}

// R = rows per lane: row r lives in lane r % 64, slot r / 64 (nefcmax <= 64 R).  R == 1 is BASELINE config 3,
// R == 4 covers the ~200 rows of config 5.
// JG: efc_J is read from the env's block in HBM (Jg) instead of the frame (kernel variant 4, env-steps with more rows than the
// frame's share of efc_J; see make_constraint)
template <int G, int R, bool CGS = false, bool JG = false>  // CGS: conjugate gradient (no Hessian; Polak-Ribiere directions preconditioned by M^-1)
STAGE void fwd_constraint_newton(CModel m, CLayout L, const EnvLite &e, double *gbase = nullptr)
{
	static_assert(G == 64, "the Newton solver maps rows / Hessian columns to the 64 lanes of one wavefront");
	double *f = e.f;
	int *fi = e.fi;
	const LaneId lane = e.lane;  // (re-derived at every use: nothing computed from it is hoisted out of the iteration and spilled)
	const int nv = m.nv;
	const int nefc = __builtin_amdgcn_readfirstlane(fi[L.nefc]);
	// JG: J -- and, when the frame's other row arrays are capped too (L.rcap < nefcmax), every per-row array, the row metadata and
	// the cone blocks -- come from the env's block in HBM (mjb_dev.h, RowBlock); the branches fold away in the other instantiations
	[[maybe_unused]] RowBlock gb{};
	if constexpr (JG) gb = mjb_rowblock(gbase, m.nefcmax, m.nv, m.nconmax, L.hcs);
	const bool gall = JG && L.rcap < m.nefcmax;
	const double *const Jb = JG ? gb.J : f + L.efc_J;
	const double *const Dp = gall ? gb.D : f + L.efc_D;
	const double *const arefp = gall ? gb.aref : f + L.efc_aref;
	const double *const flp = gall ? gb.fl : f + L.efc_frictionloss;
	double *const forcep = gall ? gb.force : f + L.efc_force;
	const int *const typep = gall ? gb.type : fi + L.efc_type;
	const int *const idp = gall ? gb.id : fi + L.efc_id;
	int *const metap = gall ? gb.meta : fi + L.iscratch;
	const int rstride = gall ? m.nefcmax : L.rcap;   // jaref | jv | hw are rstride doubles apart
	const bool hcrow = !gall && L.hcrow;             // cone block of a contact: at hcd * (its first row), or at hcs * contact
	[[maybe_unused]] auto sync = [&]() {
		if constexpr (JG) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");  // (row data crosses lanes through HBM)
		gsync<G>();
	};
	if (nefc == 0) {
		for (int d = lane; d < nv; d += G) {
			const double a = f[L.qacc_smooth + d];
			f[L.qacc + d] = a;
			f[L.qacc_warmstart + d] = a;
			f[L.qfrc_constraint + d] = 0;
		}
		if (lane == 0) fi[L.solver_iter] = 0;
		gsync<G>();
		return;
	}
	EPROF_BEGIN();
	double *Md = f + L.nwt_M, *H = f + L.nwt_H, *Hc = gall ? gb.hc : f + L.nwt_hc;
	const int hcs = L.hcs, hcd = L.hcd;  // cone blocks: hcd x hcd (hcd = the model's largest contact dim), hcs doubles apart
	double *qa = f + L.nwt_vec, *grad = qa + 2 * nv, *srch = grad + nv;  // (qa + nv: M qacc in the frame's layout -- carried in registers)
	double *jar_s = gall ? gb.nwt_row : f + L.nwt_row, *jv_s = jar_s + rstride, *hw = jv_s + rstride;  // per-row jaref, jv, Hessian weight
	auto hcb = [&](int con, int row) -> double * { return Hc + (hcrow ? hcd * row : hcs * con); };
	const bool dofact = lane < nv;
	const int k = dofact ? lane : 0;
	const double tol = m.tolerance[0];
	const double scale = 1.0 / (MP_MEANINERTIA(m, e) * (nv > 1 ? nv : 1));
	// The dense symmetric M.  nv <= 32: lane k keeps ROW k in 32 registers, gathered once from the qM layout through the host table
	// M_sym (entries the tree leaves zero, and columns >= nv, are +0: products with them leave every sum bit-unchanged) -- the
	// frame holds no dense copy (900 doubles on config 5) and the three M x products of an iteration read no LDS for M.
	// nv > 32: a dense copy in the frame (entry per lane).
	const bool mreg = nv <= 32;
	double Mrow[32];
	if (mreg) {
		MJB_KEEP_BRANCH();
		int ms[32];
		msym_row(m, k, ms);
#pragma unroll
		for (int c = 0; c < 32; c++) {
			const int adr = ms[c];
			const double v = f[L.qM + (adr >= 0 ? adr : 0)];
			Mrow[c] = (dofact && adr >= 0) ? v : 0.0;
		}
	} else {
		MJB_KEEP_BRANCH();
#pragma unroll
		for (int c = 0; c < 32; c++) Mrow[c] = 0.0;
		for (int t = lane; t < nv * nv; t += G) Md[t] = 0;
		sync();
		for (int en = lane; en < m.nM; en += G) {
			const int i = m.M_rowdof[en], j = m.M_coldof[en];
			const double v = f[L.qM + en];
			Md[i * nv + j] = v;
			Md[j * nv + i] = v;
		}
		sync();
	}
	// (M x)_k for the lane's dof k (x: nv doubles in LDS); same summation order on both paths
	auto m_dot = [&](const double *x) -> double {
		double t = 0;
		if (mreg) {
			MJB_KEEP_BRANCH();
#pragma unroll
			for (int c = 0; c < 32; c++) t += Mrow[c] * x[c < nv ? c : 0];
		} else if (dofact) {
#pragma unroll 5
			for (int c = 0; c < nv; c++) t += Md[k * nv + c] * x[c];
		}
		return t;
	};

	// row kind: scalar (limit / frictionless / pyramidal), cone leader (first row of an elliptic contact), cone member
	bool rowact[R], scalar_row[R], leader[R], bilat[R];
	int rr[R], cdim[R], rcon[R];
	double D[R], aref[R], cmu[R], cdmi[R], fl[R];  // fl: force limit of a dry-friction row, 0 for every other row
#pragma unroll
	for (int i = 0; i < R; i++) {
		const int r = lane + 64 * i;
		rowact[i] = r < nefc;
		rr[i] = rowact[i] ? r : 0;
		const int rtype = rowact[i] ? typep[r] : 0;
		rcon[i] = rowact[i] ? idp[r] : 0;
		const bool is_cone = rowact[i] && rtype == MJB_CNSTR_CONTACT_ELLIPTIC;
		scalar_row[i] = rowact[i] && !is_cone;
		bilat[i] = rowact[i] && rtype == MJB_CNSTR_EQUALITY;
		leader[i] = is_cone && fi[L.contact_efc_address + rcon[i]] == r;
		cdim[i] = is_cone ? fi[L.contact_dim + rcon[i]] : 0;
		D[i] = rowact[i] ? Dp[r] : 0.0;
		fl[i] = (rowact[i] && m.nfriction > 0) ? flp[r] : 0.0;
		aref[i] = rowact[i] ? arefp[r] : 0.0;
		cmu[i] = leader[i] ? f[L.contact_friction + 5 * rcon[i]] / sqrt(fmax(MJB_MINVAL, m.impratio[0])) : 1.0;
		cdmi[i] = 1.0 / (cmu[i] * cmu[i] * (1 + cmu[i] * cmu[i]));  // Dm = D_0 / (mu^2 (1 + mu^2)): the divisor once per step, not per update
		// what the Hessian build needs to know about the row, in ONE int (read by whichever lane feeds the row to the matrix
		// cores): -1 = scalar row (weight hw[r]), else first row of its cone | dim << 8 | contact << 12
		if (rowact[i]) metap[r] = is_cone ? (fi[L.contact_efc_address + rcon[i]] | (cdim[i] << 8) | (rcon[i] << 12)) : -1;
	}


	// J_r . x - aref_r for the lane's rows (x: nv doubles in LDS)
	auto row_dots = [&](const double *x, double sub, double *out) {
#pragma unroll
		for (int i = 0; i < R; i++) {
			out[i] = 0;
			if (64 * i >= nefc) continue;  // wave-uniform: no row of this slot exists
			const double *Jr = Jb + rr[i] * nv;
			double s = -sub * aref[i];
#pragma unroll 5
			for (int c = 0; c < nv; c++) s += Jr[c] * x[c];
			out[i] = s;
		}
	};

	// nv <= 32: (M x)_k and the lane's J_r . x - sub aref_r in one go -- x crosses from LDS to registers once (broadcast reads shared by
	// both products), every load of a row of J is in flight at once (clamped columns times an x that is zero beyond nv: exact no-ops)
	// where the counted loop fetched five at a time; same summation order as m_dot / row_dots
	// (one row per lane only: with more the 32 extra registers of x cost the 2- / 4-row kernels spills -- config 3 under Newton -3.5 %)
	auto dots = [&](const double *x, double sub, double &mx, double *out) {
		if (R == 1 && mreg) {
			MJB_KEEP_BRANCH();
			// (eight columns to a chunk, a REAL wave-uniform branch per chunk: a chunk inside [0, nv) loads at constant offsets, all
			//  eight in flight; only the chunk that straddles nv clamps and zeroes.  Written as one `c < nv ? .. : ..` per column the
			//  compiler turned the uniform conditions into 25 branches around single loads, each waited for on its own: 101 s_waitcnt
			//  for 45 loads, 4.2 k cycles per call -- five to six calls a step on config 5)
			double xz[32];
			static_for<4>([&](auto cc) {
				constexpr int c0 = 8 * decltype(cc)::value;
				if (c0 + 8 <= nv) {
					MJB_KEEP_BRANCH();
#pragma unroll
					for (int q = 0; q < 8; q++) xz[c0 + q] = x[c0 + q];
				} else {
					MJB_KEEP_BRANCH();
#pragma unroll
					for (int q = 0; q < 8; q++) {
						const int cl = c0 + q < nv ? c0 + q : nv - 1;
						const double v = x[cl];
						xz[c0 + q] = c0 + q < nv ? v : 0.0;
					}
				}
			});
			double t = 0;
#pragma unroll
			for (int c = 0; c < 32; c++) t += Mrow[c] * xz[c];
			mx = t;
#pragma unroll
			for (int i = 0; i < R; i++) {
				out[i] = 0;
				if (64 * i >= nefc) continue;  // wave-uniform: no row of this slot exists
				const double *Jr = Jb + rr[i] * nv;
				double s = -sub * aref[i];
				static_for<4>([&](auto cc) {
					constexpr int c0 = 8 * decltype(cc)::value;
					if (c0 + 8 <= nv) {
						MJB_KEEP_BRANCH();
						double j8[8];
#pragma unroll
						for (int q = 0; q < 8; q++) j8[q] = Jr[c0 + q];
#pragma unroll
						for (int q = 0; q < 8; q++) s += j8[q] * xz[c0 + q];
					} else {
						MJB_KEEP_BRANCH();
						double j8[8];
#pragma unroll
						for (int q = 0; q < 8; q++) j8[q] = Jr[c0 + q < nv ? c0 + q : nv - 1];
#pragma unroll
						for (int q = 0; q < 8; q++) s += j8[q] * xz[c0 + q];  // (xz is zero beyond nv: the clamped entries drop out exactly)
					}
				});
				out[i] = s;
			}
		} else {
			MJB_KEEP_BRANCH();
			mx = m_dot(x);
			row_dots(x, sub, out);
		}
	};

	// constraint update at the jaref values parked in `jar_s` (the argument: jar_s, or jv_s for the second warmstart candidate): returns this lane's cost share; forces (and, when
	// `hess`, the Hessian weights hw / cone blocks Hc) written to LDS
	// (cones: straight-line code in the contact's leader lane -- every load of the contact's rows in flight at once through clamped
	//  indices, the three zones by selects: the leaders of a wave sit in different zones anyway and ran them one after the other,
	//  with a branch and an LDS round trip per `j < dim` test; DMAX = the model's largest contact dimension rounded up to 4 or 6)
	auto cone_update_d = [&](auto DIMC, bool hess, const double *jar_s) -> double {
		constexpr int DMAX = decltype(DIMC)::value;
		double cost = 0;
#pragma unroll
		for (int i = 0; i < R; i++) {
			const int r = rr[i];
			if (64 * i >= nefc) continue;
			if (scalar_row[i] && fl[i] > 0) {
				const double x = jar_s[r], rf = fl[i] / D[i];
				const bool quad = x > -rf && x < rf;
				forcep[r] = quad ? -D[i] * x : (x < 0 ? fl[i] : -fl[i]);
				if (hess) hw[r] = quad ? D[i] : 0.0;
				cost += quad ? 0.5 * D[i] * x * x : fl[i] * (-0.5 * rf + fabs(x));
			} else if (scalar_row[i]) {
				const double x = jar_s[r];
				const bool act = x < 0 || bilat[i];
				forcep[r] = act ? -D[i] * x : 0.0;
				if (hess) hw[r] = act ? D[i] : 0.0;
				cost += act ? 0.5 * D[i] * x * x : 0.0;
			} else if (leader[i]) {
				const int dim = cdim[i];
				const double mu = cmu[i];
				const double *cfri = f + L.contact_friction + 5 * rcon[i];
				double U[DMAX], x[DMAX], Dj[DMAX], fr[DMAX], TT = 0;
#pragma unroll
				for (int j = 0; j < DMAX; j++) {
					const int jj = j < dim ? j : 0;
					x[j] = jar_s[r + jj];
					Dj[j] = Dp[r + jj];
					fr[j] = cfri[(j > 0 && j < dim) ? j - 1 : 0];
				}
#pragma unroll
				for (int j = 0; j < DMAX; j++) {
					const bool in = j < dim;
					x[j] = in ? x[j] : 0.0;
					Dj[j] = in ? Dj[j] : 0.0;
					fr[j] = j == 0 ? mu : fr[j];
					U[j] = fr[j] * x[j];
					if (j > 0) TT += U[j] * U[j];
				}
				// (|T| and 1 / |T| from one reciprocal square root -- the oracle takes sqrt, then divides: equal to rounding; Dm from the
				//  per-contact constant the row setup keeps)
				const bool tpos = TT > 1e-290;
				const double iTr = tpos ? rsqrt_pos(tpos ? TT : 1.0) : 0.0;
				const double N = U[0], T = tpos ? TT * iTr : 0.0;
				const bool top = N >= mu * T, bottom = !top && mu * N + T <= 0, middle = !top && !bottom;
				// (middle zone => T > 0; the other zones compute it on T = 1 and discard it)
				const double iT = middle ? iTr : 1.0;
				const double Dm = Dj[0] * cdmi[i], NmT = N - mu * T, iT3 = iT * iT * iT;
				const double f0 = -Dm * NmT * mu;
				double g[DMAX];
				g[0] = mu;
#pragma unroll
				for (int j = 1; j < DMAX; j++) g[j] = j < dim ? -mu * fr[j] * U[j] * iT : 0.0;
#pragma unroll
				for (int j = 0; j < DMAX; j++) {
					cost += (bottom && j < dim) ? 0.5 * Dj[j] * x[j] * x[j] : 0.0;
					const double fm = j == 0 ? f0 : -f0 * iT * U[j] * fr[j];
					const double fj = top ? 0.0 : (bottom ? -Dj[j] * x[j] : fm);
					if (j < dim) forcep[r + j] = fj;
				}
				cost += middle ? 0.5 * Dm * NmT * NmT : 0.0;
				if (hess) {
					double *hc = hcb(rcon[i], r);
					// (with hcrow the NEXT contact's block starts hcd * dim doubles further on: a contact of dimension 3 among contacts of dimension 4 owns
					//  three block rows, not hcd.  The block rows past its dimension -- which nobody reads -- therefore write row 0 once more, with row 0's
					//  values, instead of zeros into the neighbour's first row; a select on the address and the operands, no per-lane branch.  Round 5: the
					//  zeros went to the neighbour, tools/replay_reset.py found the env-step of the power grasp at which that made the Hessian indefinite.)
#pragma unroll
					for (int j = 0; j < DMAX; j++) {
						const bool rowin = j < 3 || j < dim;  // (an elliptic cone has 3, 4 or 6 rows: only the block rows from the fourth on can lie past it)
						const double gj = rowin ? g[j] : g[0];
#pragma unroll
						for (int c2 = 0; c2 < DMAX; c2++) {
							if (j >= hcd || c2 >= hcd) continue;  // (wave-uniform: the block is hcd x hcd)
							double vm = Dm * gj * g[c2];
							if (j >= 1 && c2 >= 1) vm += rowin ? -Dm * NmT * mu * fr[j] * fr[c2] * ((j == c2 ? iT : 0.0) - U[j] * U[c2] * iT3) : 0.0;
							const double vb = rowin ? (j == c2 ? Dj[j] : 0.0) : (c2 == 0 ? Dj[0] : 0.0);
							const bool in = c2 < dim;
							hc[(rowin ? j : 0) * hcd + c2] = (in && !top) ? (bottom ? vb : vm) : 0.0;
						}
					}
#pragma unroll
					for (int j = 0; j < DMAX; j++)
						if (j < dim) hw[r + j] = 0;
				}
			}
		}
		return cost;
	};
	auto cone_update = [&](bool hess, const double *src) -> double {
		if (hcd <= 4) {
			MJB_KEEP_BRANCH();
			return cone_update_d(std::integral_constant<int, 4>{}, hess, src);
		}
		MJB_KEEP_BRANCH();
		return cone_update_d(std::integral_constant<int, 6>{}, hess, src);
	};

	EPROF(24);
	// warmstart: the cheaper of qacc_warmstart and qacc_smooth.  Both candidates' M q and J q - aref are formed back to back (the
	// second one parked in jv_s, free until the line search) and the winner's are KEPT: mj_solPrimal computes Ma and jaref once and
	// from then on moves them along the search direction (Ma += alpha Mv, jaref += alpha jv) -- round 3 multiplied again at the top
	// of every iteration (3.9 k of the 10 k cycles of its gradient step on config 5).
	double ma, jaref[R];
	{
		const double *qw = f + L.qacc_warmstart, *qs = f + L.qacc_smooth;
		double t0, x0[R], t1, x1[R];
		dots(qw, 1.0, t0, x0);
		dots(qs, 1.0, t1, x1);
		const double gk0 = dofact ? 0.5 * (t0 - f[L.qfrc_smooth + k]) * (qw[k] - qs[k]) : 0.0;
		const double gk1 = dofact ? 0.5 * (t1 - f[L.qfrc_smooth + k]) * (qs[k] - qs[k]) : 0.0;
#pragma unroll
		for (int i = 0; i < R; i++)
			if (rowact[i]) {
				jar_s[rr[i]] = x0[i];
				jv_s[rr[i]] = x1[i];
			}
		sync();
		const double ck0 = cone_update(false, jar_s);
		const double ck1 = cone_update(false, jv_s);
		double sg0 = gk0, sc0 = ck0, sg1 = gk1, sc1 = ck1, sz = 0;
		wave_sum3(sg0, sc0, sg1);
		wave_sum3(sc1, sz, sz);
		const double cost0 = sg0 + sc0, cost1 = sg1 + sc1;
		const double best = (m.disableflags & MJB_DSBL_WARMSTART) ? 1e300 : cost0;
		const bool smooth = cost1 < best;  // (wave-uniform)
		ma = smooth ? t1 : t0;
#pragma unroll
		for (int i = 0; i < R; i++) jaref[i] = smooth ? x1[i] : x0[i];
		if (dofact) qa[k] = smooth ? qs[k] : qw[k];
		sync();
	}

	EPROF(25);
	double cost = 0, prev_cost = 0;
	double cg_gold = 0, cg_mgold = 0, cg_sold = 0;  // CG: this lane's element of the previous gradient, M^-1 gradient and search
	int iter = 0;
	for (;;) {
		EPROF(30);
		// forces, cost, gradient at the current Ma = M qacc, jaref = J qacc - aref (carried in registers, see the warmstart)
		const double gk = dofact ? 0.5 * (ma - f[L.qfrc_smooth + k]) * (qa[k] - f[L.qacc_smooth + k]) : 0.0;
#pragma unroll
		for (int i = 0; i < R; i++)
			if (rowact[i]) jar_s[rr[i]] = jaref[i];
		sync();
#if defined(MJB_PROFILE_NWT) && !defined(MJB_PROFILE_LS)
		EPROF(20);
#endif
		const double ck = cone_update(true, jar_s);
#if defined(MJB_PROFILE_NWT) && !defined(MJB_PROFILE_LS)
		EPROF(21);
#endif
		double gauss = gk, csum = ck, cz = 0;
		wave_sum3(gauss, csum, cz);
		prev_cost = cost;
		cost = gauss + csum;
		sync();
#if defined(MJB_PROFILE_NWT) && !defined(MJB_PROFILE_LS)
		EPROF(22);
#endif
		double gr = 0;
		if (dofact) {
			// (blocks of eight rows with every load of a block in flight; rows past nefc: row 0 times a zero force, an exact no-op)
			double s = 0;
#pragma nounroll
			for (int i0 = 0; i0 < nefc; i0 += 8) {
				double jv8[8], fv8[8];
				if (i0 + 8 <= nefc) {  // (a real branch per block: see dots)
					MJB_KEEP_BRANCH();
#pragma unroll
					for (int q = 0; q < 8; q++) {
						jv8[q] = Jb[(i0 + q) * nv + k];
						fv8[q] = forcep[i0 + q];
					}
				} else {
					MJB_KEEP_BRANCH();
#pragma unroll
					for (int q = 0; q < 8; q++) {
						const int ii = i0 + q < nefc ? i0 + q : 0;
						jv8[q] = Jb[ii * nv + k];
						const double fv = forcep[ii];
						fv8[q] = i0 + q < nefc ? fv : 0.0;
					}
				}
#pragma unroll
				for (int q = 0; q < 8; q++) s += jv8[q] * fv8[q];
			}
			f[L.qfrc_constraint + k] = s;
			gr = ma - f[L.qfrc_smooth + k] - s;
			grad[k] = gr;
		}
		if (iter > 0) {
			const double improvement = scale * (prev_cost - cost);
			const double gnorm = scale * sqrt(wave_sum(gr * gr));
			if (improvement < tol || gnorm < tol || iter >= m.iterations) break;
		}
#if defined(MJB_PROFILE_NWT) && !defined(MJB_PROFILE_LS)
		EPROF(23);
#endif
		EPROF(26);
		double x = 0;
		if constexpr (CGS) {
			// mj_solPrimal without the Hessian: Mgrad = M^-1 grad (sparse L'DL solve in LDS), beta = grad.(Mgrad - Mgrad_old) /
			// max(MINVAL, grad_old.Mgrad_old) clipped at 0, search = -Mgrad + beta search_old
			double *mg = srch + nv;
			if (dofact) mg[k] = gr;
			sync();
			solve<G>(m, e, mg, f + L.qLD, f + L.qLDiagInv);
			const double mgk = dofact ? mg[k] : 0.0;
			double beta = 0;
			if (iter > 0) {
				const double num = wave_sum(gr * (mgk - cg_mgold)), den = wave_sum(cg_gold * cg_mgold);
				beta = num / fmax(MJB_MINVAL, den);
				if (beta < 0) beta = 0;
			}
			x = mgk - beta * cg_sold;  // (the search direction is -x)
			cg_gold = gr;
			cg_mgold = mgk;
		} else {
		// H = M + J' W J on the matrix cores: 16x16 tiles of v_mfma_f64_16x16x4_f64 over 4-row slabs of J.
		// A[i][kk] = (W J)[r0+kk][a0+i] (row weight, or the contact's cone block times its rows), B[kk][j] = J[r0+kk][b0+j];
		// lane l feeds A[l&15][l>>4], B[l>>4][l&15] and receives D[(l>>4) + 4 q][l&15], q = 0..3.
		{
			const int li = lane & 15, lk = lane >> 4;
			const int ntile = (nv + 15) >> 4;
			// One pass over the 4-row slabs per tile ROW ta: the weighted operand A = (W J)[slab][16 ta + .] is formed once and
			// multiplied with the column blocks tb <= ta (one accumulator each); the next slab's operands are fetched before
			// the current slab's MFMAs issue, so the dependent LDS reads (row metadata -> cone block / J) overlap them.
			// (straight-line: rows past nefc and columns past nv read clamped addresses and are zeroed by selects -- with branches
			//  around the loads the scheduler cannot run a slab's operand fetch under the previous slab's MFMAs)
			auto operands = [&](int r0, int ca, bool ina, double &av, double (&bv)[4], int ta) {
				const int r = r0 + lk;
				const bool live = r < nefc;
				const int rc = live ? r : 0;
				const double *Jr = Jb + rc * nv;
#pragma unroll
				for (int tb = 0; tb < 4; tb++) {
					const int col = 16 * tb + li;
					const double v = Jr[col < nv ? col : 0];
					bv[tb] = (live && tb <= ta && col < nv) ? v : 0.0;
				}
				// a scalar row is a 1 x 1 "block" whose weight sits in hw[r]; all loads of the slab leave together once the
				// row's metadata int has arrived
				const int meta = metap[rc];
				const bool cone = meta >= 0;
				const int adr = cone ? (meta & 255) : rc, dim = cone ? ((meta >> 8) & 15) : 1, con = cone ? (meta >> 12) : 0;
				const double *wp = cone ? hcb(con, adr) + hcd * (rc - adr) : hw + rc;
				const double *Jc = Jb + adr * nv + (ina ? ca : 0);
				double wv[6], jv6[6];
#pragma unroll
				for (int s2 = 0; s2 < 6; s2++) {
					const int sc = s2 < dim ? s2 : 0;
					wv[s2] = wp[sc];
					jv6[s2] = Jc[sc * nv];
				}
				double a = 0;
#pragma unroll
				for (int s2 = 0; s2 < 6; s2++) a += s2 < dim ? wv[s2] * jv6[s2] : 0.0;
				av = (live && ina) ? a : 0.0;
			};
			// nv <= 32 (two tile rows): ONE pass over the slabs feeds all three tiles of the lower triangle -- the slab's rows of J,
			// its metadata and its weights are fetched once instead of once per tile row (21 LDS reads per slab instead of 34;
			// cone blocks no larger than 4 x 4, config 5: 15), and the three MFMAs of a slab issue back to back.
			auto one_pass = [&](auto DIMC) {
				constexpr int DMAX = decltype(DIMC)::value;
				const int c1 = 16 + li;
				const bool in1 = c1 < nv, in0 = li < nv;
				auto fetch = [&](int r0, double &a0, double &a1, double &b0, double &b1) {
					const int r = r0 + lk;
					const bool live = r < nefc;
					const int rc = live ? r : 0;
					const double *Jr = Jb + rc * nv;
					const double v0 = Jr[in0 ? li : 0], v1 = Jr[in1 ? c1 : 0];
					b0 = (live && in0) ? v0 : 0.0;
					b1 = (live && in1) ? v1 : 0.0;
					const int meta = metap[rc];
					const bool cone = meta >= 0;
					const int adr = cone ? (meta & 255) : rc, dim = cone ? ((meta >> 8) & 15) : 1, con = cone ? (meta >> 12) : 0;
					const double *wp = cone ? hcb(con, adr) + hcd * (rc - adr) : hw + rc;
					const double *Jc = Jb + adr * nv;
					double wv[DMAX], j0[DMAX], j1[DMAX];
#pragma unroll
					for (int s2 = 0; s2 < DMAX; s2++) {
						const int sc = s2 < dim ? s2 : 0;
						wv[s2] = wp[sc];
						j0[s2] = Jc[sc * nv + (in0 ? li : 0)];
						j1[s2] = Jc[sc * nv + (in1 ? c1 : 0)];
					}
					double s0 = 0, s1 = 0;
#pragma unroll
					for (int s2 = 0; s2 < DMAX; s2++) {
						s0 += s2 < dim ? wv[s2] * j0[s2] : 0.0;
						s1 += s2 < dim ? wv[s2] * j1[s2] : 0.0;
					}
					a0 = (live && in0) ? s0 : 0.0;
					a1 = (live && in1) ? s1 : 0.0;
				};
				// (the three accumulators are PINNED in accumulation registers for the whole loop -- inline asm with `a` constraints:
				//  left to the compiler the loop-carried tiles lived in VGPRs and every slab copied all 24 registers to AGPRs for its
				//  MFMAs and back, v_accvgpr_write x 24 + v_accvgpr_read x 24 + s_nop 12 per slab, a third of the loop.  The hazard
				//  recogniser does not look inside inline asm: the s_nop ahead of each MFMA covers a VALU write of its A / B operands
				//  right before it, successive MFMAs of one slab write different tiles, a tile's next use as SrcC is a whole slab of
				//  LDS reads later, and the results are read 18+ wait states after the last MFMA (mfma_drain))
				mjb_d4 t00 = mjb_d4{ 0, 0, 0, 0 }, t10 = mjb_d4{ 0, 0, 0, 0 }, t11 = mjb_d4{ 0, 0, 0, 0 };
				asm volatile("" : "+a"(t00), "+a"(t10), "+a"(t11));
				// (two slabs per trip: a slab's operands are the end of a chain -- row metadata, addresses, a dozen LDS reads, the
				//  weighted sums -- that the three MFMAs of the previous slab (96 cycles) do not cover; two independent chains side
				//  by side do what one-deep prefetching did not: config 5, 8.2 k -> see profiles/r04 per build)
#ifndef MJB_H_SLABS
#define MJB_H_SLABS 2  // slabs per trip (MI355X, config 5: 1 -> 6.47 M, 2 -> 6.54 M env-steps/s)
#endif
				for (int r0 = 0; r0 < nefc; r0 += 4 * MJB_H_SLABS) {
					double av0[MJB_H_SLABS], av1[MJB_H_SLABS], bv0[MJB_H_SLABS], bv1[MJB_H_SLABS];
#pragma unroll
					for (int u = 0; u < MJB_H_SLABS; u++) fetch(r0 + 4 * u, av0[u], av1[u], bv0[u], bv1[u]);  // (past the last slab: all zero)
#pragma unroll
					for (int u = 0; u < MJB_H_SLABS; u++)
						asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %3, %5, %0\n\tv_mfma_f64_16x16x4_f64 %1, %4, %5, %1\n\tv_mfma_f64_16x16x4_f64 %2, %4, %6, %2"
						             : "+a"(t00), "+a"(t10), "+a"(t11)
						             : "v"(av0[u]), "v"(av1[u]), "v"(bv0[u]), "v"(bv1[u]));
				}
				asm volatile("s_nop 15\n\ts_nop 7" : "+a"(t00), "+a"(t10), "+a"(t11));  // (mfma_drain: XDL write -> VALU read)
#pragma unroll
				// (nv <= 32: the Hessian is stored as its packed lower triangle, entry (r, c <= r) at r (r + 1) / 2 + c -- 465 doubles instead of
				//  900 on config 5, which is what lets THREE wide frames share a CU's LDS; MJB_HPACK below)
				for (int q = 0; q < 4; q++) {
					const int row = lk + 4 * q, r1 = 16 + row;
					if (row < nv && li <= row) H[MJB_HPACK(row, li)] = t00[q];
					if (r1 < nv && li < nv) H[MJB_HPACK(r1, li)] = t10[q];
					if (r1 < nv && c1 <= r1) H[MJB_HPACK(r1, c1)] = t11[q];
				}
			};
			if (mreg && hcd <= 4) {
				MJB_KEEP_BRANCH();
				one_pass(std::integral_constant<int, 4>{});
			} else if (mreg) {
				MJB_KEEP_BRANCH();
				one_pass(std::integral_constant<int, 6>{});
			} else
			for (int ta = 0; ta < ntile; ta++) {
				mjb_d4 acc[4];
#pragma unroll
				for (int tb = 0; tb < 4; tb++) acc[tb] = mjb_d4{ 0, 0, 0, 0 };
				const int ca = 16 * ta + li;
				const bool ina = ca < nv;
				double av, bv[4], an, bn[4];
				operands(0, ca, ina, av, bv, ta);
				for (int r0 = 0; r0 < nefc; r0 += 4) {
					operands(r0 + 4, ca, ina, an, bn, ta);  // (past the last slab: all zero)
#pragma unroll
					for (int tb = 0; tb < 4; tb++)
						if (tb <= ta) acc[tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[tb], acc[tb], 0, 0, 0);
					av = an;
#pragma unroll
					for (int tb = 0; tb < 4; tb++) bv[tb] = bn[tb];
				}
#pragma unroll
				for (int tb = 0; tb < 4; tb++) {
					if (tb > ta) continue;
#pragma unroll
					for (int q = 0; q < 4; q++) {
						const int row = 16 * ta + lk + 4 * q, col = 16 * tb + li;
						// (nv <= 32: H holds J' W J alone; M joins it when the Cholesky loads its row, from the lane's registers)
						if (row < nv && col < nv) H[row * nv + col] = mreg ? acc[tb][q] : Md[row * nv + col] + acc[tb][q];
					}
				}
			}
		}
		sync();
		EPROF(27);
		// Cholesky, lane = row of H; lane j keeps 1 / L_jj so that the substitutions multiply instead of divide and never
		// read the diagonal back.  nv <= 32: right-looking with the lane's ROW IN REGISTERS -- column j is scaled in
		// place, then every lane subtracts l_kj * l_cj from its entries c > j with l_cj fetched from lane c by
		// v_readlane: no LDS traffic and no serial inner product (the subtractions hit each entry in the same order
		// as the left-looking column form below, so the factor is bit-identical).  The forward substitution reads the
		// row from the same registers; the rows are parked in H for the backward one (which needs columns).
		double myrinv = 1.0;
		x = gr;
		if (nv <= 32) {
			MJB_KEEP_BRANCH();
			double Hr[32];
			// (32 loads in flight, each pinned where it is issued: left alone, the compiler sinks a load into the select that uses it
			//  and turns the select into a branch on a lane mask it has spilled -- 32 branches with a full LDS round trip each, a
			//  quarter of this stage.  `lim`: one vector compare per column instead of a held `dofact && c < nv` mask pair)
			const int lim = dofact ? nv : 0;
			// (packed rows: the entries right of the diagonal are never read back -- the factorisation computes them, nothing uses them --
			//  so they load the row's first entry instead, and the store below sends them to a dump slot behind the triangle)
			const int kb = k * (k + 1) / 2;
#pragma unroll
			for (int c = 0; c < 32; c++) {
				Hr[c] = H[kb + (c <= k ? c : 0)];
				asm volatile("" : "+v"(Hr[c]));
			}
#pragma unroll
			for (int c = 0; c < 32; c++) Hr[c] = c < lim ? Mrow[c] + Hr[c] : 0.0;
			// (two half-loops: one 32-column nest exceeds LLVM's pragma-unroll size cap and would leave Hr in scratch)
			chol_cols16<0>(Hr, nv, e.lane, myrinv);
			if (nv > 16) {
				MJB_KEEP_BRANCH();
				chol_schur16(Hr, lane, H, sync);
				chol_cols16<16>(Hr, nv, e.lane, myrinv);
			}
			if (dofact) {
				const int dump = nv * (nv + 1) / 2;
#pragma unroll
				for (int c = 0; c < 32; c++)
					if (c < nv) H[c <= k ? kb + c : dump] = Hr[c];
			}
			// search = -H^-1 grad : lane k holds element k
			// (elements >= nv are zero: no guards inside the halves, see chol_cols16)
			const int ln = e.lane;  // (fresh per substitution, not per Newton solve: see chol_cols16)
#pragma unroll
			for (int i = 0; i < 16; i++) {
				const double xi = wave_bcast(x * myrinv, i);
				if (ln == i) x = xi;
				else x -= ((ln < nv && ln > i) ? Hr[i] : 0.0) * xi;
			}
			if (nv > 16) {
				MJB_KEEP_BRANCH();
#pragma unroll
				for (int i = 16; i < 32; i++) {
					const double xi = wave_bcast(x * myrinv, i);
					if (ln == i) x = xi;
					else x -= ((ln < nv && ln > i) ? Hr[i] : 0.0) * xi;
				}
			}
			sync();
		} else {
			for (int j = 0; j < nv; j++) {
				double s = 0;
				if (dofact && k >= j) {
					s = H[k * nv + j];
#pragma unroll 4
					for (int c = 0; c < j; c++) s -= H[k * nv + c] * H[j * nv + c];
				}
				double sj = wave_bcast(s, j);
				if (sj < MJB_MINVAL) sj = MJB_MINVAL;
				const double rinv = rsqrt(sj);
				if (dofact && k >= j) H[k * nv + j] = (k == j) ? sj * rinv : s * rinv;
				if (k == j) myrinv = rinv;
				sync();
			}
#pragma unroll 4
			for (int i = 0; i < nv; i++) {
				const double xi = wave_bcast(x * myrinv, i);
				const double lki = (dofact && k > i) ? H[k * nv + i] : 0.0;
				if (k == i) x = xi;
				else x -= lki * xi;
			}
		}
		EPROF(31);
		if (nv <= 32) {
			// backward substitution: column k of L (parked in H by rows) fetched up front -- 31 independent LDS reads -- so that
			// the dependent chain is (mul, readlane pair, fma) per row; elements >= nv are zero, no guards
			MJB_KEEP_BRANCH();
			double colk[32];
			const int lb = e.lane;  // (fresh per substitution)
#pragma unroll
			for (int i = 1; i < 32; i++) {
				const int ln = lb;
				const bool has = ln < i && i < nv;
				const double v = H[has ? i * (i + 1) / 2 + ln : 0];
				colk[i] = has ? v : 0.0;
			}
#pragma unroll
			for (int i = 31; i >= 0; i--) {
				if (i >= 16 && nv <= 16) continue;  // (compile-time half, wave-uniform test)
				const double xi = wave_bcast(x * myrinv, i);
				if (lb == i) x = xi;
				else x -= (i > 0 ? colk[i] : 0.0) * xi;
			}
		} else {
#pragma unroll 4
		for (int i = nv - 1; i >= 0; i--) {
			const double xi = wave_bcast(x * myrinv, i);
			const double lik = (dofact && k < i) ? H[i * nv + k] : 0.0;
			if (k == i) x = xi;
			else x -= lik * xi;
		}
		}
		}
		EPROF(28);
		const double sk = dofact ? -x : 0.0;
		cg_sold = sk;
		double ss = sk * sk, g1 = dofact ? sk * (ma - f[L.qfrc_smooth + k]) : 0.0, gz = 0;  // (|search|^2 and the line search's linear Gauss term side by side)
		wave_sum3(ss, g1, gz);
		const double snorm = sqrt(ss);
		if (snorm < MJB_MINVAL) break;
		if (dofact) srch[k] = sk;
		sync();
		// line search along the Newton direction
		double mv, jv[R];
		dots(srch, 0.0, mv, jv);
#pragma unroll
		for (int i = 0; i < R; i++)
			if (rowact[i]) jv_s[rr[i]] = jv[i];
		sync();
#ifdef MJB_PROFILE_LS  // (libmjb_prof_ls.so: slots 20 / 21 / 22 = the line search's M v | J v, its per-contact constants + Gauss terms, its trial points)
		EPROF(20);
#endif
		// per-contact line-search constants: registers of the leader lane (R == 1), or parked in the contact's cone
		// block Hc (free once H is built) when a lane owns several rows and registers are scarce
		ConeLine cl1 = ConeLine{ 0, 0, 0, 0, 0, 0, 0, 0, 1, 0 };
#pragma unroll
		for (int i = 0; i < R; i++) {
			if (leader[i]) {
				ConeLine c = ConeLine{ 0, 0, 0, 0, 0, 0, 0, 0, 1, 0 };
				const double mu = cmu[i];
				const double *cfri = f + L.contact_friction + 5 * rcon[i];
				c.mu = mu;
				c.N0 = mu * jaref[i];
				c.N1 = mu * jv[i];
				c.Dm = D[i] * cdmi[i];
				for (int j = 0; j < 6; j++) {
					if (j >= cdim[i]) break;
					const double xj = jar_s[rr[i] + j], vj = jv_s[rr[i] + j], Dj = Dp[rr[i] + j];
					c.q0b += 0.5 * Dj * xj * xj;
					c.q1b += Dj * xj * vj;
					c.q2b += 0.5 * Dj * vj * vj;
					if (j > 0) {
						const double U = cfri[j - 1] * xj, V = cfri[j - 1] * vj;
						c.TT += U * U;
						c.UV += U * V;
						c.VV += V * V;
					}
				}
				if constexpr (R == 1) {
					cl1 = c;
				} else {
					double *o = hcb(rcon[i], rr[i]);
					o[0] = c.N0; o[1] = c.N1; o[2] = c.TT; o[3] = c.UV; o[4] = c.VV;
					o[5] = c.q0b; o[6] = c.q1b; o[7] = c.q2b; o[8] = c.mu; o[9] = c.Dm;
				}
			}
		}
		const double g0 = gauss;
		const double g2 = wave_sum(dofact ? 0.5 * sk * mv : 0.0);
		const double gtol = tol * 0.01 * snorm / scale;  // mjOption.ls_tolerance = 0.01
#ifdef MJB_PROFILE_LS
		EPROF(21);
#endif
		auto ls_eval = [&](LsPoint &p) {
			const double a = p.alpha;
			double c0 = 0, c1 = 0, c2 = 0;
#pragma unroll
			for (int i = 0; i < R; i++) {
				if (64 * i >= nefc) continue;
				if constexpr (R == 1) {
					ls_row(a, scalar_row[i], bilat[i], leader[i], jaref[i], jv[i], D[i], fl[i], cl1, c0, c1, c2);
				} else {
					ConeLine c = cl1;
					if (leader[i]) {
						const double *o = hcb(rcon[i], rr[i]);
						c = ConeLine{ o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[8], o[9] };
					}
					ls_row(a, scalar_row[i], bilat[i], leader[i], jaref[i], jv[i], D[i], fl[i], c, c0, c1, c2);
				}
			}
			double s0 = c0, s1 = c1, s2 = c2;
			wave_sum3(s0, s1, s2);
			p.cost = a * a * g2 + a * g1 + g0 + s0;
			p.d0 = 2 * a * g2 + g1 + s1;
			p.d1 = 2 * g2 + s2;
			if (p.d1 <= 0) p.d1 = MJB_MINVAL;
		};
		double alpha;
#ifdef MJB_PROFILE_NWT  // (slot 19 counts the trial points of the line search: "cycles per call" = evaluations per iteration)
#define LS_EVAL(P) do { ls_eval(P); prof_rec(e.env, e.lane, 19, 1, 0); } while (0)
#else
#define LS_EVAL(P) ls_eval(P)
#endif
		{
			LsPoint p0, p1, p2, pmid, p1n, p2n;
			int lsit = 0;
			const int maxls = 50;  // mjOption.ls_iterations
			bool done = false;
			p0.alpha = 0;
			LS_EVAL(p0);
			p1.alpha = p0.alpha - p0.d0 / p0.d1;
			LS_EVAL(p1);
			if (p0.cost < p1.cost) p1 = p0;
			alpha = p1.alpha;
			if (fabs(p1.d0) < gtol) done = true;
			const int dir = p1.d0 < 0 ? 1 : -1;
			bool p2update = false;
			p2 = p1;
			while (!done && p1.d0 * dir <= -gtol && lsit < maxls) {
				p2 = p1;
				p2update = true;
				p1.alpha = p1.alpha - p1.d0 / p1.d1;
				LS_EVAL(p1);
				lsit++;
				alpha = p1.alpha;
				if (fabs(p1.d0) < gtol) done = true;
			}
			if (!done && !(lsit >= maxls || !p2update)) {
				p1n.alpha = p1.alpha - p1.d0 / p1.d1;
				p2n.alpha = p2.alpha - p2.d0 / p2.d1;
				while (lsit < maxls) {
					pmid.alpha = 0.5 * (p1.alpha + p2.alpha);
					LS_EVAL(pmid);
					LS_EVAL(p1n);
					LS_EVAL(p2n);
					lsit++;
					// converged candidate with the lowest cost wins (order p1n, p2n, pmid)
					bool have = false;
					double bc = 0, ba = 0;
					if (fabs(p1n.d0) < gtol) { have = true; bc = p1n.cost; ba = p1n.alpha; }
					if (fabs(p2n.d0) < gtol && (!have || p2n.cost < bc)) { have = true; bc = p2n.cost; ba = p2n.alpha; }
					if (fabs(pmid.d0) < gtol && (!have || pmid.cost < bc)) { have = true; bc = pmid.cost; ba = pmid.alpha; }
					if (have) {
						alpha = ba;
						done = true;
						break;
					}
					bool updated = false;
					for (int c = 0; c < 3; c++) {
						const LsPoint q = c == 0 ? p1n : (c == 1 ? p2n : pmid);
						const double lo = p1.alpha < p2.alpha ? p1.alpha : p2.alpha;
						const double hi = p1.alpha < p2.alpha ? p2.alpha : p1.alpha;
						if (!(q.alpha > lo && q.alpha < hi)) continue;
						if ((q.d0 < 0) == (p1.d0 < 0)) p1 = q;
						else p2 = q;
						updated = true;
					}
					if (!updated) break;
					p1n.alpha = p1.alpha - p1.d0 / p1.d1;
					p2n.alpha = p2.alpha - p2.d0 / p2.d1;
				}
				if (!done) alpha = p1.cost < p2.cost ? p1.alpha : p2.alpha;
			}
		}
#undef LS_EVAL
#ifdef MJB_PROFILE_LS
		EPROF(22);
#endif
		EPROF(29);
#ifdef MJB_PROFILE_NWT
		prof_rec(e.env, e.lane, 19, 0, 1);
#endif
		if (alpha == 0) break;
		if (dofact) qa[k] += alpha * sk;
		ma += alpha * mv;
#pragma unroll
		for (int i = 0; i < R; i++) jaref[i] += alpha * jv[i];
		iter++;
		sync();
	}
	if (lane == 0) fi[L.solver_iter] = iter;
	if (dofact) {
		const double a = qa[k];
		f[L.qacc + k] = a;
		f[L.qacc_warmstart + k] = a;
	}
	sync();
}
