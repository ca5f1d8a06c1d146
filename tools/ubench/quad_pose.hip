// quad_pose.hip -- VERDICT r05 #3: would a quad-per-env layout (4 lanes per env: the x / y / z / w components of the spatial algebra on quad_perm DPP) shorten the
// per-step chain of the lane = env kernel?  The chain's unit is the pose of a body from its parent's: q = q_parent * q_local(joint), R = R(q), p = p_parent + R_parent * p_local
// (mj_kinematics for a hinge).  (a) lane = env: one lane does the whole unit -- 16 + 18 + 9 fp64 multiply-adds and the rest of quat2mat -- 64 envs per wavefront;
// (b) quad: lane c of a quad holds component c (quaternion w x y z; rows of R; p), every product needs the OTHER lanes' components through quad_perm DPP -- which only
// exists for 32-bit moves (v_mov_b32_dpp): two per double -- 16 envs per wavefront.
// Prints cycles per body of a dependent chain (one wavefront) and the instruction counts the ISA shows.  Same arithmetic, checksums compared.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ double quad_get(double v, int sel)  // value of lane `sel` of this lane's quad (sel compile-time after unrolling)
{
	int lo = __double2loint(v), hi = __double2hiint(v);
	switch (sel) {
	case 0: lo = __builtin_amdgcn_mov_dpp(lo, 0x00, 0xf, 0xf, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x00, 0xf, 0xf, true); break;
	case 1: lo = __builtin_amdgcn_mov_dpp(lo, 0x55, 0xf, 0xf, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x55, 0xf, 0xf, true); break;
	case 2: lo = __builtin_amdgcn_mov_dpp(lo, 0xaa, 0xf, 0xf, true); hi = __builtin_amdgcn_mov_dpp(hi, 0xaa, 0xf, 0xf, true); break;
	default: lo = __builtin_amdgcn_mov_dpp(lo, 0xff, 0xf, 0xf, true); hi = __builtin_amdgcn_mov_dpp(hi, 0xff, 0xf, 0xf, true); break;
	}
	return __hiloint2double(hi, lo);
}

// (a) one env per lane
__global__ void k_lane(const double *ql, const double *pl, int nbody, int reps, double *out, unsigned long long *cyc)
{
	double q[4] = { 1, 0, 0, 0 }, p[3] = { 0, 0, 0 }, R[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
	const double s = 1e-3 * threadIdx.x;
	unsigned long long t0 = __builtin_readcyclecounter();
	for (int r = 0; r < reps; r++)
		for (int b = 0; b < nbody; b++) {
			const double a0 = ql[4 * b], a1 = ql[4 * b + 1] + s, a2 = ql[4 * b + 2], a3 = ql[4 * b + 3];
			const double l0 = pl[3 * b], l1 = pl[3 * b + 1], l2 = pl[3 * b + 2];
			// p = p + R * p_local (parent's frame), q = q * q_local, R = R(q)
			p[0] += R[0] * l0 + R[1] * l1 + R[2] * l2;
			p[1] += R[3] * l0 + R[4] * l1 + R[5] * l2;
			p[2] += R[6] * l0 + R[7] * l1 + R[8] * l2;
			const double t0q = q[0] * a0 - q[1] * a1 - q[2] * a2 - q[3] * a3, t1q = q[0] * a1 + q[1] * a0 + q[2] * a3 - q[3] * a2;
			const double t2q = q[0] * a2 - q[1] * a3 + q[2] * a0 + q[3] * a1, t3q = q[0] * a3 + q[1] * a2 - q[2] * a1 + q[3] * a0;
			const double n = 1.5 - 0.5 * (t0q * t0q + t1q * t1q + t2q * t2q + t3q * t3q);  // (one Newton step towards unit length: keeps the chain bounded)
			q[0] = t0q * n; q[1] = t1q * n; q[2] = t2q * n; q[3] = t3q * n;
			const double q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3], q11 = q[1] * q[1], q12 = q[1] * q[2], q13 = q[1] * q[3],
			             q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
			R[0] = q00 + q11 - q22 - q33; R[4] = q00 - q11 + q22 - q33; R[8] = q00 - q11 - q22 + q33;
			R[1] = 2 * (q12 - q03); R[2] = 2 * (q13 + q02); R[3] = 2 * (q12 + q03); R[5] = 2 * (q23 - q01); R[6] = 2 * (q13 - q02); R[7] = 2 * (q23 + q01);
		}
	unsigned long long t1 = __builtin_readcyclecounter();
	out[blockIdx.x * blockDim.x + threadIdx.x] = p[0] + 2 * p[1] + 3 * p[2] + q[0] + q[1] + q[2] + q[3];
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// (b) four lanes per env: lane c holds q[c]; p[c] and row c of R for c < 3 (lane 3 idles there)
__global__ void k_quad(const double *ql, const double *pl, int nbody, int reps, double *out, unsigned long long *cyc)
{
	const int c = threadIdx.x & 3;
	const double s = 1e-3 * (threadIdx.x >> 2);
	double q = c == 0 ? 1.0 : 0.0, p = 0, R0 = c == 0, R1 = c == 1, R2 = c == 2;  // row c of R
	// sign / permutation pattern of the quaternion product for output component c:  t_c = sum_k sg[c][k] * q[k] * a[perm[c][k]]
	const double sg1 = (c == 0) ? -1.0 : 1.0, sg2 = (c == 0 || c == 1) ? -1.0 : 1.0, sg3 = (c == 0 || c == 2) ? -1.0 : 1.0;
	(void)sg2;
	unsigned long long t0 = __builtin_readcyclecounter();
	for (int r = 0; r < reps; r++)
		for (int b = 0; b < nbody; b++) {
			const double a0 = ql[4 * b], a1 = ql[4 * b + 1] + s, a2 = ql[4 * b + 2], a3 = ql[4 * b + 3];
			const double l0 = pl[3 * b], l1 = pl[3 * b + 1], l2 = pl[3 * b + 2];
			p += R0 * l0 + R1 * l1 + R2 * l2;  // (row c of the parent's R: no exchange needed)
			// quaternion product: lane c needs all four q[k]
			const double x0 = quad_get(q, 0), x1 = quad_get(q, 1), x2 = quad_get(q, 2), x3 = quad_get(q, 3);
			// the operand of the local quaternion each term takes, per output component (wave-uniform model data, selected per lane)
			const double b0 = c == 0 ? a0 : (c == 1 ? a1 : (c == 2 ? a2 : a3));
			const double b1 = c == 0 ? a1 : (c == 1 ? a0 : (c == 2 ? a3 : a2));
			const double b2 = c == 0 ? a2 : (c == 1 ? a3 : (c == 2 ? a0 : a1));
			const double b3 = c == 0 ? a3 : (c == 1 ? a2 : (c == 2 ? a1 : a0));
			const double g2 = c == 0 ? -1.0 : (c == 1 ? 1.0 : (c == 2 ? 1.0 : -1.0)), g3 = c == 0 ? -1.0 : (c == 1 ? -1.0 : (c == 2 ? 1.0 : 1.0));
			const double g1 = c == 0 ? -1.0 : (c == 1 ? 1.0 : (c == 2 ? -1.0 : 1.0));
			(void)sg1; (void)sg3;
			double t = x0 * b0 + g1 * x1 * b1 + g2 * x2 * b2 + g3 * x3 * b3;
			// |t|^2 over the quad: two butterfly steps
			double n2 = t * t;
			{
				int lo = __double2loint(n2), hi = __double2hiint(n2);
				lo = __builtin_amdgcn_mov_dpp(lo, 0xb1, 0xf, 0xf, true); hi = __builtin_amdgcn_mov_dpp(hi, 0xb1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
				n2 += __hiloint2double(hi, lo);
				lo = __double2loint(n2); hi = __double2hiint(n2);
				lo = __builtin_amdgcn_mov_dpp(lo, 0x4e, 0xf, 0xf, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x4e, 0xf, 0xf, true);  // quad_perm [2,3,0,1]
				n2 += __hiloint2double(hi, lo);
			}
			q = t * (1.5 - 0.5 * n2);
			// row c of R(q): needs all four components again
			const double y0 = quad_get(q, 0), y1 = quad_get(q, 1), y2 = quad_get(q, 2), y3 = quad_get(q, 3);
			const double d0 = y0 * y0 + y1 * y1 - y2 * y2 - y3 * y3, d1 = y0 * y0 - y1 * y1 + y2 * y2 - y3 * y3, d2 = y0 * y0 - y1 * y1 - y2 * y2 + y3 * y3;
			const double e01 = 2 * (y1 * y2 - y0 * y3), e02 = 2 * (y1 * y3 + y0 * y2), e10 = 2 * (y1 * y2 + y0 * y3), e12 = 2 * (y2 * y3 - y0 * y1),
			             e20 = 2 * (y1 * y3 - y0 * y2), e21 = 2 * (y2 * y3 + y0 * y1);
			R0 = c == 0 ? d0 : (c == 1 ? e10 : e20);
			R1 = c == 0 ? e01 : (c == 1 ? d1 : e21);
			R2 = c == 0 ? e02 : (c == 1 ? e12 : d2);
		}
	unsigned long long t1 = __builtin_readcyclecounter();
	// same checksum as (a), assembled by lane 0 of the quad
	const double p0 = quad_get(p, 0), p1 = quad_get(p, 1), p2 = quad_get(p, 2);
	const double s0 = quad_get(q, 0) + quad_get(q, 1) + quad_get(q, 2) + quad_get(q, 3);
	if (c == 0) out[blockIdx.x * (blockDim.x >> 2) + (threadIdx.x >> 2)] = p0 + 2 * p1 + 3 * p2 + s0;
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
	const int nbody = 9, reps = 2000;
	std::vector<double> ql(4 * nbody), pl(3 * nbody);
	for (int b = 0; b < nbody; b++) {
		const double a = 0.1 + 0.05 * b;
		ql[4 * b] = cos(a); ql[4 * b + 1] = sin(a) * 0.6; ql[4 * b + 2] = sin(a) * 0.0; ql[4 * b + 3] = sin(a) * 0.8;
		pl[3 * b] = 0.1; pl[3 * b + 1] = 0.02 * b; pl[3 * b + 2] = 0.3;
	}
	double *dq, *dp, *o1, *o2;
	unsigned long long *c1, *c2;
	hipMalloc(&dq, ql.size() * 8); hipMalloc(&dp, pl.size() * 8); hipMalloc(&o1, 64 * 8); hipMalloc(&o2, 64 * 8); hipMalloc(&c1, 8); hipMalloc(&c2, 8);
	hipMemcpy(dq, ql.data(), ql.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dp, pl.data(), pl.size() * 8, hipMemcpyHostToDevice);
	for (int w = 0; w < 2; w++) {
		hipLaunchKernelGGL(k_lane, dim3(1), dim3(64), 0, 0, dq, dp, nbody, reps, o1, c1);
		hipLaunchKernelGGL(k_quad, dim3(1), dim3(64), 0, 0, dq, dp, nbody, reps, o2, c2);
	}
	hipDeviceSynchronize();
	double h1[64], h2[16];
	unsigned long long y1, y2;
	hipMemcpy(h1, o1, 64 * 8, hipMemcpyDeviceToHost); hipMemcpy(h2, o2, 16 * 8, hipMemcpyDeviceToHost);
	hipMemcpy(&y1, c1, 8, hipMemcpyDeviceToHost); hipMemcpy(&y2, c2, 8, hipMemcpyDeviceToHost);
	double worst = 0;
	for (int e = 0; e < 16; e++) worst = fmax(worst, fabs(h1[e] - h2[e]));
	printf("{\"bodies\": %d, \"lane_env_cycles_per_body\": %.1f, \"quad_cycles_per_body\": %.1f, \"envs_per_wavefront\": [64, 16], \"checksum_diff\": %.3e}\n", nbody * reps,
	       (double)y1 / (nbody * reps), (double)y2 / (nbody * reps), worst);
	return 0;
}
