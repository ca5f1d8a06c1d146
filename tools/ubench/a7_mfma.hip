// A7 (mj_projectConstraint) on the matrix cores?  Measured, not argued.
//
// The PGS kernel needs, per env-step, AR = B J' (+ diag R) with B = J M^-1: nefc x nv times nv x nefc, and it needs ROW r of AR
// in the REGISTERS of lane r (the Gauss-Seidel sweep does res += AR[.][i] * delta with no memory access).  In the kernel B_r is
// already in lane r's registers (16 doubles, solved there) and J is in LDS.  Two ways to get the rows:
//   valu : what mjb_constraint.h does -- AR_ri = B_r . J_i, row J_i read as LDS broadcasts at immediate offsets, four rows per
//          wave-uniform guard, result lands where it is needed.
//   mfma : v_mfma_f64_16x16x4_f64 tiles.  Operand A[i][k] must come from lane (i, k) = another lane's register -> B is written to
//          LDS first (nefc x 16 doubles); the 16 x 16 result tile arrives as D[(l >> 4) + 4 q][l & 15] -> it goes back through LDS
//          (a 16 x 64 strip at a time) and every lane reads its row from there.
// One wavefront per "env", many envs per launch, s_memtime around the build (cycles per build, median over the waves).  The
// checksum column verifies both produce the same matrix.  Extra LDS the mfma path needs per env: B (8 KB) + one 16 x 64 strip
// (8 KB) = 16 KB on top of J (the PGS frame of config 3 has 20 480 bytes in total at 8 envs per CU).
//   hipcc -O3 --offload-arch=gfx950 a7_mfma.hip -o a7_mfma && ./a7_mfma
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int NV = 16;  // padded nv (config 3: 15)

__device__ __forceinline__ double wave_sum(double v)
{
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
	return v;
}

template <bool MFMA>
__global__ void __launch_bounds__(64) k_build(const double *Jg, const double *Bg, int nefc, int reps, double *sum, unsigned long long *cyc)
{
	__shared__ __attribute__((aligned(16))) double J[64 * NV];
	__shared__ __attribute__((aligned(16))) double Bs[MFMA ? 64 * NV : 1];
	__shared__ __attribute__((aligned(16))) double strip[MFMA ? 16 * 64 : 1];
	const int lane = threadIdx.x;
	for (int t = lane; t < 64 * NV; t += 64) J[t] = Jg[t];
	double x[NV];
#pragma unroll
	for (int k = 0; k < NV; k++) x[k] = Bg[lane * NV + k];
	__syncthreads();
	double AR[64];
	double chk = 0;
	unsigned long long total = 0;
	for (int rep = 0; rep < reps; rep++) {
#pragma unroll
		for (int i = 0; i < 64; i++) AR[i] = 0;
		asm volatile("" ::: "memory");
		const unsigned long long t0 = __builtin_readcyclecounter();
		if constexpr (!MFMA) {
#pragma unroll
			for (int i = 0; i < 64; i += 4) {
				if (i < nefc) {
					asm volatile("" ::: "memory");
					const double *J0 = J + i * NV;
#pragma unroll
					for (int q = 0; q < 4; q++) {
						const double *Ji = J0 + q * NV;
						double a0 = 0, a1 = 0;
#pragma unroll
						for (int k = 0; k < NV; k += 2) {
							a0 += x[k] * Ji[k];
							a1 += x[k + 1] * Ji[k + 1];
						}
						AR[i + q] = a0 + a1;
					}
				}
			}
		} else {
			// B rows -> LDS (operand A is indexed by (row, k) = (another lane, its register))
#pragma unroll
			for (int k = 0; k < NV; k++) Bs[lane * NV + k] = x[k];
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			const int li = lane & 15, lk = lane >> 4;
			const int nt = (nefc + 15) >> 4;
			for (int ta = 0; ta < nt; ta++) {  // 16 rows of AR at a time
				for (int tb = 0; tb < nt; tb++) {
					d4 acc = { 0, 0, 0, 0 };
#pragma unroll
					for (int k0 = 0; k0 < NV; k0 += 4) {
						const double a = Bs[(16 * ta + li) * NV + k0 + lk];  // A[i][kk] = B[16 ta + i][k0 + kk]
						const double b = J[(16 * tb + li) * NV + k0 + lk];   // B[kk][j] = J[16 tb + j][k0 + kk]
						acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
					}
#pragma unroll
					for (int q = 0; q < 4; q++) strip[(lk + 4 * q) * 64 + 16 * tb + li] = acc[q];  // D[(l >> 4) + 4 q][l & 15]
				}
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
				// the 16 lanes that own these rows pick them up (static register indices: every lane runs the loop, 16 keep the values)
				const bool mine = (lane >> 4) == ta;
				const double *row = strip + (lane & 15) * 64;
#pragma unroll
				for (int i = 0; i < 64; i++) {
					const double v = row[i];
					AR[i] = mine ? v : AR[i];
				}
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
			}
		}
		asm volatile("" ::: "memory");
		double keep = 0;
#pragma unroll
		for (int i = 0; i < 64; i++) keep += (i < nefc && lane < nefc) ? AR[i] * (1 + i) : 0.0;
		const unsigned long long t1 = __builtin_readcyclecounter();
		total += t1 - t0;
		chk += keep;
	}
	const double s = wave_sum(chk);
	if (lane == 0) {
		sum[blockIdx.x] = s / reps;
		cyc[blockIdx.x] = total / reps;
	}
}

int main()
{
	std::vector<double> J(64 * NV), B(64 * NV);
	for (int i = 0; i < 64 * NV; i++) {
		J[i] = ((i * 2654435761u) % 1000) / 1000.0 - 0.5;
		B[i] = ((i * 40503u + 17) % 1000) / 1000.0 - 0.5;
	}
	for (int r = 0; r < 64; r++) J[r * NV + 15] = B[r * NV + 15] = 0;  // nv = 15
	double *dJ, *dB, *dsum;
	unsigned long long *dcyc;
	const int blocks = 2048, reps = 50;
	hipMalloc(&dJ, J.size() * 8);
	hipMalloc(&dB, B.size() * 8);
	hipMalloc(&dsum, blocks * 8);
	hipMalloc(&dcyc, blocks * 8);
	hipMemcpy(dJ, J.data(), J.size() * 8, hipMemcpyHostToDevice);
	hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
	std::vector<double> hs(blocks);
	std::vector<unsigned long long> hc(blocks);
	printf("{\"bench\": \"A7 AR = B J' row-per-lane, one wavefront per env, nv = 15\", \"results\": [");
	bool first = true;
	for (int nefc : { 20, 32, 64 }) {
		for (int mf = 0; mf < 2; mf++) {
			if (mf) hipLaunchKernelGGL(k_build<true>, dim3(blocks), dim3(64), 0, 0, dJ, dB, nefc, reps, dsum, dcyc);
			else hipLaunchKernelGGL(k_build<false>, dim3(blocks), dim3(64), 0, 0, dJ, dB, nefc, reps, dsum, dcyc);
			hipDeviceSynchronize();
			hipMemcpy(hs.data(), dsum, blocks * 8, hipMemcpyDeviceToHost);
			hipMemcpy(hc.data(), dcyc, blocks * 8, hipMemcpyDeviceToHost);
			std::sort(hc.begin(), hc.end());
			printf("%s{\"nefc\": %d, \"path\": \"%s\", \"cycles_per_build_median\": %llu, \"checksum\": %.9f, \"extra_lds_bytes\": %d}", first ? "" : ", ", nefc,
			       mf ? "mfma_f64_16x16x4 + LDS redistribution" : "valu (B row in registers, J as LDS broadcasts)", hc[blocks / 2], hs[0], mf ? 16384 : 0);
			first = false;
		}
	}
	printf("]}\n");
	return 0;
}
