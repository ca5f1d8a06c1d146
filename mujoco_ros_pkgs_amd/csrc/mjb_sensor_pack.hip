// mjb_sensor_pack.hip -- device-side equivalent of the reference's sensors plugin:
// `sensordata` -> the float32 values its lastStageCallback publishes every step, with the registered per-axis
// Gaussian noise models (/root/reference mujoco_ros_sensors/src/mujoco_sensor_handler_plugin.cpp:175-437, noise
// registration :123-173).  One thread per (env, sensor); HBM-bound: reads 8 B and writes 2 x 4 B per component.
// Semantics (incl. the reference's un-normalised noisy branch) are spelled out in oracle/mjo_sensor_pack.c, which
// this kernel follows operation for operation; the RNG is the engine's Philox stream keyed
// (seed ^ "SENSOR", global env, step, sensordata address + component).
#include <hip/hip_runtime.h>

#include "mjb_dev.h"
#include "mjb_math.h"

namespace {

__global__ void __launch_bounds__(256) mjb_sensor_pack_kernel(const KernelParams MJB_AS4 *__restrict__ P, const int *__restrict__ set_flag,
                                                              const double *__restrict__ mean, const double *__restrict__ sigma,
                                                              unsigned long long seed, long long env_offset, unsigned int step,
                                                              float *__restrict__ value, float *__restrict__ truth)
{
	const DevModel MJB_AS4 &m = P->m;
	const DevState MJB_AS4 &s = P->s;
	const int nsens = m.nsensor, S = m.nsensordata;
	const long long total = (long long)s.nenv * nsens;
	for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
		const int env = (int)(t / nsens), n = (int)(t - (long long)env * nsens);
		const int adr = m.sensor_adr[n], dim = m.sensor_dim[n], type = m.sensor_type[n];
		const double cut = m.sensor_cutoff[n] > 0 ? m.sensor_cutoff[n] : 1.0;
		const double *sd = s.sensordata + (size_t)env * S + adr;
		float *val = value + (size_t)env * S + adr, *tru = truth + (size_t)env * S + adr;
		const int flag = set_flag[n];
		float tr[4] = { 0, 0, 0, 0 };
		for (int k = 0; k < 4; k++)
			if (k < dim) {
				tr[k] = (float)(sd[k] / cut);
				tru[k] = tr[k];
			}
		if (flag == 0) {
			for (int k = 0; k < 4; k++)
				if (k < dim) val[k] = tr[k];
			continue;
		}
		double noise[3] = { 0, 0, 0 };
		int idx = 0;
		const unsigned long long genv = (unsigned long long)(env_offset + env);
		for (int k = 0; k < 3; k++) {
			if (dim == 1 && k > 0) break;
			if (dim == 1 || (flag & (1 << k))) {
				noise[k] = philox_normal(seed, genv, step, (unsigned int)(adr + k)) * sigma[3 * n + idx] + mean[3 * n + idx];
				idx++;
			}
		}
		if (type == MJB_SENS_BALLQUAT || type == MJB_SENS_FRAMEQUAT) {
			double q[4] = { tr[0], tr[1], tr[2], tr[3] };
			const double nq = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
			for (int k = 0; k < 4; k++) q[k] /= nq;
			const double cr = cos(0.5 * noise[0]), sr = sin(0.5 * noise[0]), cp = cos(0.5 * noise[1]), sp = sin(0.5 * noise[1]);
			const double cy = cos(0.5 * noise[2]), sy = sin(0.5 * noise[2]);
			double r[4] = { cr * cp * cy + sr * sp * sy, sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy,
				            cr * cp * sy - sr * sp * cy };
			const double nr = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
			for (int k = 0; k < 4; k++) r[k] /= nr;
			double o[4];
			qmul(o, r, q);
			const double no = sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
			for (int k = 0; k < 4; k++) val[k] = (float)(o[k] / no);
		} else {
			for (int k = 0; k < 4; k++)
				if (k < dim) val[k] = (float)(sd[k] + (k < 3 ? noise[k] : 0.0) / cut);
		}
	}
}

}  // namespace

int mjb_launch_sensor_pack(const KernelParams *Pdev, int nenv, int nsensor, const int *set_flag, const double *mean,
                           const double *sigma, unsigned long long seed, long long env_offset, unsigned int step,
                           float *value, float *truth, void *stream)
{
	const long long total = (long long)nenv * nsensor;
	if (total <= 0) return 0;
	long long blocks = (total + 255) / 256;
	if (blocks > 256 * 64) blocks = 256 * 64;  // grid-stride beyond 64 blocks per CU
	hipLaunchKernelGGL(mjb_sensor_pack_kernel, dim3((unsigned int)blocks), dim3(256), 0, (hipStream_t)stream,
	                   (const KernelParams MJB_AS4 *)Pdev, set_flag, mean, sigma, seed ^ 0x53454e534f52ULL, env_offset, step,
	                   value, truth);
	return (int)hipGetLastError();
}
