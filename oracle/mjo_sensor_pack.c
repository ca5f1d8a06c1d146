/* mjo_sensor_pack.c -- CPU restatement (TEST INFRASTRUCTURE, see mjo.h) of what the reference's sensors plugin
 * publishes from `sensordata` after every step:
 *   /root/reference mujoco_ros_sensors/src/mujoco_sensor_handler_plugin.cpp:175-437 (lastStageCallback),
 *   noise registration :123-173 (set_flag bit k = noise on component k; mean / sigma are packed: the n-th SET bit
 *   uses mean[n], sigma[n]).
 * Per sensor (cutoff' = cutoff > 0 ? cutoff : 1), messages are float32:
 *   ground truth                      = sensordata / cutoff'
 *   value, no noise model registered  = sensordata / cutoff'
 *   value, noise model registered     = sensordata + noise_k / cutoff'   (the reference does NOT divide the reading in
 *                                       this branch -- kept as is), noise_k = N(0,1) sigma + mean for set bits, else 0;
 *                                       scalar sensors always use mean[0], sigma[0]
 *   quaternions (framequat, ballquat) = normalize(setRPY(noise_r, noise_p, noise_y) * normalize(float32 ground truth))
 * The reference draws from one std::mt19937 in sensor order; the batched engine replaces it by the counter-based
 * Philox stream keyed (seed, env, step, sensordata address + component), identical here and in the HIP kernel. */
#include <math.h>
#include <stdint.h>

#include "mjo.h"

static int is_quat(int type) { return type == MJB_SENS_BALLQUAT || type == MJB_SENS_FRAMEQUAT; }

void mjo_sensor_pack(const mjb_model_desc *m, const double *sensordata, const int *set_flag, const double *mean,
                     const double *sigma, uint64_t seed, uint64_t env, uint32_t step, float *value, float *truth)
{
	seed ^= 0x53454e534f52ULL;  /* a stream of its own next to the ctrl-noise injector's */
	for (int n = 0; n < m->nsensor; n++) {
		const int adr = m->sensor_adr[n], dim = m->sensor_dim[n], type = m->sensor_type[n];
		const double cutoff = m->sensor_cutoff[n] > 0 ? m->sensor_cutoff[n] : 1.0;
		const int flag = set_flag ? set_flag[n] : 0;
		for (int k = 0; k < dim; k++) truth[adr + k] = (float)(sensordata[adr + k] / cutoff);
		if (flag == 0) {
			for (int k = 0; k < dim; k++) value[adr + k] = truth[adr + k];
			continue;
		}
		double noise[3] = { 0, 0, 0 };
		int idx = 0;
		for (int k = 0; k < 3; k++) {
			if (dim == 1 && k > 0) break;
			if (dim == 1 || (flag & (1 << k))) {
				noise[k] = mjo_normal(seed, env, step, (uint32_t)(adr + k)) * sigma[3 * n + idx] + mean[3 * n + idx];
				idx++;
			}
		}
		if (is_quat(type)) {
			/* tf2: q_orig = normalize(msg), q_rot = setRPY(r, p, y), result = normalize(q_rot * q_orig); (w, x, y, z) order */
			double q[4] = { truth[adr], truth[adr + 1], truth[adr + 2], truth[adr + 3] };
			double nq = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
			for (int k = 0; k < 4; k++) q[k] /= nq;
			const double cr = cos(0.5 * noise[0]), sr = sin(0.5 * noise[0]), cp = cos(0.5 * noise[1]), sp = sin(0.5 * noise[1]);
			const double cy = cos(0.5 * noise[2]), sy = sin(0.5 * noise[2]);
			double r[4] = { cr * cp * cy + sr * sp * sy, sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy,
				            cr * cp * sy - sr * sp * cy };
			double nr = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
			for (int k = 0; k < 4; k++) r[k] /= nr;
			double o[4] = { r[0] * q[0] - r[1] * q[1] - r[2] * q[2] - r[3] * q[3], r[0] * q[1] + r[1] * q[0] + r[2] * q[3] - r[3] * q[2],
				            r[0] * q[2] - r[1] * q[3] + r[2] * q[0] + r[3] * q[1], r[0] * q[3] + r[1] * q[2] - r[2] * q[1] + r[3] * q[0] };
			double no = sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
			for (int k = 0; k < 4; k++) value[adr + k] = (float)(o[k] / no);
		} else {
			for (int k = 0; k < dim; k++) value[adr + k] = (float)(sensordata[adr + k] + (k < 3 ? noise[k] : 0.0) / cutoff);
		}
	}
}
