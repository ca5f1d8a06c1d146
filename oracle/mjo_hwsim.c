/* mjo_hwsim.c -- CPU restatement (TEST INFRASTRUCTURE, see mjo.h) of DefaultRobotHWSim::writeSim
 * (/root/reference mujoco_ros_control/src/default_robot_hw_sim.cpp:248-326) for ONE env, as the engine's device-side
 * stage implements it (include/mjb.h, mjb_hwsim_*): per controlled joint the command is written into mjData by its
 * control method; PID = control_toolbox::Pid::computeCommand (absent dependency, restated from its documented
 * algorithm).  gains[k] = { p, i, d, i_max, i_min, effort_limit, lower, upper }; pid[k] = { integral, last error }. */
#include <math.h>

#include "mjo.h"

void mjo_hwsim_write(const mjb_model_desc *m, mjo_data *d, int n, const int *joint, const int *method, const int *kind,
                     const int *antiwindup, const double *gains, const double *cmd_pos, const double *cmd_vel,
                     const double *cmd_eff, const double *cmd_hold, double *pid, int estop)
{
	const double dt = m->timestep[0];
	for (int k = 0; k < n; k++) {
		const int j = joint[k], qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
		const double *gn = gains + 8 * k;
		const double pos = d->qpos[qa], vel = d->qvel[da];
		const double cpos = estop ? cmd_hold[k] : cmd_pos[k];
		double error = 0;
		int use_pid = 0;
		switch (method[k]) {
		case MJB_HW_EFFORT: d->qfrc_applied[da] = estop ? 0.0 : cmd_eff[k]; break;
		case MJB_HW_POSITION:
			d->qpos[qa] = cpos;
			d->qvel[da] = 0;
			d->qfrc_applied[da] = 0;
			break;
		case MJB_HW_VELOCITY:
			d->qvel[da] = estop ? 0.0 : cmd_vel[k];
			d->qfrc_applied[da] = 0;
			break;
		case MJB_HW_POSITION_PID:
			if (kind[k] == MJB_HW_REVOLUTE) {
				double c = (gn[7] > gn[6]) ? fmin(fmax(cpos, gn[6]), gn[7]) : cpos;
				error = c - pos;
			} else if (kind[k] == MJB_HW_CONTINUOUS) {
				const double two_pi = 6.283185307179586476925;
				double a = fmod(fmod(cpos - pos, two_pi) + two_pi, two_pi);
				if (a > 0.5 * two_pi) a -= two_pi;
				error = a;
			} else {
				error = cpos - pos;
			}
			use_pid = 1;
			break;
		case MJB_HW_VELOCITY_PID:
			error = estop ? -vel : cmd_vel[k] - vel;
			use_pid = 1;
			break;
		default: break;
		}
		if (use_pid) {
			double ierr = pid[2 * k], last = pid[2 * k + 1];
			double derr = (error - last) / dt;
			ierr += dt * error;
			if (antiwindup[k] && gn[1] != 0) ierr = fmin(fmax(ierr, gn[4] / fabs(gn[1])), gn[3] / fabs(gn[1]));
			double iterm = gn[1] * ierr;
			if (!antiwindup[k]) iterm = fmin(fmax(iterm, gn[4]), gn[3]);
			double cmd = gn[0] * error + iterm + gn[2] * derr;
			if (gn[5] > 0) cmd = fmin(fmax(cmd, -gn[5]), gn[5]);
			pid[2 * k] = ierr;
			pid[2 * k + 1] = error;
			d->qfrc_applied[da] = cmd;
		}
	}
}
