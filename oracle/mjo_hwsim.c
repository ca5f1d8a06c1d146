/* mjo_hwsim.c -- CPU restatement (TEST INFRASTRUCTURE, see mjo.h) of DefaultRobotHWSim::writeSim
 * (/root/reference mujoco_ros_control/src/default_robot_hw_sim.cpp:248-326) for ONE env, as the engine's device-side
 * stage implements it (include/mjb.h, mjb_hwsim_*): per controlled joint the command is written into mjData by its
 * control method; PID = control_toolbox::Pid::computeCommand (absent dependency, restated from its documented
 * algorithm).  gains[k] = { p, i, d, i_max, i_min, effort_limit, lower, upper }; pid[k] = { integral, last error }. */
#include <math.h>
#include <stdlib.h>

#include "mjo.h"

/* ---- ROS package `angles` (angles/angles.h; a dependency of the reference that is absent from /root/reference), restated
 * from its published header: shortest_angular_distance, two_pi_complement, find_min_max_delta,
 * shortest_angular_distance_with_limits ---- */
#define TWO_PI 6.283185307179586476925
static double angdist(double from, double to)
{
	double a = fmod(fmod(to - from, TWO_PI) + TWO_PI, TWO_PI); /* normalize_angle_positive */
	if (a > 0.5 * TWO_PI) a -= TWO_PI;
	return a;
}

static double two_pi_complement(double a)
{
	if (a > TWO_PI || a < -TWO_PI) a = fmod(a, TWO_PI);
	if (a < 0) return TWO_PI + a;
	if (a > 0) return -TWO_PI + a;
	return TWO_PI;
}

static int find_min_max_delta(double from, double left, double right, double *dmin, double *dmax)
{
	const double pi = 3.14159265358979323846;
	double d0 = angdist(from, left), d1 = angdist(from, right), d2 = two_pi_complement(d0), d3 = two_pi_complement(d1);
	if (d0 == 0) {
		*dmin = d0;
		*dmax = fmax(d1, d3);
		return 1;
	}
	if (d1 == 0) {
		*dmax = d1;
		*dmin = fmin(d0, d2);
		return 1;
	}
	double lo = d0, lo2 = d2, hi = d1, hi2 = d3;
	if (d2 < lo) { lo = d2; lo2 = d0; }
	if (d3 > hi) { hi = d3; hi2 = d1; }
	if (lo <= hi2 || hi >= lo2) {
		*dmin = hi2;
		*dmax = lo2;
		return left == -pi && right == pi;
	}
	*dmin = lo;
	*dmax = hi;
	return 1;
}

static double angdist_with_limits(double from, double to, double left, double right)
{
	double dmin = -TWO_PI, dmax = TWO_PI, tmin = -TWO_PI, tmax = TWO_PI;
	int inside = find_min_max_delta(from, left, right, &dmin, &dmax);
	double delta = angdist(from, to), comp = two_pi_complement(delta);
	if (inside) {
		if (delta >= dmin && delta <= dmax) return delta;
		if (comp >= dmin && comp <= dmax) return comp;
		find_min_max_delta(to, left, right, &tmin, &tmax);
		if (fabs(tmin) < fabs(tmax)) return fmax(delta, comp);
		if (fabs(tmin) > fabs(tmax)) return fmin(delta, comp);
		return fabs(delta) < fabs(comp) ? delta : comp;
	}
	find_min_max_delta(to, left, right, &tmin, &tmax);
	if (fabs(dmin) < fabs(dmax)) return fmin(delta, comp);
	if (fabs(dmin) > fabs(dmax)) return fmax(delta, comp);
	return fabs(delta) < fabs(comp) ? delta : comp;
}

/* writeSim proper.  jpos / jvel: the joint state the PID sees -- DefaultRobotHWSim's joint_position_ / joint_velocity_, what
 * readSim sampled at the last controller update (NULL: mjData's qpos / qvel of this step); dt: the period writeSim is called with */
static void hwsim_write_core(const mjb_model_desc *m, mjo_data *d, int n, const int *joint, const int *method, const int *kind,
                             const int *antiwindup, const double *gains, const double *cmd_pos, const double *cmd_vel,
                             const double *cmd_eff, const double *cmd_hold, double *pid, int estop, const double *jpos,
                             const double *jvel, double dt)
{
	for (int k = 0; k < n; k++) {
		const int j = joint[k], qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
		const double *gn = gains + 8 * k;
		const double pos = jpos ? jpos[k] : d->qpos[qa], vel = jvel ? jvel[k] : d->qvel[da];
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
				/* command saturated to the joint limits (pj_sat_interface_.enforceLimits, :263), error by
				 * angles::shortest_angular_distance_with_limits (:289-291) */
				int lim = gn[7] > gn[6];
				double c = lim ? fmin(fmax(cpos, gn[6]), gn[7]) : cpos;
				error = lim ? angdist_with_limits(pos, c, gn[6], gn[7]) : c - pos;
			} else if (kind[k] == MJB_HW_CONTINUOUS) {
				error = angdist(pos, cpos);
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

void mjo_hwsim_write(const mjb_model_desc *m, mjo_data *d, int n, const int *joint, const int *method, const int *kind,
                     const int *antiwindup, const double *gains, const double *cmd_pos, const double *cmd_vel,
                     const double *cmd_eff, const double *cmd_hold, double *pid, int estop)
{
	hwsim_write_core(m, d, n, joint, method, kind, antiwindup, gains, cmd_pos, cmd_vel, cmd_eff, cmd_hold, pid, estop, NULL, NULL,
	                 m->timestep[0]);
}

/* ros::Time(double).toNSec(): sec = floor(t), nsec = round((t - sec) 1e9) (ros/time.h, TimeBase::fromSec) */
static long ros_ns(double t)   /* (long: 64 bits on the LP64 hosts this builds for) */
{
	const double sec = floor(t);
	return (long)sec * 1000000000L + (long)floor((t - sec) * 1e9 + 0.5);
}

/* MujocoRosControlPlugin::controlCallback (/root/reference mujoco_ros_control/src/mujoco_ros_control_plugin.cpp:153-194) around
 * writeSim, for ONE env at sim time d->time (what ros::Time::now() reads inside the callback: /clock carries the time the step
 * started at).  cad = { last_update_sim_time [ns], last_write_sim_time [ns], joint_position_[n], joint_velocity_[n] }, initially
 * { 0, 0, 1.0 .., 0.0 .. } (default_robot_hw_sim.cpp:129-130).
 *   - a time that went backwards (reset) re-arms both stamps (:160-169);
 *   - the controllers are updated -- here: readSim samples the joint state the PIDs will see (:229-245; revolute / continuous
 *     joints unwrapped by shortest_angular_distance) -- when a control period has passed, or on the first call at a non-zero time
 *     (:171-176): nothing is read or written at t = 0;
 *   - writeSim runs whenever an update has ever happened and time moved since the last write, with period = time - last write
 *     (:190-193): every step, also between controller updates, on the joint state of the LAST update;
 *   - the e-stop edge (:180-185) restarts the controller manager's controllers, which live on the host: the PIDs of
 *     DefaultRobotHWSim itself (pid_controllers_, :300, :319) are never reset by the reference, and are not here.
 * Returns 1 when writeSim ran. */
int mjo_hwsim_control_callback(const mjb_model_desc *m, mjo_data *d, int n, const int *joint, const int *method, const int *kind,
                               const int *antiwindup, const double *gains, const double *cmd_pos, const double *cmd_vel,
                               const double *cmd_eff, const double *cmd_hold, double *pid, int estop, double *cad,
                               double control_period)
{
	const long t = ros_ns(d->time[0]), period_ns = ros_ns(control_period);
	long lu = (long)cad[0], lw = (long)cad[1];
	double *jp = cad + 2, *jv = jp + n;
	if (t < lu) {
		lu = t;
		lw = t;
	}
	const long sim_period = t - lu;
	const int reset_ctrls = lu == 0;
	if (sim_period >= period_ns || (reset_ctrls && sim_period != 0)) {
		lu = t;
		for (int k = 0; k < n; k++) {  /* readSim */
			const int j = joint[k];
			const double position = d->qpos[m->jnt_qposadr[j]];
			if (kind[k] == MJB_HW_PRISMATIC) jp[k] = position;
			else jp[k] += angdist(jp[k], position);
			jv[k] = d->qvel[m->jnt_dofadr[j]];
		}
	}
	int wrote = 0;
	if (lu != 0 && t > lw) {
		hwsim_write_core(m, d, n, joint, method, kind, antiwindup, gains, cmd_pos, cmd_vel, cmd_eff, cmd_hold, pid, estop, jp, jv,
		                 1e-9 * (double)(t - lw));
		lw = t;
		wrote = 1;
	}
	cad[0] = (double)lu;
	cad[1] = (double)lw;
	return wrote;
}
