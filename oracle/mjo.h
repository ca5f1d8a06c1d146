/* mjo.h — CPU ORACLE of the batched step engine.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C fp64, single-env restatement of what `mj_step(model, data)` computes at the reference's
 * three call sites (/root/reference mujoco_ros/src/mujoco_env.cpp:498, :552, :593) and of
 * `mj_forward` / `mj_resetData` (mujoco_env.cpp:329, :621, :252).  The arithmetic itself lives in the
 * third-party dependency google-deepmind/mujoco, pinned 2.3.7 (mujoco_ros/CMakeLists.txt:61,
 * .github/workflows/ci.yaml:23), which is NOT vendored under /root/reference and is absent from
 * this image.  Each function therefore restates the published algorithm of
 * the named MuJoCo 2.3.7 engine function ([UPSTREAM] engine_*.c) and is anchored on the
 * reference's call sites and on what the reference's own tests pin at that boundary
 * (SURVEY.md §8c).
 *
 * PARITY UNPINNED: the reference holds no golden vector for any dynamic quantity and libmujoco
 * cannot be built or loaded here, so this oracle is pinned only by (a) the reference-test facts of
 * SURVEY.md §8c (time advance, reset state, rest equilibrium), (b) closed-form / independently
 * derived dynamics (tests/test_oracle_*.py), (c) internal identities.  It has not been compared with
 * real MuJoCo output.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this library.
 */
#ifndef MJO_H_
#define MJO_H_

#include <stdint.h>

#include "../include/mjb.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MJO_MINVAL 1e-15 /* mjMINVAL */

typedef struct mjo_data {
#define MJB_DS(name, rows, cols) double *name;
#define MJB_DD(name, rows, cols) double *name;
#define MJB_DD2(name, rows, cols) double *name;
#define MJB_DI(name, rows, cols) int *name;
#include "../include/mjb_data_fields.def"
#undef MJB_DS
#undef MJB_DD
#undef MJB_DD2
#undef MJB_DI
	/* mjData.warning[].number, accumulated over the life of the data (not cleared by mjo_reset_data -- the engine's batch
	 * counters, mjb_warning, accumulate the same way) */
	unsigned long long warning[MJB_NWARNING];
	/* collision-function override per geom-type pair (mjb_register_collision; index 8 * min(type) + max(type)) */
	int colfunc[64];
	/* this env's geom sizes / types (mjb_set_env_geom_size / _type); NULL: the model's */
	double *env_geom_size;
	int *env_geom_type;
	/* scratch for the Euler implicit-damping solve */
	double *scratch_MM;
	double *scratch_nv;
	double *scratch_nv2;
	double *rk_warmstart; /* [nv] RK4: the warmstart the step came in with (every sub-stage evaluation starts from it) */
	double *rk_buf;       /* [nq + 5 nv + nsensordata + 1 + 2 na] RK4 state of a step cut at its callback points (mjo_step2_rk) */
} mjo_data;

mjo_data *mjo_make_data(const mjb_model_desc *m); /* mj_makeData  */
void mjo_free_data(mjo_data *d);                  /* mj_deleteData */
int mjo_model_desc_size(void);                    /* sizeof(mjb_model_desc) this library was built with */
void mjo_reset_data(const mjb_model_desc *m, mjo_data *d); /* mj_resetData */

/* field access by mjb_field id (NULL if unknown); *n receives the element count */
double *mjo_field(const mjb_model_desc *m, mjo_data *d, int field, int *n);
int *mjo_field_int(const mjb_model_desc *m, mjo_data *d, int field, int *n);

/* ---- stages, one per row of SURVEY.md §8a ---- */
void mjo_kinematics(const mjb_model_desc *m, mjo_data *d);       /* A1 mj_kinematics      */
void mjo_com_pos(const mjb_model_desc *m, mjo_data *d);          /* A1 mj_comPos          */
void mjo_crb(const mjb_model_desc *m, mjo_data *d);              /* A2 mj_crb             */
void mjo_factor_m(const mjb_model_desc *m, mjo_data *d);         /* A3 mj_factorM         */
void mjo_solve_m(const mjb_model_desc *m, mjo_data *d, double *x); /* mj_solveM (in place) */
void mjo_transmission(const mjb_model_desc *m, mjo_data *d);     /* mj_transmission (joint) */
void mjo_com_vel(const mjb_model_desc *m, mjo_data *d);          /* A8 mj_comVel          */
void mjo_passive(const mjb_model_desc *m, mjo_data *d);          /* A8 mj_passive         */
void mjo_rne(const mjb_model_desc *m, mjo_data *d);              /* A9 mj_rne(flg_acc=0)  */
void mjo_fwd_actuation(const mjb_model_desc *m, mjo_data *d);    /* A12 mj_fwdActuation   */
void mjo_fwd_acceleration(const mjb_model_desc *m, mjo_data *d); /* A12 mj_fwdAcceleration*/
void mjo_sensor(const mjb_model_desc *m, mjo_data *d, int stage); /* A15 mj_sensorPos/Vel/Acc */
void mjo_euler(const mjb_model_desc *m, mjo_data *d);            /* A16 mj_Euler          */
void mjo_rk4(const mjb_model_desc *m, mjo_data *d);              /*     mj_RungeKutta(4)  */

/* constraint path (mjo_constraint.c) */
void mjo_collision(const mjb_model_desc *m, mjo_data *d);         /* A4+A5 mj_collision      */
void mjo_make_constraint(const mjb_model_desc *m, mjo_data *d);   /* A6 mj_makeConstraint    */
void mjo_project_constraint(const mjb_model_desc *m, mjo_data *d);/* A7 mj_projectConstraint */
void mjo_reference_constraint(const mjb_model_desc *m, mjo_data *d); /* A8 mj_referenceConstraint */
void mjo_fwd_constraint(const mjb_model_desc *m, mjo_data *d);    /* A13 mj_fwdConstraint (PGS) */

/* composite entry points */
void mjo_fwd_position(const mjb_model_desc *m, mjo_data *d);
void mjo_fwd_velocity(const mjb_model_desc *m, mjo_data *d);
void mjo_forward(const mjb_model_desc *m, mjo_data *d); /* mj_forward */
void mjo_step(const mjb_model_desc *m, mjo_data *d);    /* mj_step    */
/* split halves used to emulate the control-callback point: step1 = up to (excl.) mjcb_control */
void mjo_step1(const mjb_model_desc *m, mjo_data *d);
void mjo_step2(const mjb_model_desc *m, mjo_data *d);
/* mjo_step2 of an RK4 step cut at the callback points of its four evaluations (rk = 0..3; four calls in a row == mjo_step2) */
void mjo_step2_rk(const mjb_model_desc *m, mjo_data *d, int rk);

/* The reference's ctrl-noise injector (mujoco_env.cpp:469-481) with the engine's counter-based
 * normal generator: Philox-4x32-10 keyed by seed, counter (env, step, actuator) + Box-Muller. */
void mjo_philox4x32(uint32_t ctr[4], const uint32_t key[2]);
double mjo_normal(uint64_t seed, uint64_t env, uint32_t step, uint32_t idx);
/* the sensors plugin's per-step messages (mjo_sensor_pack.c) */
void mjo_sensor_pack(const mjb_model_desc *m, const double *sensordata, const int *set_flag, const double *mean,
                     const double *sigma, uint64_t seed, uint64_t env, uint32_t step, float *value, float *truth);
/* DefaultRobotHWSim::writeSim for one env (mjo_hwsim.c) */
void mjo_hwsim_write(const mjb_model_desc *m, mjo_data *d, int n, const int *joint, const int *method, const int *kind,
                     const int *antiwindup, const double *gains, const double *cmd_pos, const double *cmd_vel,
                     const double *cmd_eff, const double *cmd_hold, double *pid, int estop);
/* MujocoRosControlPlugin::controlCallback around writeSim (mujoco_ros_control_plugin.cpp:153-194): controller-update cadence,
 * readSim sampling, write period; cad = { last update [ns], last write [ns], joint_position_[n], joint_velocity_[n] } */
int mjo_hwsim_control_callback(const mjb_model_desc *m, mjo_data *d, int n, const int *joint, const int *method, const int *kind,
                               const int *antiwindup, const double *gains, const double *cmd_pos, const double *cmd_vel,
                               const double *cmd_eff, const double *cmd_hold, double *pid, int estop, double *cad,
                               double control_period);
void mjo_rne_post_constraint(const mjb_model_desc *m, mjo_data *d);
int mjo_needs_rne_post(const mjb_model_desc *m);
void mjo_subtree_vel(const mjb_model_desc *m, const mjo_data *d, int id, double *linvel, double *angmom); /* mj_subtreeVel, one subtree's results */
void mjo_register_collision(mjo_data *d, int geom_type1, int geom_type2, int func); /* mjb_register_collision */
void mjo_set_geom_size(const mjb_model_desc *m, mjo_data *d, const double *size);   /* [ngeom][3]; NULL: back to the model's */
void mjo_set_geom_type(const mjb_model_desc *m, mjo_data *d, const int *type);      /* [ngeom];    NULL: back to the model's */
unsigned long long mjo_warning(const mjo_data *d, int which); /* mjData.warning[which].number */
void mjo_energy(const mjb_model_desc *m, mjo_data *d);          /* mj_energyPos + mj_energyVel (mjENBL_ENERGY) */
void mjo_tendon(const mjb_model_desc *m, mjo_data *d);
void mjo_tendon_vel(const mjb_model_desc *m, mjo_data *d);
void mjo_ctrl_noise(const mjb_model_desc *m, mjo_data *d, double noise_std, double noise_rate, uint64_t seed,
                    uint64_t env, uint32_t step);

/* Roll out `nsteps` steps for envs [0,nenv) starting from env-major host arrays qpos/qvel (updated in
 * place), with OU ctrl noise as above (std==0: ctrl taken from `ctrl` [nenv][nu], held constant).
 * Uses `nthreads` OS threads (one env per thread at a time).  Returns 0.  For the cpu_baseline leg
 * of bench.py and for parity rollouts. */
int mjo_rollout(const mjb_model_desc *m, int nenv, int nsteps, double *qpos, double *qvel, const double *ctrl,
                double *sensordata, double noise_std, double noise_rate, uint64_t seed, int64_t env_offset,
                int nthreads);

#ifdef __cplusplus
}
#endif
#endif
