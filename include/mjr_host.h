/* mjr_host.h — C interface of the host runtime (libmjr_host.so): the batched, ROS-free mirror of the
 * reference's MujocoEnv / MujocoPlugin layer that drives the step engine of mjb.h.
 *
 * Two parts:
 *  1. `mjr_backend`: the stepper vtable MujocoEnv talks to.  The product implementation wraps libmjb
 *     (mjr_make_mjb_backend); it needs a HIP device and has no fallback.
 *  2. `mjr_env_*`: a flat C face of the C++ class mujoco_ros::MujocoEnv (host/mujoco_env.h) so that
 *     bindings and tests can drive it; each function names the reference member it forwards to
 *     (/root/reference mujoco_ros/src/mujoco_env.cpp, callbacks.cpp).
 */
#ifndef MJR_HOST_H_
#define MJR_HOST_H_

#include <stdint.h>

#include "mjb.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mjr_backend {
	void *self;
	int (*nenv)(void *self);
	int (*field_size)(void *self, int field);
	int (*step)(void *self, int nsteps);            /* nsteps fused full steps              */
	int (*step1)(void *self);                       /* up to the control-callback point     */
	int (*step2)(void *self);                       /* rest of the step + integration       */
	int (*forward)(void *self);                     /* mj_forward                           */
	int (*reset)(void *self, const uint8_t *mask);  /* mj_resetData (mask NULL: all)        */
	int (*get)(void *self, int field, int env_lo, int env_hi, double *host);
	int (*set)(void *self, int field, int env_lo, int env_hi, const double *host);
	int (*set_ctrl_noise)(void *self, double std, double rate, uint64_t seed, int64_t env_offset);
	int (*synchronize)(void *self);
	const char *(*last_error)(void *self);
	void (*destroy)(void *self);
	/* optional (NULL: the runtime falls back to one get / set per field): several fields per call, one synchronisation;
	 * page-locking of the runtime's host mirrors */
	int (*get_many)(void *self, int n, const int *fields, int env_lo, int env_hi, double *const *host);
	int (*set_many)(void *self, int n, const int *fields, int env_lo, int env_hi, const double *const *host);
	int (*host_register)(void *self, void *host, unsigned long long bytes);
	int (*host_unregister)(void *self, void *host);
	int (*step_async)(void *self, int nsteps); /* enqueue nsteps fused steps without waiting for them */
	int (*register_collision)(void *self, int geom_type1, int geom_type2, int func); /* mjb_register_collision */
	/* optional: per-env model parameters (mjb_set_env_*), what the reference's services change on its single mjModel
	 * (callbacks.cpp:210-370, 462-592, 641-884).  `what` = MJR_ENV_*; data for envs [env_lo, env_hi), env-major. */
	int (*set_env_param)(void *self, int what, int env_lo, int env_hi, const void *data);
	/* optional: several fields through ONE transfer (mjb_get_packed / mjb_set_packed): block = field after field, [env][dim] each */
	int (*get_packed)(void *self, int n, const int *fields, int env_lo, int env_hi, double *host_block);
	int (*set_packed)(void *self, int n, const int *fields, int env_lo, int env_hi, const double *host_block);
	/* optional: the split step for the callback envs [0, ncb) only, the rest taking the same step fused (mjb_step1_prefix /
	 * mjb_step_rest / mjb_step2_prefix); NULL: every env is split (step1 / step2) */
	int (*step1_prefix)(void *self, int ncb);
	int (*step_rest)(void *self, int ncb);
	int (*step2_prefix)(void *self, int ncb);
	/* optional: step2_prefix of the step in flight + step1_prefix of the next one as one launch (mjb_step21_prefix) */
	int (*step21_prefix)(void *self, int ncb);
	/* optional: the second half of an RK4 step cut at the callback points of its four evaluations (mjb_step2_rk_prefix): rk = 0, 1, 2
	 * finish evaluation rk and run the first half of evaluation rk + 1 for envs [0, ncb) -- ncb < 0: a backend without the prefix
	 * entry points, all envs --, rk = 3 finishes the step.  NULL: control / passive callbacks of an RK4 model fire once per step. */
	int (*step2_rk)(void *self, int ncb, int rk);
} mjr_backend;

enum {
	MJR_ENV_GRAVITY = 0,       /* double [n][3]                                                               */
	MJR_ENV_GEOM_FRICTION = 1, /* double [n][ngeom][3]                                                        */
	MJR_ENV_GEOM_SIZE = 2,     /* double [n][ngeom][3]                                                        */
	MJR_ENV_GEOM_TYPE = 3,     /* int    [n][ngeom]                                                           */
	MJR_ENV_EQUALITY = 4,      /* double [n][neq][19]: active | eq_data[11] | solref[2] | solimp[5]           */
	MJR_ENV_BODY_MASS = 5      /* double [n][nbody]; the backend derives what mj_setConst would (mjb_set_env_body_mass) */
};

/* creates a backend for (model, nenv, device); NULL on failure */
typedef mjr_backend *(*mjr_backend_factory)(const mjb_model_desc *desc, int nenv, int device, void *user);

/* The product backend: mjb_compile + mjb_make_batch on HIP device `device`.  NULL (and mjr_last_error())
 * when no GPU is usable — there is no CPU path. */
mjr_backend *mjr_make_mjb_backend(const mjb_model_desc *desc, int nenv, int device, void *unused);
const char *mjr_last_error(void);
int mjr_model_desc_size(void);  /* sizeof(mjb_model_desc) this library was built with */

/* ---- flat face of mujoco_ros::MujocoEnv ---- */
typedef struct mjr_env mjr_env;

/* Name tables (mj_name2id): arrays of NUL-terminated strings, counts taken from desc */
typedef struct mjr_names {
	const char *const *body;
	const char *const *joint;
	const char *const *geom;
	const char *const *site;
	const char *const *sensor;
	const char *const *actuator;
	const char *const *equality; /* may be NULL (then equalities cannot be addressed by name) */
	const char *const *tendon;   /* may be NULL */
} mjr_names;

/* MujocoEnv::MujocoEnv(admin_hash) (mujoco_env.cpp:68-161).  `params_json` is a JSON object holding the
 * private node parameters the constructor reads (eval_mode, unpause, num_steps, MujocoPlugins, ...).
 * NULL on failure (e.g. eval_mode without a hash: the reference throws std::runtime_error). */
mjr_env *mjr_env_create(const char *admin_hash, const char *params_json);
void mjr_env_destroy(mjr_env *e);
/* nh->setParam / deleteParam on the env's private parameter store */
int mjr_env_set_param(mjr_env *e, const char *key, const char *json_value);
int mjr_env_delete_param(mjr_env *e, const char *key);

/* queue a model (settings_.load_request = 2, main.cpp:150-151); factory NULL = the HIP engine */
int mjr_env_queue_model(mjr_env *e, const mjb_model_desc *desc, const mjr_names *names, int nenv, int device,
                        mjr_backend_factory factory, void *factory_user);
/* The same with the batch sharded over several GPUs of the node: env block i of ndev contiguous blocks lives on
 * devices[i] (a device may be listed more than once); one backend per block behind one composite mjr_backend, launches issued to
 * all blocks before any is waited for, the ctrl-noise stream keyed by the GLOBAL env index.  The model is replicated. */
int mjr_env_queue_model_devices(mjr_env *e, const mjb_model_desc *desc, const mjr_names *names, int nenv, const int *devices,
                                int ndev, mjr_backend_factory factory, void *factory_user);
int mjr_env_start(mjr_env *e);    /* startPhysicsLoop + startEventLoop (main.cpp:154-155) */
int mjr_env_shutdown(mjr_env *e); /* exit_request = 1, join both threads                   */

int mjr_env_operational_status(mjr_env *e); /* getOperationalStatus()        */
int mjr_env_pending_steps(mjr_env *e);      /* num_steps_until_exit_         */
int mjr_env_is_physics_running(mjr_env *e);
int mjr_env_is_event_running(mjr_env *e);
int mjr_env_model_valid(mjr_env *e);
const char *mjr_env_load_error(mjr_env *e);

int mjr_env_step(mjr_env *e, int num_steps, int blocking);           /* MujocoEnv::step -> 1 true / 0 false */
int mjr_env_toggle_paused(mjr_env *e, int paused, const char *hash); /* togglePaused                        */
/* Step action (callbacks.cpp:94-129): returns 1 success, 0 failed/preempted; *preempted set accordingly */
int mjr_env_step_goal(mjr_env *e, int num_steps, int *preempted);
int mjr_env_reset_request(mjr_env *e);   /* resetCB: reset_request = 1  */
int mjr_env_set_pause(mjr_env *e, int paused, const char *hash); /* setPauseCB -> success field */

/* atomics of settings_ (name in {"run","exit_request","load_request","reset_request","env_steps_request"}) */
int mjr_env_get_setting(mjr_env *e, const char *name);
int mjr_env_set_setting(mjr_env *e, const char *name, int value);
int mjr_env_set_ctrl_noise(mjr_env *e, double std, double rate);

double mjr_env_sim_time(mjr_env *e);  /* last published /clock value (stays 0 with use_sim_time = false, mujoco_env.cpp:701-703) */
double mjr_env_data_time(mjr_env *e); /* env 0's data_->time after the latest step: what the physics loop paces on */
unsigned long long mjr_env_step_count(mjr_env *e);
int mjr_env_nenv(mjr_env *e);
int mjr_env_name2id(mjr_env *e, int objtype, const char *name);
/* copy a field of one env between the host and the device (takes physics_thread_mutex_) */
int mjr_env_get_field(mjr_env *e, int field, int env, double *out);
int mjr_env_set_field(mjr_env *e, int field, int env, const double *in);

/* plugins */
int mjr_env_num_plugins(mjr_env *e);
int mjr_env_num_cb_ready_plugins(mjr_env *e);
/* flags of the i-th plugin if it is the built-in "mujoco_ros/TestPlugin" (test_plugin.h:62-73):
 * name in {"ran_reset","ran_control_cb","ran_passive_cb","ran_render_cb","ran_last_cb",
 * "ran_on_geom_changed_cb","got_config_param","got_lvl1_nested_array","got_lvl2_nested_array",
 * "got_lvl1_nested_struct","got_lvl2_nested_struct","should_fail","control_calls","last_env"};
 * -1 if unknown */
int mjr_env_test_plugin_flag(mjr_env *e, int i, const char *name, int clear);
int mjr_env_notify_geom_changed(mjr_env *e, int geom_id);
int mjr_env_set_callback_envs(mjr_env *e, int n);
/* MujocoEnv::registerCollisionFunction (mujoco_env.cpp:163-176) with a device-side pair function (MJB_COLFUNC_*) in place of
 * the host callback; returns 1 when an override of this pair type was already registered (the reference warns), 0 when it is
 * the first, -1 on error.  Overrides are dropped at the next reload (prepareReload, :950-954). */
int mjr_env_register_collision_function(mjr_env *e, int geom_type1, int geom_type2, int func);

/* ---- service handlers that read / change the model or one body's state (callbacks.cpp:177-201, 210-592, 641-884), per env
 * range [env_lo, env_hi) (env_hi < 0: every env -- what the reference does to its single model).  Each returns the service's
 * `success` (1 / 0; -1 on a bad call) and copies `status_message` into msg[msg_cap] when msg != NULL.  All of them apply the
 * eval-mode admin-hash gate of the reference.  Poses are {x, y, z, qw, qx, qy, qz}, twists {vx, vy, vz, wx, wy, wz}. */
typedef struct mjr_body_state {
	char name[64];
	double mass;
	double pose[7];
	char pose_frame[64];  /* "" or "world": anything else cannot be transformed (no tf here) and is refused as in the reference */
	double twist[6];
	char twist_frame[64];
} mjr_body_state;
int mjr_env_set_body_state(mjr_env *e, const mjr_body_state *state, int set_pose, int set_twist, int set_mass, int reset_qpos,
                           const char *admin_hash, int env_lo, int env_hi, char *msg, int msg_cap); /* setBodyStateCB :210-370 */
int mjr_env_get_body_state(mjr_env *e, const char *name, const char *admin_hash, int env, mjr_body_state *out, char *msg,
                           int msg_cap); /* getBodyStateCB :372-460 */
typedef struct mjr_geom_properties {
	char name[64];
	int type; /* mjtGeom */
	double body_mass, friction[3], size[3];
} mjr_geom_properties;
int mjr_env_set_geom_properties(mjr_env *e, const mjr_geom_properties *p, int set_type, int set_mass, int set_friction, int set_size,
                                const char *admin_hash, int env_lo, int env_hi, char *msg, int msg_cap); /* :508-592 */
int mjr_env_get_geom_properties(mjr_env *e, const char *geom_name, const char *admin_hash, int env, mjr_geom_properties *out,
                                char *msg, int msg_cap); /* :594-639 */
int mjr_env_set_gravity(mjr_env *e, const double *gravity3, const char *admin_hash, int env_lo, int env_hi, char *msg, int msg_cap); /* :462-484 */
int mjr_env_get_gravity(mjr_env *e, const char *admin_hash, int env, double *gravity3, char *msg, int msg_cap);                      /* :486-506 */
typedef struct mjr_eq_parameters {
	char name[64], element1[64], element2[64];
	int type, active; /* mjtEq: 0 connect, 1 weld, 2 joint, 3 tendon */
	double anchor[3], relpose[7], torquescale, polycoef[5];
	double dmin, dmax, width, midpoint, power, timeconst, dampratio; /* solverParameters */
} mjr_eq_parameters;
int mjr_env_set_eq_parameters(mjr_env *e, const mjr_eq_parameters *params, int n, const char *admin_hash, int env_lo, int env_hi,
                              char *msg, int msg_cap); /* setEqualityConstraintParametersArrayCB :748-780 */
/* fills out[k] for every name found (in request order); *nout = how many */
int mjr_env_get_eq_parameters(mjr_env *e, const char *const *names, int n, const char *admin_hash, int env, mjr_eq_parameters *out,
                              int *nout, char *msg, int msg_cap); /* getEqualityConstraintParametersArrayCB :862-897 */
/* reloadCB (:177-201): queue the model, wait until the loading request state is 0 again; success = model_valid, message = load_error_ */
int mjr_env_reload(mjr_env *e, const mjb_model_desc *desc, const mjr_names *names, int nenv, int device, mjr_backend_factory factory,
                   void *factory_user, char *msg, int msg_cap);
/* get_loading_request_state (:72-87): returns getOperationalStatus(); description "Sim ready" / "Loading in progress" / "Loading issued" */
int mjr_env_loading_request_state(mjr_env *e, char *description, int cap);
int mjr_env_load_initial_joint_states(mjr_env *e); /* load_initial_joint_states service (:66-71) */

/* ---- sensors plugin ("mujoco_ros_sensors/MujocoRosSensorsPlugin"): the typed records it would publish
 * (reference: mujoco_ros_sensors/src/mujoco_sensor_handler_plugin.cpp:175-436 lastStageCallback, :123-173
 * registerNoiseModelsCB).  kind: 0 ScalarStamped, 1 Vector3Stamped, 2 PointStamped, 3 QuaternionStamped. */
typedef struct mjr_sensor_record {
	char name[64], frame_id[64];
	int kind, env, has_truth;
	double stamp;
	float value[4], truth[4];
} mjr_sensor_record;
int mjr_sensors_num_records(mjr_env *e, int plugin, int env);                       /* -1: not a sensors plugin */
int mjr_sensors_get_record(mjr_env *e, int plugin, int env, int k, mjr_sensor_record *out);
/* 1 success, 0 refused (eval mode and wrong admin hash), -1 not a sensors plugin */
int mjr_sensors_register_noise(mjr_env *e, int plugin, const char *sensor_name, int set_flag, const double *mean,
                               const double *std, const char *admin_hash);

#ifdef __cplusplus
}
#endif
#endif
