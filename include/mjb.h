/* mjb.h — C-ABI of the MI355X-native batched MuJoCo-style step engine (libmjb.so).
 *
 * This is the drop-in boundary for the ONE hot path of ubi-agni/mujoco_ros_pkgs: the `mj_step` loop
 * inside `MujocoEnv::physicsLoop` (/root/reference mujoco_ros/src/mujoco_env.cpp:436-639).  In the
 * reference that path sits behind the MuJoCo 2.3.7 C API (un-vendored third-party library,
 * mujoco_ros/CMakeLists.txt:61).  Every entry point below names the MuJoCo call, and the reference
 * call site(s), it takes the place of for a batch of N independent env instances that share one
 * constant model.  Plain C: pointers and sizes only, no torch / HIP types in any signature.
 *
 * Error convention: MuJoCo aborts through mju_error; this ABI never aborts or throws.  Functions
 * returning int give 0 on success and a negative MJB_E* code on failure; functions returning a
 * pointer give NULL on failure; mjb_last_error() returns the message of the calling thread's last
 * failure.  The engine has NO CPU fallback: without a usable HIP device every compute entry point
 * fails with MJB_ENODEVICE.
 */
#ifndef MJB_H_
#define MJB_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MJB_VERSION 100

/* error codes */
#define MJB_OK 0
#define MJB_EINVAL (-1)    /* bad argument / inconsistent model */
#define MJB_ENODEVICE (-2) /* no HIP device / HIP runtime failure */
#define MJB_ENOMEM (-3)
#define MJB_EUNSUPPORTED (-4) /* model uses a feature the engine does not implement */
#define MJB_ERANGE (-5)

/* MuJoCo 2.3.7 enum values used in mjb_model_desc (mjtJoint, mjtGeom, ...) */
enum { MJB_JNT_FREE = 0, MJB_JNT_BALL = 1, MJB_JNT_SLIDE = 2, MJB_JNT_HINGE = 3 };
enum { MJB_GEOM_PLANE = 0, MJB_GEOM_SPHERE = 2, MJB_GEOM_CAPSULE = 3, MJB_GEOM_BOX = 6 };
enum { MJB_INT_EULER = 0, MJB_INT_RK4 = 1, MJB_INT_IMPLICIT = 2, MJB_INT_IMPLICITFAST = 3 };
/* mjtIntegrator.  implicitfast: qacc = (M - h D)^-1 (qfrc_smooth + qfrc_constraint) with D = d qfrc_smooth / d qvel without the Coriolis terms
 * (mjd_passive_vel + mjd_actuator_vel) -- accepted when D is a model constant and diagonal: joint damping, and the velocity terms of affine
 * actuator biases on joint transmissions (a velocity term in an affine GAIN, or tendon damping, makes mjb_compile refuse it).  mjINT_IMPLICIT
 * (the Coriolis derivatives, an LU factor) is refused. */
enum { MJB_CONE_PYRAMIDAL = 0, MJB_CONE_ELLIPTIC = 1 };
enum { MJB_SOL_PGS = 0, MJB_SOL_CG = 1, MJB_SOL_NEWTON = 2 };
enum { MJB_GAIN_FIXED = 0, MJB_GAIN_AFFINE = 1 };
enum { MJB_BIAS_NONE = 0, MJB_BIAS_AFFINE = 1 };
enum { MJB_TRN_JOINT = 0, MJB_TRN_TENDON = 3 };  /* mjtTrn: joint and (fixed) tendon transmissions; jointinparent / slidercrank / site are refused */
enum { MJB_DYN_NONE = 0, MJB_DYN_INTEGRATOR = 1, MJB_DYN_FILTER = 2 };  /* mjtDyn (mjDYN_MUSCLE = 3, mjDYN_USER = 4 are refused) */
enum { /* mjtDisableBit */
	MJB_DSBL_CONSTRAINT = 1 << 0, MJB_DSBL_EQUALITY = 1 << 1, MJB_DSBL_FRICTIONLOSS = 1 << 2,
	MJB_DSBL_LIMIT = 1 << 3, MJB_DSBL_CONTACT = 1 << 4, MJB_DSBL_PASSIVE = 1 << 5,
	MJB_DSBL_GRAVITY = 1 << 6, MJB_DSBL_CLAMPCTRL = 1 << 7, MJB_DSBL_WARMSTART = 1 << 8,
	MJB_DSBL_FILTERPARENT = 1 << 9, MJB_DSBL_ACTUATION = 1 << 10, MJB_DSBL_REFSAFE = 1 << 11,
	MJB_DSBL_SENSOR = 1 << 12, MJB_DSBL_EULERDAMP = 1 << 14
};
enum { MJB_ENBL_ENERGY = 1 << 1 }; /* mjtEnableBit (the only one implemented) */
enum { /* mjtWarning: index of mjData.warning[] -- see mjb_warning() */
	MJB_WARN_INERTIA = 0, MJB_WARN_CONTACTFULL = 1, MJB_WARN_CNSTRFULL = 2, MJB_WARN_VGEOMFULL = 3, MJB_WARN_BADQPOS = 4,
	MJB_WARN_BADQVEL = 5, MJB_WARN_BADQACC = 6, MJB_WARN_BADCTRL = 7, MJB_NWARNING = 8
};
enum { /* mjtObj */
	MJB_OBJ_UNKNOWN = 0, MJB_OBJ_BODY = 1, MJB_OBJ_XBODY = 2, MJB_OBJ_JOINT = 3, MJB_OBJ_GEOM = 5,
	MJB_OBJ_SITE = 6, MJB_OBJ_ACTUATOR = 18
};
enum { /* mjtSensor (subset implemented; values are MuJoCo's) */
	MJB_SENS_TOUCH = 0, MJB_SENS_ACCELEROMETER = 1, MJB_SENS_VELOCIMETER = 2, MJB_SENS_GYRO = 3, MJB_SENS_FORCE = 4,
	MJB_SENS_TORQUE = 5, MJB_SENS_MAGNETOMETER = 6, MJB_SENS_RANGEFINDER = 7, MJB_SENS_TENDONPOS = 10, MJB_SENS_TENDONVEL = 11,
	MJB_SENS_JOINTPOS = 8, MJB_SENS_JOINTVEL = 9, MJB_SENS_ACTUATORPOS = 12, MJB_SENS_ACTUATORVEL = 13,
	MJB_SENS_ACTUATORFRC = 14, MJB_SENS_BALLQUAT = 15, MJB_SENS_BALLANGVEL = 16, MJB_SENS_FRAMEPOS = 23,
	MJB_SENS_FRAMEQUAT = 24, MJB_SENS_FRAMEXAXIS = 25, MJB_SENS_FRAMEYAXIS = 26, MJB_SENS_FRAMEZAXIS = 27,
	MJB_SENS_FRAMELINVEL = 28, MJB_SENS_FRAMEANGVEL = 29, MJB_SENS_FRAMELINACC = 30, MJB_SENS_FRAMEANGACC = 31,
	MJB_SENS_SUBTREECOM = 32, MJB_SENS_CLOCK = 35,
	/* round 6: the rest of the scalar / 3-vector types MujocoRosSensorsPlugin serialises (mujoco_ros_sensors/src/mujoco_sensor_handler_plugin.cpp:76-105) */
	MJB_SENS_JOINTLIMITPOS = 17, MJB_SENS_JOINTLIMITVEL = 18, MJB_SENS_JOINTLIMITFRC = 19, MJB_SENS_TENDONLIMITPOS = 20,
	MJB_SENS_TENDONLIMITVEL = 21, MJB_SENS_TENDONLIMITFRC = 22, MJB_SENS_SUBTREELINVEL = 33, MJB_SENS_SUBTREEANGMOM = 34,
	/* (mjSENS_JOINTACTFRC's value in the pinned MuJoCo release cannot be read here -- the library and its headers are absent -- so the engine uses a
	 *  value outside mjtSensor's range; a binding that fills mjb_model_desc from a real mjModel maps mjSENS_JOINTACTFRC onto it) */
	MJB_SENS_JOINTACTFRC = 38
};
enum { MJB_STAGE_NONE = 0, MJB_STAGE_POS = 1, MJB_STAGE_VEL = 2, MJB_STAGE_ACC = 3 };
enum { MJB_EQ_CONNECT = 0, MJB_EQ_WELD = 1, MJB_EQ_JOINT = 2, MJB_EQ_TENDON = 3 }; /* mjtEq (distance not implemented) */
enum { /* mjtConstraint */
	MJB_CNSTR_EQUALITY = 0, MJB_CNSTR_FRICTION_DOF = 1, MJB_CNSTR_FRICTION_TENDON = 2, MJB_CNSTR_LIMIT_JOINT = 3, MJB_CNSTR_LIMIT_TENDON = 4, MJB_CNSTR_CONTACT_FRICTIONLESS = 5, MJB_CNSTR_CONTACT_PYRAMIDAL = 6,
	MJB_CNSTR_CONTACT_ELLIPTIC = 7
};

/* ---- model description: the mjModel arrays the path reads (see mjb_model_fields.def) ---- */
typedef struct mjb_model_desc {
#define MJB_SIZE(name) int name;
#define MJB_OPT_I(name) int name;
#define MJB_OPT_D(name, n) double name[n];
#define MJB_ARR_I(name, rows, cols) const int *name;
#define MJB_ARR_D(name, rows, cols) const double *name;
#include "mjb_model_fields.def"
#undef MJB_SIZE
#undef MJB_OPT_I
#undef MJB_OPT_D
#undef MJB_ARR_I
#undef MJB_ARR_D
} mjb_model_desc;

/* ---- per-env data field ids (order of mjb_data_fields.def) ---- */
typedef enum mjb_field {
#define MJB_DS(name, rows, cols) MJB_F_##name,
#define MJB_DD(name, rows, cols) MJB_F_##name,
#define MJB_DD2(name, rows, cols) MJB_F_##name,
#define MJB_DI(name, rows, cols) MJB_F_##name,
#include "mjb_data_fields.def"
#undef MJB_DS
#undef MJB_DD
#undef MJB_DD2
#undef MJB_DI
	MJB_F_COUNT
} mjb_field;

typedef struct mjb_model mjb_model; /* compiled, immutable model (host copy + device blob) */
typedef struct mjb_batch mjb_batch; /* N env instances on one GPU */

/* Message of the calling thread's last failed call ("" if none). */
const char *mjb_last_error(void);
/* sizeof(mjb_model_desc) of the build: a binding generated from another revision of include/mjb_model_fields.def finds out here, not in a crash. */
int mjb_model_desc_size(void);
int mjb_version(void);
/* Number of usable HIP devices (0 when there is none; never fails). */
int mjb_device_count(void);

/* Validate + copy a model description.  Takes the place of `mj_loadXML`/`mj_loadModel` for the
 * batched path (reference: mujoco_env.cpp:836-843); the arrays are copied, `desc` may be freed. */
mjb_model *mjb_compile(const mjb_model_desc *desc);
void mjb_free_model(mjb_model *m); /* mj_deleteModel, mujoco_env.cpp:747 */

/* Dimension (doubles or ints per env) of a data field for this model; <0 on error. */
int mjb_field_size(const mjb_model *m, int field);
/* 1 if the field is an int field (MJB_DI), 0 if double. */
int mjb_field_is_int(int field);
/* 1 if the field is persistent state (MJB_DS). */
int mjb_field_is_state(int field);
const char *mjb_field_name(int field);
/* Doubles in one per-env LDS frame (all double fields) — layout documented in DESIGN.md. */
int mjb_frame_doubles(const mjb_model *m);
/* Bytes of LDS one env occupies (doubles + ints): fused != 0 -> the compact frame of mjb_step, else the full frame of
 * mjb_forward / mjb_step1 / mjb_step2.  Resident envs per CU = floor(160 KiB / that).
 * fused == 2: the WIDE fused frame of a Newton model with more than 128 rows of capacity (up to 128 rows of efc_J and of every per-row
 * array in LDS instead of 64, two rows per lane: the largest multiple of four >= 96 with which THREE frames share a CU -- 112 on the
 * Shadow-Hand-like model -- else 128 at two per CU; == the default fused frame for every other model).  A batch switches its long fused launches
 * to it when more than a quarter of the previous launch's env-steps had more than 64 rows, and back below 5 % (MJB_WIDE_FRAME=0 / 1
 * pins the choice); mjb_fused_frame(batch) tells which one the last launch ran on. */
int mjb_frame_bytes(const mjb_model *m, int fused);
/* Diagnostic: offset (doubles; ints for int fields; -1 = absent) of data field `field` in the fused / full frame. */
int mjb_frame_offset(const mjb_model *m, int field, int fused);

/* Allocate N env instances on HIP device `device`, all at the reset state (qpos = qpos0).
 * Takes the place of `mj_makeData` (mujoco_env.cpp:872), once per env. */
mjb_batch *mjb_make_batch(const mjb_model *m, int nenv, int device);
void mjb_free_batch(mjb_batch *b); /* mj_deleteData, mujoco_env.cpp:748 */
int mjb_nenv(const mjb_batch *b);

/* Launch geometry: lanes of a wavefront that cooperate on one env (8,16,32,64) and envs per
 * workgroup.  0 selects the engine default for the model.  Results do not depend on it. */
int mjb_set_launch(mjb_batch *b, int lanes_per_env, int envs_per_block);

/* Advance every env by `nsteps` full steps, fused in ONE kernel launch with the state held in LDS
 * between steps.  Takes the place of `nsteps` x `mj_step(model, data)` per env
 * (mujoco_env.cpp:498, :552, :593).  Asynchronous on the batch's stream. */
int mjb_step(mjb_batch *b, int nsteps);

/* Split step for host callbacks.  mjb_step1 runs position + velocity stages (everything `mj_step`
 * does before it invokes `mjcb_control`, incl. passive forces) and leaves the full frame in the
 * HBM workspace; the caller may then read/modify ctrl, qfrc_applied, xfrc_applied, qfrc_passive
 * (mjb_get/mjb_set) — this is where controlCallback / passiveCallback run (mujoco_env.h:242-251) —
 * and mjb_step2 finishes the step (actuation, acceleration, constraint solve, acc sensors,
 * integration).  step1+step2 == one mjb_step(b,1). */
int mjb_step1(mjb_batch *b);
int mjb_step2(mjb_batch *b);
/* The split for a PREFIX of the batch only -- the envs that have host callbacks (MujocoEnv's callback set): envs [0, ncb) are
 * stepped in two halves around the control-callback point, envs [ncb, nenv) take the same step as ONE fused launch.  Every env's
 * arithmetic is independent of the launch that carries it (same Philox key, same kernels): the result equals a whole-batch step.
 * One step = mjb_step1_prefix(ncb) -> copy the callback envs' fields out -> mjb_step_rest(ncb) [asynchronous: it runs while the
 * host callbacks do] -> copy their writes back -> mjb_step2_prefix(ncb) [advances the step counter; issues the rest itself if
 * the caller skipped mjb_step_rest].  Derived fields are readable for envs [0, ncb) only between the two halves.
 * Abandoning a split step: before mjb_step_rest, a new mjb_step1_prefix simply restarts it; after mjb_step_rest the other envs have
 * taken the step already, so only its second half (mjb_step2_prefix / mjb_step21_prefix / mjb_step2_rk_prefix) or a whole-batch
 * mjb_reset / mjb_step may follow -- mjb_step1_prefix returns MJB_EINVAL. */
int mjb_step1_prefix(mjb_batch *b, int ncb);
int mjb_step_rest(mjb_batch *b, int ncb);
int mjb_step2_prefix(mjb_batch *b, int ncb);
/* mjb_step2_prefix(ncb) of the split step in flight and mjb_step1_prefix(ncb) of the NEXT step as one launch for the callback envs
 * (one kernel and one device -> host round trip per step instead of two when steps follow each other: what a caller with control /
 * passive callbacks only needs -- the state it can read afterwards is the finished step's, the derived fields and the pos / vel
 * stage sensordata already the next step's).  Same results as the two calls it replaces: bit for bit when both run the same
 * kernel (every constrained model, and unconstrained ones outside the dense kernels' reach); for a model the two halves run through
 * the register-dense kernels (nv, nbody, nu, njnt <= 16, Euler, no constraint rows -- BASELINE config 2) the chained launch runs the
 * generic kernel instead, whose factor / solve sum in a different order: equal to rounding (tests/test_gpu_fused_consistency.py
 * bounds it at 1e-9 over a burst), not to the last bit. */
int mjb_step21_prefix(mjb_batch *b, int ncb);
/* The second half of an RK4 step (<option integrator="RK4">) cut at the callback points of its four evaluations: MuJoCo's
 * mj_RungeKutta evaluates mj_forwardSkip -- and with it mjcb_passive / mjcb_control -- once per evaluation (the reason the
 * reference has lastStageCallback at all, mujoco_ros/include/mujoco_ros/plugin_utils.h:119-125).  After mjb_step1_prefix and the
 * callbacks of evaluation 0: rk = 0, 1, 2 finish evaluation rk and run the first half of evaluation rk + 1 (its view, at
 * time = t0 + c h, is readable for the callback envs afterwards); rk = 3 finishes the step (advances the step counter).
 * Four calls in a row == mjb_step2_prefix of the same model, bit for bit (tests/test_rk4.py). */
int mjb_step2_rk_prefix(mjb_batch *b, int ncb, int rk);

/* Recompute all derived quantities without integrating (mj_forward: mujoco_env.cpp:329, :621;
 * callbacks.cpp:573) and leave the full frame in the HBM workspace for mjb_get. */
int mjb_forward(mjb_batch *b);

/* Reset envs with mask[i] != 0 (mask == NULL: all) to qpos0 / zero velocity, activation, control,
 * applied forces, warmstart, time (mj_resetData: mujoco_env.cpp:252). */
int mjb_reset(mjb_batch *b, const uint8_t *mask);

/* Copy a field for envs [env_lo, env_hi) between host memory (env-major, field_size per env) and
 * the device.  Derived fields are readable after mjb_forward / mjb_step1 / mjb_step2 (they come
 * from the frame workspace); only state fields and the callback-writable force fields can be set.
 * Synchronous. */
int mjb_get(mjb_batch *b, int field, int env_lo, int env_hi, double *host);
int mjb_set(mjb_batch *b, int field, int env_lo, int env_hi, const double *host);
int mjb_get_int(mjb_batch *b, int field, int env_lo, int env_hi, int *host);
/* The same for several double fields at once: every copy is enqueued asynchronously on the batch's stream and the call
 * synchronises ONCE (mjb_get_many) or not at all (mjb_set_many: ordered before the next launch).  This is what the host
 * runtime moves around a plugin callback round -- the mjData view fields of SURVEY.md 8a row T1 -- instead of one blocking
 * copy per field.  host[k] has the layout mjb_get / mjb_set use for fields[k].  Page-locking the host buffers
 * (mjb_host_register) makes the copies true DMA transfers. */
int mjb_get_many(mjb_batch *b, int n, const int *fields, int env_lo, int env_hi, double *const *host);
int mjb_set_many(mjb_batch *b, int n, const int *fields, int env_lo, int env_hi, const double *const *host);
/* The same through ONE transfer: a device kernel gathers the fields of envs [env_lo, env_hi) into one block laid out field after
 * field ([env][dim] each, in the order given; zero-sized fields take no room) and a single copy moves it -- what the callback
 * rounds of the host runtime use (a split step moves ~25 fields out and ~8 back; one copy per field costs more than the step).
 * mjb_get_packed synchronises; mjb_set_packed is asynchronous on the batch's stream (host_block must stay untouched until the
 * next synchronising call).  At most 40 fields. */
int mjb_get_packed(mjb_batch *b, int n, const int *fields, int env_lo, int env_hi, double *host_block);
int mjb_set_packed(mjb_batch *b, int n, const int *fields, int env_lo, int env_hi, const double *host_block);
int mjb_host_register(void *host, unsigned long long bytes);   /* hipHostRegister; 0 on success */
int mjb_host_unregister(void *host);

/* Raw HBM pointer of a state field's env-major array [nenv][field_size] (for RCCL gathers and
 * zero-copy tensor wrappers).  NULL for derived fields. */
void *mjb_device_ptr(mjb_batch *b, int field);

/* Device-side control noise, the reference's Ornstein-Uhlenbeck injector (mujoco_env.cpp:469-481):
 * before every step  noise = rate*noise + scale*N(0,1),  ctrl = noise  with
 * rate = exp(-dt/max(ctrl_noise_rate, mjMINVAL)), scale = ctrl_noise_std*sqrt(1-rate^2).
 * N(0,1) comes from a counter-based Philox-4x32-10 stream keyed (seed, global env id, step,
 * actuator), so a CPU oracle regenerates the identical sequence.  std == 0 disables it.
 * `env_offset` is the global index of this batch's env 0 (multi-GPU sharding). */
int mjb_set_ctrl_noise(mjb_batch *b, double ctrl_noise_std, double ctrl_noise_rate, uint64_t seed,
                       int64_t env_offset);
/* How the LAST fused mjb_step launch obtained the injector's normals (the values are identical in all three): 0 = generated inside the
 * step kernel, 1 = by mjb_noise_kernel ahead of the step kernel on the same stream, 2 = by mjb_noise_kernel on a side stream while
 * the previous launch ran (unconstrained kernels only; budget MJB_NOISE_PREGEN_MB).  Measurement aid, no reference counterpart. */
int mjb_noise_mode(const mjb_batch *b);
/* 1 = the default fused frame, 2 = the wide one (mjb_frame_bytes): what the batch's fused launches currently run on. */
int mjb_fused_frame(const mjb_batch *b);
/* The lane = env form of the unconstrained fused step (csrc/mjb_lane_env.hip): one env per LANE, the state of 64 envs in a
 * wavefront's registers, compiled per model topology (csrc/lane_env_topos.h).  Same step (mj_step, mujoco_env.cpp:498,552,593),
 * results equal to the generic kernels' to rounding (not bit for bit: a batch that mixes fused launches with split steps
 * should pin one form); measured on MI355X it is ahead of the 16-lanes-per-env kernel from 4096 envs (276 vs 227 M env-steps/s) and 12x
 * ahead at 65 536.  mode: -1 = automatic (fused launches over >= MJB_LANE_ENV_MIN_ENVS envs, default 4096 -- the whole batch, or the non-callback envs of a
 * split step, mjb_step_rest -- of a model whose
 * topology is compiled in, with no per-env model overrides / hwsim stage / xfrc_applied), 0 = never, 1 = whenever eligible.
 * The environment variable MJB_LANE_ENV (same values) sets the default of new batches (read by mjb_make_batch).  While mjb_set_stats is
 * counting, fused launches run the generic kernels whatever the mode (the counters live in those).  No reference counterpart. */
int mjb_set_lane_env(mjb_batch *b, int mode);
/* The lane = env kernel evaluates the sensor stages (A15) at the LAST step of a fused launch: sensordata is an output of the launch and nothing inside
 * it reads the values (the generic kernels evaluate them at every step into the LDS frame).  on = 1 makes it evaluate -- and store -- them at every
 * step: same results after the launch, and the per-step cost of A15 on that kernel becomes measurable (bench.py: other_configs.2_sensors_every_step). */
int mjb_set_sensors_every_step(mjb_batch *b, int on);
/* The SPLIT step of plain-PGS models (csrc/mjb_smooth_kernel.h + mjb_cstep_kernel in csrc/mjb_step.hip): per step, the smooth stages of mj_step
 * (kinematics .. qacc_smooth) run one env per LANE and hand geom frames, cdof, both L'DL factors,
 * qfrc_smooth and qacc_smooth to the constraint stages (A4 - A7, A13, A16) through a per-env record in HBM; those run one env per wavefront.  The
 * batch is cut into slices (MJB_SPLIT_SLICES, default 2) that alternate the two kernels on streams of their own.  Same step (mj_step, mujoco_env.cpp:498,552,593), results equal
 * to the fused kernel's to rounding.  Eligible: a model whose topology is compiled in (csrc/smooth_topos.h: free / ball / hinge / slide joints, one
 * per body), PGS with pyramidal or frictionless contacts, nv <= 16, Euler, no tendons / equalities / mocap bodies / activations, position- and
 * velocity-stage sensors of the lane = env list only, no per-env gravity / mass overrides, no hwsim stage, no xfrc_applied, no frame dump.
 * mode: 1 = whenever eligible, 0 = never, -1 (default) = only when the environment variable MJB_SPLIT_MIN_ENVS names a batch size (whole-batch
 * fused launches of at least that many envs).  A launch pair per step ends with its slice's slowest env, which only averages out over many envs per
 * slice: measured on config 3 (profiles/r06_split_step.txt) 33.2 M env-steps/s against the fused kernel's 27.5 M at 32 768 envs in the rollout's
 * light-contact phase, 26.2 against 23.3 M over steps 1000 - 4000, 23.4 against 24.6 M over steps 500 - 2000, 18.7 against 23.3 M at 4096 envs.
 * sensordata is that of the launch's LAST step (as in the lane = env kernel).  No reference counterpart. */
int mjb_set_split_step(mjb_batch *b, int mode);
/* >= 0: index of the compiled-in topology of the split step the batch's model matches, -1: none.  *used_last (may be NULL) = 1 when the last fused
 * launch ran as a split step, *slices (may be NULL) = the env slices it was cut into. */
int mjb_split_step_info(const mjb_batch *b, int *used_last, int *slices);
int mjb_model_split_step(const mjb_model *m);
/* >= 0: index of the compiled-in topology the batch's model matches; -2: none compiled in, but the model's structure fits the kernel -- its
 * first eligible launch builds the kernel for it through hiprtc (libhiprtc.so and csrc/mjb_lane_env_kernel.h next to libmjb.so; a few
 * seconds, cached per process); -3: that build was not possible (mjb_lane_env_error says why) and the generic kernels run; -1: the model does not
 * fit (constraint rows, free / ball joints, RK4, ...).  *used_last (may be NULL) = 1 when the last fused launch ran the lane = env kernel. */
int mjb_lane_env_info(const mjb_batch *b, int *used_last);
/* hiprtc builds of the lane = env kernel are kept on disk: $MJB_JIT_CACHE (default $XDG_CACHE_HOME/mjb_jit or ~/.cache/mjb_jit; "0" = off), one code
 * object per (gfx arch, kernel-header fingerprint, LDS budget, form, topology).  Counts of this process: kernels compiled / taken from the cache. */
void mjb_lane_env_jit_counts(int *compiled, int *disk_hits);
/* The same classification for a compiled model, without a batch or a device (>= 0 / -2 / -1 as above). */
int mjb_model_lane_env(const mjb_model *m);
const char *mjb_lane_env_error(void);
/* The lane = env kernel's FORM, process-wide: how many wavefronts share the 64 envs of a block.  0 = one (the whole step in one instruction stream);
 * 1 = two, the step's position half (poses, cinert, composite inertias, qM, factors, solves, Euler) and velocity half (velocities, forces,
 * qfrc_smooth) side by side on two SIMDs of a CU, both computing the poses; 2 = two, PIPELINED: one wavefront computes every pose once and
 * hands it on body by body through LDS (a workgroup barrier per body), the other follows one body behind with cinert, cdof, velocities and
 * forces and hands cdof / cinert / qfrc_smooth back -- nothing is computed twice.  A lone wavefront on a SIMD issues one instruction every ~4
 * cycles whatever it is, so while the batch leaves SIMDs idle the step's length is the longest instruction stream; 3 = THREE: the first wavefront runs
 * the pose chain and nothing else, the second follows it with cinert and cdof and then takes the composite inertias, qM, the factors, the solves and
 * Euler, the third the velocities and forces.  Form 3 runs whenever a block has a CU's LDS to itself (<= 64 x CUs envs: 16 384 on MI355X; form 2 is
 * the same with two wavefronts, by request only), form 1 up to twice that, form 0 beyond.  -1 = that rule (default; the environment
 * variable MJB_LANE_ENV_DUO = 0 / 1 / 2 / 3 overrides it).  A forced form that does not fit (LDS) falls back to the next lower one.  Results of the
 * four forms agree to rounding.  Returns the previous setting.  mjb_lane_env_last_form: the form of this process's last lane = env launch (-1: none yet).
 * Measurement / test knob, no reference counterpart. */
int mjb_lane_env_set_form(int form);
int mjb_lane_env_last_form(void);

/* Stream control: the hipStream_t (as void*) kernels are launched on; default is a stream the
 * batch owns.  mjb_synchronize waits for it. */
void *mjb_get_stream(mjb_batch *b);
int mjb_set_stream(mjb_batch *b, void *hip_stream);
int mjb_synchronize(mjb_batch *b);

/* Number of env auto-resets so far (MuJoCo's mj_checkPos / mj_checkVel / mj_checkAcc warnings: a state
 * that went NaN or beyond mjMAXVAL is reset to qpos0 exactly as mj_step does). */
int mjb_warning_count(mjb_batch *b, unsigned long long *count);
/* mjData.warning[which].number summed over the batch's envs (which = MJB_WARN_*):
 *   BADQPOS / BADQVEL / BADQACC  mj_checkPos / mj_checkVel / mj_checkAcc found NaN or |x| > mjMAXVAL and reset the env
 *                                (mjb_warning_count is their sum);
 *   CONTACTFULL                  an env produced more contacts than nconmax in one step: the contacts past the capacity,
 *                                in pair order, were dropped (mj_addContact's rule);
 *   CNSTRFULL                    an env's constraint rows exceeded nefcmax in one step: the first item (equality, friction
 *                                row, limit, contact -- MuJoCo's row order) that did not fit and every item after it were
 *                                dropped.  (MuJoCo 2.3.7 sizes its rows from the arena and drops ALL rows when that fails;
 *                                with a fixed per-env capacity the engine keeps the prefix that fits -- the oracle applies the
 *                                same rule.)
 * The other entries stay 0. */
int mjb_warning(mjb_batch *b, int which, unsigned long long *count);

/* Workload statistics of the constrained kernels, accumulated on the device by every forward evaluation that reaches the constraint
 * solver (one per env-step under Euler): what the workload actually asks of the solver.  Off by default; mjb_set_stats(b, 1) allocates
 * and ZEROES the counters, mjb_set_stats(b, 0) stops counting.  Layout of out[MJB_NSTATS] (mjb_get_stats):
 *   [0] evaluations counted   [1] sum of mjData.solver_iter (PGS sweeps / Newton / CG iterations)
 *   [2 + r], r = 0..256       histogram of nefc (rows of the evaluation; the last bin collects r >= 256)
 *   [259 + c], c = 0..128     histogram of ncon
 * Measurement aid (bench.py's ncon / nefc fields); MuJoCo's counterpart is reading d->ncon, d->nefc, d->solver_iter after mj_step. */
enum { MJB_NSTATS = 2 + 257 + 129 };
int mjb_set_stats(mjb_batch *b, int on);
int mjb_get_stats(mjb_batch *b, unsigned long long *out);

/* Aggregate metrics of the batch (the <= 16-double vector the RCCL all-reduce carries), computed on the
 * device from the state arrays on the batch's stream:
 *   out[0..7]  additive:  env-steps taken, auto-resets (BADQPOS + BADQVEL + BADQACC), CONTACTFULL, CNSTRFULL,
 *                         sum of potential energy, sum of kinetic energy (both 0 unless mjENBL_ENERGY), nenv, 0
 *   out[8..15] maxima:    max |qacc|, max |qvel|, max time, 0 ...
 * so that a job-wide vector is one SUM all-reduce of the first half and one MAX all-reduce of the second.
 * mjb_metrics copies the vector to the host (synchronous); mjb_metrics_device enqueues the reduction and returns the
 * device pointer of the 16 doubles (valid until the next call; ordered on the batch's stream). */
int mjb_metrics(mjb_batch *b, double *out16);
void *mjb_metrics_device(mjb_batch *b);

/* mjData after mj_step holds the derived quantities of the last forward pass (xpos, contacts, efc_*, ...), which
 * the reference's lastStageCallback / renderCallback read (mujoco_env.cpp:506-515, 430-436).  Fused mjb_step
 * keeps them in LDS only; with keep_frame on, every launch also stores the full frame of its LAST step so that
 * mjb_get / mjb_get_int serve derived fields afterwards (costs the compact LDS layout and one frame store per
 * launch).  Off by default. */
int mjb_set_keep_frame(mjb_batch *b, int on);

/* ---- sensors-plugin equivalent ----
 * What MujocoRosSensorsPlugin::lastStageCallback publishes from sensordata after a step
 * (/root/reference mujoco_ros_sensors/src/mujoco_sensor_handler_plugin.cpp:175-437), for every env at once, on the
 * device, as float32 in sensordata layout ([nenv][nsensordata], quaternions stay (w, x, y, z)):
 *   which = 1 "ground truth" = sensordata / cutoff           (the plugin's gt publisher)
 *   which = 0 "value"        = the same when the sensor has no noise model, else sensordata + noise / cutoff
 *                              (scalars / vectors; per-axis N(mean, sigma)) or setRPY(noise) * q (quaternions).
 * mjb_sensor_set_noise is registerNoiseModelsCB (:123-173): bit k of set_flag enables noise on component k, the
 * n-th set bit reads mean3[n] / sigma3[n], flags accumulate; set_flag = 0 clears the sensor's model (extension).
 * The plugin's std::mt19937 is replaced by the engine's counter-based Philox stream keyed
 * (seed, global env, steps taken so far, component), so values are reproducible and identical on CPU and GPU. */
int mjb_sensor_set_noise(mjb_batch *b, int sensor, int set_flag, const double *mean3, const double *sigma3);
int mjb_sensor_pack(mjb_batch *b, uint64_t seed);
int mjb_sensor_get(mjb_batch *b, int which, int env_lo, int env_hi, float *host);
void *mjb_sensor_device_ptr(mjb_batch *b, int which);

/* ---- collision-function overrides (MujocoEnv::registerCollisionFunction, /root/reference mujoco_ros/src/mujoco_env.cpp:163-176) ----
 * The reference lets a plugin replace MuJoCo's pair function for a geom-type pair (the mjCOLLISIONFUNC table) with a host
 * callback until the next reload (:950-954).  Host callbacks cannot run inside the fused device step, so the override names
 * one of the engine's device-side pair functions instead:
 *   MJB_COLFUNC_DEFAULT  the built-in primitive function of the pair (restores the default)
 *   MJB_COLFUNC_NONE     the pair type produces no contacts (a collision function that returns 0)
 *   MJB_COLFUNC_SPHERES  both geoms are replaced by their bounding spheres (planes stay planes): at most one contact
 * geom_type1 / geom_type2 are mjtGeom values in either order; the override applies to every candidate pair of those types of
 * this batch from the next launch on.  As in the reference, whose mjCOLLISIONFUNC table is indexed by the geoms' CURRENT types, an env whose
 * geom types were changed with mjb_set_env_geom_type gets, for each candidate pair, the override registered for the pair's types in THAT env. */
enum { MJB_COLFUNC_DEFAULT = 0, MJB_COLFUNC_NONE = 1, MJB_COLFUNC_SPHERES = 2 };
int mjb_register_collision(mjb_batch *b, int geom_type1, int geom_type2, int func);

/* ---- per-env model parameters (the subset that needs no mj_setConst) ----
 * The reference mutates its single mjModel through services (setGravity, setGeomProperties friction:
 * /root/reference mujoco_ros/src/callbacks.cpp:462-592, 641-884); in a batch every env may carry its own value (domain
 * randomisation).  Envs never written keep the model's value.  gravity: [env][3]; friction: [env][ngeom][3].
 * (masses: mjb_set_env_mass_params below.)  A <contact><pair> that states its friction keeps it: mjModel.pair_friction is a compiled constant
 * the geoms' frictions do not reach (collpair_param, include/mjb_model_fields.def). */
int mjb_set_env_gravity(mjb_batch *b, int env_lo, int env_hi, const double *gravity);
int mjb_set_env_geom_friction(mjb_batch *b, int env_lo, int env_hi, const double *friction);
/* setGeomProperties' set_size / set_type (callbacks.cpp:555-575) per env: size [env][ngeom][3], type [env][ngeom] (mjtGeom: plane,
 * sphere, capsule, box; ellipsoid / cylinder are accepted and yield no contacts).  As in the reference the bounding radii are NOT recomputed ("AABBs are not recomputed", :557-560) and the
 * candidate pair list stays the model's; a pair whose new types have no pair function yields no contacts. */
int mjb_set_env_geom_size(mjb_batch *b, int env_lo, int env_hi, const double *size);
int mjb_set_env_geom_type(mjb_batch *b, int env_lo, int env_hi, const int *type);
/* setEqualityConstraintParameters (/root/reference mujoco_ros/src/callbacks.cpp:641-884: active flag, solref, solimp and the
 * type's data -- anchor / relpose / torquescale / polycoef) per env: params[env][neq][19] =
 * { active (0 / 1), eq_data[11], solref[2], solimp[5] } for every equality of the model, in model order. */
int mjb_set_env_equality(mjb_batch *b, int env_lo, int env_hi, const double *params);
/* setBodyState's mass (callbacks.cpp:210-370) followed by mj_setConst (:251-256): the engine takes, per env, the masses AND
 * what mj_setConst derives from them -- the caller computes those (the reference has libmujoco: mj_setConst on a scratch
 * mjModel; mujoco_ros_pkgs_amd/engine.py does it with its own numpy dynamics).  params[env][mjb_env_mass_stride(model)] =
 * body_mass[nbody] | body_subtreemass[nbody] | body_inertia[nbody][3] | dof_invweight0[nv] | body_invweight0[nbody][2] |
 * tendon_invweight0[ntendon] | meaninertia.  (Batches carrying these overrides run the generic kernels.) */
int mjb_env_mass_stride(const mjb_model *m);
int mjb_set_env_mass_params(mjb_batch *b, int env_lo, int env_hi, const double *params);
/* The same without the caller having MuJoCo (or Python) at hand: mj_setConst's derivation is done here, host side, in plain C++
 * (body Jacobians at qpos0 -> joint-space inertia -> its inverse; mjb_api.hip).  mjb_derive_mass_params fills ONE packed block
 * (out[mjb_env_mass_stride(model)]) from body_mass[nbody] and body_inertia[nbody][3] (NULL: the model's principal inertias) and
 * needs no device; mjb_set_env_body_mass derives a block per env (body_mass[env][nbody], body_inertia[env][nbody][3] or NULL)
 * and uploads them (= callbacks.cpp:244-258: model_->body_mass[id] = mass; mj_setConst). */
int mjb_derive_mass_params(const mjb_model *m, const double *body_mass, const double *body_inertia, double *out);
int mjb_set_env_body_mass(mjb_batch *b, int env_lo, int env_hi, const double *body_mass, const double *body_inertia);

/* ---- device-side DefaultRobotHWSim::writeSim ----
 * The reference's ros_control bridge writes the controllers' joint commands into mjData on every control callback
 * (/root/reference mujoco_ros_control/src/default_robot_hw_sim.cpp:248-326): EFFORT -> qfrc_applied, POSITION -> qpos
 * (qvel = 0), VELOCITY -> qvel, POSITION_PID / VELOCITY_PID -> qfrc_applied = clamp(PID(error, dt), +-effort_limit),
 * with the e-stop rules (:251-261, :275, :305, :313-316).  Here the same write runs on the device at the start of
 * every step of every env, from per-env command arrays resident in HBM, so closed-loop batches need no host
 * callback.  PID = control_toolbox::Pid::computeCommand (absent dependency; restated from its documented algorithm:
 * p e + clamp(i int(e), i_min, i_max) + d de/dt, anti-windup variant clamps the integral).  Position errors:
 * prismatic cmd - q; continuous shortest angular distance; revolute: cmd clamped to [lower, upper] minus q (the
 * reference calls angles::shortest_angular_distance_with_limits, equal to this whenever both lie inside the limits).
 * Deviations, deliberate: joints are addressed by their MuJoCo joint id and qposadr (the reference indexes
 * jnt_dofadr with the transmission index, which only coincides for hinge / slide-only models in URDF order). */
enum { MJB_HW_EFFORT = 0, MJB_HW_POSITION = 1, MJB_HW_POSITION_PID = 2, MJB_HW_VELOCITY = 3, MJB_HW_VELOCITY_PID = 4 };
enum { MJB_HW_REVOLUTE = 0, MJB_HW_CONTINUOUS = 1, MJB_HW_PRISMATIC = 2 };
typedef struct mjb_hwsim_joint {
	int joint;            /* MuJoCo joint id (hinge or slide) */
	int method;           /* MJB_HW_* control method */
	int kind;             /* MJB_HW_REVOLUTE / CONTINUOUS / PRISMATIC */
	int antiwindup;
	double p, i, d, i_max, i_min;
	double effort_limit;  /* <= 0: unlimited */
	double lower, upper;  /* joint limits (revolute / prismatic) */
} mjb_hwsim_joint;
/* Register the controlled joints (replaces any previous set; n = 0 switches the stage off).  Commands start at 0. */
int mjb_hwsim_configure(mjb_batch *b, int n, const mjb_hwsim_joint *joints);
/* which: 0 position, 1 velocity, 2 effort commands; cmd is [env_hi - env_lo][n] in registration order */
int mjb_hwsim_set_command(mjb_batch *b, int which, int env_lo, int env_hi, const double *cmd);
void *mjb_hwsim_command_ptr(mjb_batch *b, int which); /* device [nenv][n] */
int mjb_hwsim_estop(mjb_batch *b, int active);
/* The controller cadence MujocoRosControlPlugin::controlCallback wraps around writeSim (mujoco_ros_control/src/
 * mujoco_ros_control_plugin.cpp:153-194), for the device-side stage: readSim -- the joint state the PIDs see -- every
 * `control_period` of sim time (ros::Duration arithmetic, first at the first non-zero time: nothing is read or written at t = 0,
 * :171-176), writeSim at every step after the first update with period = time - last write (:190-193), a time that went
 * backwards re-arms both stamps (:160-169).  control_period <= 0: a write at every step on the step's own state (default).
 * A control_period below the timestep is accepted with a warning on stderr (the reference: ROS_WARN, :100-105) -- the controller
 * then updates at every step.
 * The e-stop EDGE of :180-185 restarts the controller manager's controllers -- host-side objects; the PIDs of DefaultRobotHWSim
 * itself are never reset by the reference, nor here. */
int mjb_hwsim_set_period(mjb_batch *b, double control_period);

/* Profiling builds only (libmjb_prof.so): per-stage shader-cycle sums [0..31] and call counts [32..63] of
 * env 0; all zero in the production build.  A launch records the TWO probe ids [first_id, first_id + 2) selected by
 * mjb_debug_profile_window (32 bytes of LDS: the profiled kernel keeps the shipped kernel's residency); the tool
 * sweeps the window (tools/profile_stages.py).  No reference counterpart (MuJoCo's own mjTIMER_* slots are the
 * buckets the reference's GUI profiler plots, viewer.cpp:335-346). */
int mjb_debug_profile(mjb_batch *b, unsigned long long *out64, int clear);
int mjb_debug_profile_window(mjb_batch *b, int first_id);

/* Timing helper for bench.py: run `nlaunch` launches of mjb_step(b, nsteps) bracketed by HIP
 * events recorded on the batch's stream; returns the mean per-launch duration in milliseconds
 * through *ms_per_launch. */
int mjb_time_steps(mjb_batch *b, int nsteps, int nlaunch, double *ms_per_launch);

#ifdef __cplusplus
}
#endif
#endif /* MJB_H_ */
