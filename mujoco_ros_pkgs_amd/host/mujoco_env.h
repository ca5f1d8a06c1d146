// mujoco_env.h — ROS-free, batched restatement of the reference's core runtime class.
//
// Mirrors /root/reference mujoco_ros/include/mujoco_ros/mujoco_env.h:128-399 and
// mujoco_ros/src/mujoco_env.cpp for the hot path's scheduler and its step / reset / pause request
// semantics: same member names, same refusal rules, same
// callback order.  What changed: `model_` / `data_` are N env instances living on the GPU behind the
// C-ABI step engine (include/mjb.h); ROS parameters come from a ParamServer; /clock is an atomic.
//
// The stepper is reached through `mjr_backend` (a C vtable).  The product backend is libmjb (HIP) and
// nothing else; tests may inject a different vtable to exercise this host logic without a GPU.
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mjr_host.h"
#include "common_types.h"
#include "plugin_utils.h"

namespace mujoco_ros {

using MujocoEnvMutex = std::recursive_mutex;  // mujoco_env.h:90-92

struct ModelNames {
	std::vector<std::string> body, joint, geom, site, sensor, actuator, equality, tendon;
};

// ---- request / response payloads of the model-mutation services (mujoco_ros_msgs: BodyState, GeomProperties,
// EqualityConstraintParameters; poses {x, y, z, qw, qx, qy, qz}, twists {vx, vy, vz, wx, wy, wz})
struct BodyState {
	std::string name;
	double mass = 0;
	double pose[7] = { 0, 0, 0, 0, 0, 0, 0 };
	std::string pose_frame;   // header.frame_id: "" / "world", anything else needs tf (absent here -> refused like a failed transform)
	double twist[6] = { 0, 0, 0, 0, 0, 0 };
	std::string twist_frame;
};
struct GeomProperties {
	std::string name;
	int type = 0;  // mjtGeom
	double body_mass = 0, friction[3] = { 0, 0, 0 }, size[3] = { 0, 0, 0 };
};
struct EqualityConstraintParameters {
	std::string name, element1, element2;
	int type = 0;  // mjtEq
	bool active = false;
	double anchor[3] = { 0, 0, 0 }, relpose[7] = { 0, 0, 0, 0, 0, 0, 0 }, torquescale = 0, polycoef[5] = { 0, 0, 0, 0, 0 };
	struct { double dmin = 0, dmax = 0, width = 0, midpoint = 0, power = 0, timeconst = 0, dampratio = 0; } solverParameters;
};

class MujocoEnv {
public:
	explicit MujocoEnv(const std::string &admin_hash = std::string(), const ParamServer *initial_params = nullptr);
	~MujocoEnv();
	MujocoEnv(const MujocoEnv &) = delete;

	const double syncMisalign = 0.1;        // max mis-alignment before re-sync (sim seconds)  mujoco_env.h:144
	const double simRefreshFraction = 0.7;  // mujoco_env.h:145
	static constexpr double render_ui_rate_lower_bound_ = 0.0333;  // viewer.h:122 (only used with a viewer)

	// Noise to apply to the control signal (mujoco_env.h:148-150): forwarded to the device-side injector
	double ctrl_noise_std = 0.0;
	double ctrl_noise_rate = 0.0;

	MujocoEnvMutex physics_thread_mutex_;
	ParamServer params_;  // stands in for the private ros::NodeHandle

	struct {  // mujoco_env.h:162-193
		bool headless = true;
		bool render_offscreen = false;
		bool use_sim_time = true;
		int real_time_index = 0;  // batched default: unbound
		int busywait = 0;
		bool eval_mode = false;
		char admin_hash[64] = { 0 };
		std::atomic_int run = { 0 };
		std::atomic_int exit_request = { 0 };
		std::atomic_int visual_init_request = { 0 };
		std::atomic_int load_request = { 0 };
		std::atomic_int reset_request = { 0 };
		std::atomic_int speed_changed = { 0 };
		std::atomic_int env_steps_request = { 0 };
	} settings_;

	struct {
		float measured_slowdown = 1.0;
		bool model_valid = false;
	} sim_state_;

	static constexpr float percentRealTime[] = { -1,  // unbound                        mujoco_env.h:236-239
		                                          100, 80,   66,   50, 40,   33,   25, 20,   16,   13, 10,
		                                          8,   6.6f, 5.0f, 4,  3.3f, 2.5f, 2,  1.6f, 1.3f, 1,  .8f,
		                                          .66f, .5f, .4f, .33f, .25f, .2f, .16f, .13f, .1f };

	std::vector<MujocoPluginPtr> const &getPlugins() const { return plugins_; }

	// ---- model (re)load.  The reference queues a filename (settings_.load_request = 2, main.cpp:150-151);
	// here the queue holds an already compiled model description + the batch size + the backend factory.
	void queueModel(const mjb_model_desc *desc, const ModelNames &names, int nenv, int device = 0,
	                mjr_backend_factory factory = nullptr, void *factory_user = nullptr);

	void startPhysicsLoop();
	void startEventLoop();
	void waitForPhysicsJoin();
	void waitForEventsJoin();
	void shutdown();  // exit_request + join (MujocoEnvTestWrapper::shutdown, mujoco_env_fixture.h:62-70)

	int getOperationalStatus();  // mujoco_env.cpp:740-743
	int getPendingSteps() const { return num_steps_until_exit_.load(); }
	int isPhysicsRunning() const { return is_physics_running_; }
	int isEventRunning() const { return is_event_running_; }
	const std::string &loadError() const { return load_error_; }

	// mujoco_env.cpp:913-945
	bool step(int num_steps = 1, bool blocking = true);
	// mujoco_env.cpp:716-731
	bool togglePaused(bool paused, const std::string &admin_hash = std::string());

	// Step-action equivalent (callbacks.cpp:94-129): returns success; `preempt` may be set from another thread
	struct StepResult {
		bool success = false;
		bool preempted = false;
	};
	StepResult onStepGoal(int num_steps, const std::atomic_bool *preempt = nullptr,
	                      std::vector<int> *feedback_steps_left = nullptr);
	// service equivalents (callbacks.cpp:159-208): always "return true", report through the fields
	struct ServiceResponse {
		bool success = true;
		std::string status_message;
	};
	ServiceResponse setPauseCB(bool paused, const std::string &admin_hash);
	ServiceResponse shutdownCB();
	ServiceResponse resetCB();
	// ---- model / body-state services (callbacks.cpp:177-201, 210-592, 641-897; host/services.cpp).  The reference mutates its ONE
	// mjModel / mjData; here every handler takes an env range [env_lo, env_hi) (env_hi < 0: all envs) and writes per-env
	// parameter overrides through the backend (mjb_set_env_*), or reads env `env`.  Gate, lookups, rules and messages as there.
	ServiceResponse setBodyStateCB(BodyState state, bool set_pose, bool set_twist, bool set_mass, bool reset_qpos,
	                               const std::string &admin_hash, int env_lo = 0, int env_hi = -1);
	struct GetBodyStateResponse : ServiceResponse { BodyState state; };
	GetBodyStateResponse getBodyStateCB(const std::string &name, const std::string &admin_hash, int env = 0);
	ServiceResponse setGeomPropertiesCB(const GeomProperties &properties, bool set_type, bool set_mass, bool set_friction, bool set_size,
	                                    const std::string &admin_hash, int env_lo = 0, int env_hi = -1);
	struct GetGeomPropertiesResponse : ServiceResponse { GeomProperties properties; };
	GetGeomPropertiesResponse getGeomPropertiesCB(const std::string &geom_name, const std::string &admin_hash, int env = 0);
	ServiceResponse setGravityCB(const double gravity[3], const std::string &admin_hash, int env_lo = 0, int env_hi = -1);
	struct GetGravityResponse : ServiceResponse { double gravity[3] = { 0, 0, 0 }; };
	GetGravityResponse getGravityCB(const std::string &admin_hash, int env = 0);
	ServiceResponse setEqualityConstraintParametersArrayCB(const std::vector<EqualityConstraintParameters> &parameters,
	                                                       const std::string &admin_hash, int env_lo = 0, int env_hi = -1);
	struct GetEqualityResponse : ServiceResponse { std::vector<EqualityConstraintParameters> parameters; };
	GetEqualityResponse getEqualityConstraintParametersArrayCB(const std::vector<std::string> &names, const std::string &admin_hash, int env = 0);
	// reloadCB (:177-201): queue, wait for the loading request to drain, success = sim_state_.model_valid, message = load_error_
	ServiceResponse reloadCB(const mjb_model_desc *desc, const ModelNames &names, int nenv, int device = 0,
	                         mjr_backend_factory factory = nullptr, void *factory_user = nullptr);
	struct LoadingRequestState { int value = 0; std::string description; };
	LoadingRequestState getLoadingRequestState();        // :72-87
	ServiceResponse loadInitialJointStatesCB();          // :66-71

	// proxies (mujoco_env.h:241-251, callbacks.cpp:131-157)
	void runControlCbs();
	void runPassiveCbs();
	void runRenderCbs(mjvScene *scene);
	void runLastStageCbs();
	void notifyGeomChanged(int geom_id);
	// mujoco_env.cpp:163-176 with a device-side pair function (MJB_COLFUNC_*) instead of a host mjfCollision: 1 = an override of
	// this pair type existed (the reference's warning case), 0 = first registration, -1 = refused by the backend
	int registerCollisionFunction(int geom_type1, int geom_type2, int func);

	// per-env data access for tests / services (host mirror, refreshed from the device on demand)
	const mjModel *getModelPtr() const { return model_valid_.load() ? &model_ : nullptr; }
	mjData *getDataPtr(int env = 0);  // pulls the state fields of `env` from the device
	void commitData(int env = 0);     // pushes qpos / qvel / ctrl / qfrc_applied / xfrc_applied of `env` back
	int nenv() const { return nenv_; }
	double simTime() const { return sim_time_.load(); }  // what /clock carries (publishSimTime, :699-714)
	double dataTime() const { return data_time_.load(); }  // data_->time of env 0
	mjr_backend *backend() { return backend_; }
	unsigned long long stepCount() const { return step_count_.load(); }
	// envs [0, n) get host callbacks each step; default: all
	void setCallbackEnvs(int n) { cb_envs_ = n; }

protected:
	void physicsLoop();  // mujoco_env.cpp:436-639
	void eventLoop();    // mujoco_env.cpp:197-244
	void resetSim();     // mujoco_env.cpp:246-264
	void loadInitialJointStates();  // mujoco_env.cpp:266-389
	void setJointPosition(double pos, int joint_id, int jnt_axis = 0);  // :391-396 (all envs)
	void setJointVelocity(double vel, int joint_id, int jnt_axis = 0);  // :398-402
	void completeEnvSetup();  // :404-415
	void loadPlugins();       // :417-434
	void loadWithModelAndData();  // :745-769
	bool initModelFromQueue();    // :771-911
	void prepareReload();         // :947-961
	void publishSimTime(mjtNum time);
	// one burst of up to `n` env steps with the reference's per-step 5-tuple
	// (mj_step, publishSimTime, runLastStageCbs, render hand-off, counters); returns steps done
	int stepBurst(int n, bool count_requests);
	void pullViews(int lo, int hi, bool derived);
	void pullFields(const int *fields, int n, int lo, int hi);
	void unpinMirrors();
	void pushViews(int lo, int hi, bool with_passive = false);
	void bindView(int env, mjData &d);

	// queued model
	struct Queued {
		mjb_model_desc desc{};
		std::vector<std::vector<int>> iarr;
		std::vector<std::vector<double>> darr;
		ModelNames names;
		int nenv = 0, device = 0;
		mjr_backend_factory factory = nullptr;
		void *factory_user = nullptr;
		bool valid = false;
	} queued_, current_;

	mjModel model_{};
	std::atomic_bool model_valid_ = { false };  // (read by the physics loop before it takes the mutex: TSan, profiles/r03_sanitizers.txt)
	int nenv_ = 0;
	mjr_backend *backend_ = nullptr, *backend_new_ = nullptr;
	Queued staged_;
	std::vector<mjData> views_;                     // one per env
	std::vector<std::vector<double>> host_fields_;  // [field][nenv * dim] host mirrors
	std::vector<MujocoPluginPtr> plugins_;
	std::vector<MujocoPlugin *> cb_ready_plugins_;  // objects managed by plugins_
	std::vector<std::string> plugin_warnings_;
	int cb_envs_ = -1;
	mjData *cb_view_ = nullptr;  // env instance the running callback round is for
	bool xfrc_used_ = false;
	std::set<std::pair<int, int>> custom_collisions_;  // mujoco_env.h:285
	// what the callback-ready plugins declared (MujocoPlugin::callbackMask / viewFields), resolved in loadPlugins()
	unsigned cb_mask_ = 0;
	bool cb_all_fields_ = true;
	std::vector<int> cb_fields_;       // state fields to mirror after a step when cb_all_fields_ is false
	std::vector<void *> pinned_;       // host mirrors page-locked through the backend
	// small callback rounds move their fields through ONE packed transfer each way (mjr_backend::get_packed / set_packed): the
	// page-locked block they land in, scattered into / gathered from the per-field mirrors on the host
	std::vector<double> pack_host_;
	bool transferPacked(const std::vector<int> &fields, int lo, int hi, bool to_host);
	std::vector<double> init_qpos_, init_qvel_, init_qfrc_;

	// per-env mirrors of the model parameters the services change (allocated on first use; [env][...], initialised from the model)
	std::vector<double> env_gravity_, env_body_mass_, env_geom_friction_, env_geom_size_, env_equality_;
	std::vector<int> env_geom_type_;
	bool setEqualityConstraintParameters(const EqualityConstraintParameters &parameters, int lo, int hi, std::string &note);  // :641-746
	bool getEqualityConstraintParameters(EqualityConstraintParameters &parameters, int env);                               // :782-860
	bool authorized(const std::string &admin_hash) const { return !settings_.eval_mode || admin_hash == settings_.admin_hash; }
	int pushEnvParam(int what, int lo, int hi, const void *data, std::string &err);
	void ensureMirrors();
	void applyMassChange(int lo, int hi, std::string &err);  // `mj_setConst` with qpos kept (:244-258): the backend derives the constants

	std::atomic_int num_steps_until_exit_ = { -1 };  // (written by the physics thread, read by getPendingSteps() callers)
	std::atomic_int is_physics_running_ = { 0 }, is_event_running_ = { 0 };
	std::atomic<double> sim_time_ = { 0.0 };   // the /clock equivalent: only advances with use_sim_time
	std::atomic<double> data_time_ = { 0.0 };  // env 0's data->time after the latest step / forward: what the loop paces on
	std::atomic<unsigned long long> step_count_ = { 0 };
	std::string load_error_;
	std::thread physics_thread_handle_, event_thread_handle_;
	mjvScene scn_;
	unsigned long long noise_seed_ = 12345;
};

}  // namespace mujoco_ros
