// mujoco_env.cpp — batched, ROS-free restatement of the reference's scheduler around the step
// (/root/reference mujoco_ros/src/mujoco_env.cpp and the step / reset / pause handlers of callbacks.cpp).
// The physics itself is NOT here: every step goes through the mjr_backend vtable (libmjb on the GPU).
#include "mujoco_env.h"

#include <algorithm>

#include <cmath>
#include <cstring>
#include <sstream>
#include <stdexcept>

namespace mujoco_ros {

using Clock = std::chrono::steady_clock;
using Seconds = std::chrono::duration<double>;

constexpr float MujocoEnv::percentRealTime[];

namespace {
// host-mirrored fields handed to plugins through mjData
const int kStateFields[] = { MJB_F_qpos, MJB_F_qvel, MJB_F_act, MJB_F_ctrl, MJB_F_qacc, MJB_F_qacc_warmstart, MJB_F_qfrc_applied,
	                         MJB_F_xfrc_applied, MJB_F_sensordata, MJB_F_time, MJB_F_mocap_pos, MJB_F_mocap_quat };
const int kDerivedFields[] = { MJB_F_qfrc_passive, MJB_F_xpos, MJB_F_xquat, MJB_F_xmat, MJB_F_xipos, MJB_F_ximat,
	                           MJB_F_cvel, MJB_F_subtree_com, MJB_F_site_xpos, MJB_F_site_xmat, MJB_F_geom_xpos,
	                           MJB_F_geom_xmat, MJB_F_actuator_force, MJB_F_qfrc_bias, MJB_F_qfrc_actuator };
const int kWritableFields[] = { MJB_F_qpos, MJB_F_qvel, MJB_F_ctrl, MJB_F_qfrc_applied, MJB_F_xfrc_applied, MJB_F_mocap_pos,
	                            MJB_F_mocap_quat };
}  // namespace

MujocoEnv::MujocoEnv(const std::string &admin_hash, const ParamServer *initial_params)
{
	if (initial_params) params_ = *initial_params;
	// mujoco_env.cpp:88-102
	if (!admin_hash.empty()) std::strncpy(settings_.admin_hash, admin_hash.c_str(), sizeof(settings_.admin_hash) - 1);
	params_.param<bool>("eval_mode", settings_.eval_mode, false);
	if (settings_.eval_mode && !settings_.admin_hash[0]) {
		settings_.exit_request = 1;
		throw std::runtime_error("Evaluation mode requires a hash to verify critical operations are allowed. No hash was "
		                         "provided, aborting launch.");
	}
	// mujoco_env.cpp:104-114: offscreen rendering is out of scope; the flag only keeps the render-callback hook alive
	bool no_x = false;
	params_.param<bool>("render_offscreen", settings_.render_offscreen, false);
	params_.param<bool>("no_x", no_x, false);
	if (no_x && settings_.render_offscreen) settings_.render_offscreen = false;
	params_.param<bool>("use_sim_time", settings_.use_sim_time, true);
	publishSimTime(0);
	// mujoco_env.cpp:127-131
	bool run = true;
	params_.param<bool>("unpause", run, true);
	settings_.run = run;
	// mujoco_env.cpp:142
	{
		int num_steps = -1;
		params_.param<int>("num_steps", num_steps, -1);
		num_steps_until_exit_.store(num_steps);
	}
	params_.param<int>("realtime_index", settings_.real_time_index, 0);
	params_.param<double>("ctrl_noise_std", ctrl_noise_std, 0.0);
	params_.param<double>("ctrl_noise_rate", ctrl_noise_rate, 0.0);
}

MujocoEnv::~MujocoEnv()
{
	shutdown();
	cb_ready_plugins_.clear();
	plugins_.clear();
	unpinMirrors();
	if (backend_) backend_->destroy(backend_->self);
	if (backend_new_) backend_new_->destroy(backend_new_->self);
}

// ------------------------------------------------------------------------------------ threads
void MujocoEnv::startPhysicsLoop() { physics_thread_handle_ = std::thread(&MujocoEnv::physicsLoop, this); }
void MujocoEnv::startEventLoop() { event_thread_handle_ = std::thread(&MujocoEnv::eventLoop, this); }
void MujocoEnv::waitForPhysicsJoin()
{
	if (physics_thread_handle_.joinable()) physics_thread_handle_.join();
}
void MujocoEnv::waitForEventsJoin()
{
	if (event_thread_handle_.joinable()) event_thread_handle_.join();
}
void MujocoEnv::shutdown()
{
	settings_.exit_request.store(1);
	waitForPhysicsJoin();
	waitForEventsJoin();
}

int MujocoEnv::getOperationalStatus()
{
	return std::max(settings_.load_request.load(), std::max(settings_.visual_init_request.load(), settings_.reset_request.load()));
}

void MujocoEnv::publishSimTime(mjtNum time)
{
	// the loop paces itself on data_->time whatever use_sim_time says (mujoco_env.cpp:466-467, :536, :560); only the
	// /clock publication is gated by it (:701-703)
	data_time_.store(time);
	if (!settings_.use_sim_time) return;
	sim_time_.store(time);  // the reference publishes /clock and spins until ros::Time::now() >= time (:699-714)
}

// ------------------------------------------------------------------------------------ model (re)load
void MujocoEnv::queueModel(const mjb_model_desc *desc, const ModelNames &names, int nenv, int device,
                           mjr_backend_factory factory, void *factory_user)
{
	std::lock_guard<MujocoEnvMutex> lock(physics_thread_mutex_);
	Queued &q = queued_;
	q = Queued();
	q.desc = *desc;
	// deep copy of every array so the caller's buffers may go away
#define MJB_SIZE(name)
#define MJB_OPT_I(name)
#define MJB_OPT_D(name, n)
#define MJB_ARR_I(name, rows, cols)                                                                     \
	q.iarr.emplace_back(desc->name ? std::vector<int>(desc->name, desc->name + (size_t)desc->rows * (cols)) : std::vector<int>());
#define MJB_ARR_D(name, rows, cols)                                                                     \
	q.darr.emplace_back(desc->name ? std::vector<double>(desc->name, desc->name + (size_t)desc->rows * (cols)) : std::vector<double>());
#include "../../include/mjb_model_fields.def"
#undef MJB_ARR_I
#undef MJB_ARR_D
	{
		size_t ii = 0, di = 0;
#define MJB_ARR_I(name, rows, cols) q.desc.name = q.iarr[ii++].data();
#define MJB_ARR_D(name, rows, cols) q.desc.name = q.darr[di++].data();
#include "../../include/mjb_model_fields.def"
#undef MJB_SIZE
#undef MJB_OPT_I
#undef MJB_OPT_D
#undef MJB_ARR_I
#undef MJB_ARR_D
	}
	q.names = names;
	q.nenv = nenv;
	q.device = device;
	q.factory = factory ? factory : mjr_make_mjb_backend;
	q.factory_user = factory_user;
	q.valid = true;
	settings_.load_request.store(2);
}

// mujoco_env.cpp:771-911: build the new model/data; on failure keep the old model and report load_error_
bool MujocoEnv::initModelFromQueue()
{
	if (!queued_.valid) {
		load_error_ = "no model queued";
		return false;
	}
	staged_ = std::move(queued_);
	queued_ = Queued();
	// vectors moved: re-point the descriptor
	{
		size_t ii = 0, di = 0;
#define MJB_SIZE(name)
#define MJB_OPT_I(name)
#define MJB_OPT_D(name, n)
#define MJB_ARR_I(name, rows, cols) staged_.desc.name = staged_.iarr[ii++].data();
#define MJB_ARR_D(name, rows, cols) staged_.desc.name = staged_.darr[di++].data();
#include "../../include/mjb_model_fields.def"
#undef MJB_SIZE
#undef MJB_OPT_I
#undef MJB_OPT_D
#undef MJB_ARR_I
#undef MJB_ARR_D
	}
	backend_new_ = staged_.factory(&staged_.desc, staged_.nenv, staged_.device, staged_.factory_user);
	if (!backend_new_) {
		load_error_ = std::string("could not create the step backend: ") + mjr_last_error();
		sim_state_.model_valid = model_valid_.load();  // the old model (if any) stays in place
		staged_ = Queued();
		return false;
	}
	return true;
}

// mujoco_env.cpp:947-961
// mujoco_env.cpp:163-176.  The reference's duplicate test looks for (type1, type2) AND (type2, type2) -- a typo for the
// swapped pair; here the pair is unordered, which is what the warning text describes.
int MujocoEnv::registerCollisionFunction(int geom_type1, int geom_type2, int func)
{
	std::lock_guard<MujocoEnvMutex> lock(physics_thread_mutex_);
	if (!backend_ || !backend_->register_collision) return -1;
	const std::pair<int, int> key(std::min(geom_type1, geom_type2), std::max(geom_type1, geom_type2));
	const bool dup = custom_collisions_.count(key) != 0;
	if (dup)
		plugin_warnings_.push_back("A user defined collision callback for collisions between geoms of type " + std::to_string(geom_type1) +
		                           " and " + std::to_string(geom_type2) + " have already been registered. This might lead to unexpected behavior!");
	if (backend_->register_collision(backend_->self, geom_type1, geom_type2, func) != 0) return -1;
	custom_collisions_.insert(key);
	return dup ? 1 : 0;
}

void MujocoEnv::prepareReload()
{
	// "Resetting collision cbs to default" (:949-954): a freshly made backend starts with the built-in pair functions
	custom_collisions_.clear();
	cb_ready_plugins_.clear();
	plugins_.clear();
}

// mujoco_env.cpp:745-769
void MujocoEnv::unpinMirrors()
{
	if (backend_ && backend_->host_unregister)
		for (void *p : pinned_) backend_->host_unregister(backend_->self, p);
	pinned_.clear();
	// (the packed transfer buffer is page-locked where it is first sized, transferPacked: emptied here so that the next transfer
	//  sizes and registers it again with the backend that is current then -- left alone, the transfers after a reload ran from
	//  pageable memory)
	pack_host_.clear();
	pack_host_.shrink_to_fit();
}

void MujocoEnv::loadWithModelAndData()
{
	unpinMirrors();  // (the mirrors are reallocated below; their page locks go with the backend that made them)
	if (backend_) backend_->destroy(backend_->self);
	backend_ = backend_new_;
	backend_new_ = nullptr;
	current_ = std::move(staged_);
	staged_ = Queued();
	{
		size_t ii = 0, di = 0;
#define MJB_SIZE(name)
#define MJB_OPT_I(name)
#define MJB_OPT_D(name, n)
#define MJB_ARR_I(name, rows, cols) current_.desc.name = current_.iarr[ii++].data();
#define MJB_ARR_D(name, rows, cols) current_.desc.name = current_.darr[di++].data();
#include "../../include/mjb_model_fields.def"
#undef MJB_SIZE
#undef MJB_OPT_I
#undef MJB_OPT_D
#undef MJB_ARR_I
#undef MJB_ARR_D
	}
	const mjb_model_desc &d = current_.desc;
	nenv_ = current_.nenv;
	mjModel &m = model_;
	m.nq = d.nq; m.nv = d.nv; m.nu = d.nu; m.na = d.na; m.nbody = d.nbody; m.njnt = d.njnt; m.ngeom = d.ngeom;
	m.nsite = d.nsite; m.nsensor = d.nsensor; m.nsensordata = d.nsensordata;
	m.opt.timestep = d.timestep[0];
	for (int k = 0; k < 3; k++) m.opt.gravity[k] = d.gravity[k];
	m.opt.tolerance = d.tolerance[0];
	m.opt.impratio = d.impratio[0];
	m.opt.integrator = d.integrator; m.opt.cone = d.cone; m.opt.solver = d.solver; m.opt.iterations = d.iterations;
	m.opt.disableflags = d.disableflags;
	m.desc = &current_.desc;
	m.jnt_type = d.jnt_type; m.jnt_qposadr = d.jnt_qposadr; m.jnt_dofadr = d.jnt_dofadr; m.jnt_bodyid = d.jnt_bodyid;
	m.body_jntadr = d.body_jntadr; m.body_jntnum = d.body_jntnum; m.geom_bodyid = d.geom_bodyid; m.geom_type = d.geom_type;
	m.site_bodyid = d.site_bodyid; m.sensor_type = d.sensor_type; m.sensor_adr = d.sensor_adr; m.sensor_dim = d.sensor_dim;
	m.sensor_objid = d.sensor_objid; m.sensor_objtype = d.sensor_objtype; m.sensor_refid = d.sensor_refid;
	m.sensor_reftype = d.sensor_reftype;
	m.qpos0 = d.qpos0; m.body_mass = d.body_mass; m.geom_size = d.geom_size; m.geom_friction = d.geom_friction;
	m.sensor_cutoff = d.sensor_cutoff;
	m.joint_names = current_.names.joint; m.body_names = current_.names.body; m.geom_names = current_.names.geom;
	m.site_names = current_.names.site; m.sensor_names = current_.names.sensor; m.actuator_names = current_.names.actuator;
	m.equality_names = current_.names.equality; m.tendon_names = current_.names.tendon;
	env_gravity_.clear(); env_body_mass_.clear(); env_geom_friction_.clear(); env_geom_size_.clear(); env_equality_.clear();
	env_geom_type_.clear();  // (a fresh backend starts from the model's values)
	model_valid_ = true;

	// host mirrors + views
	host_fields_.assign(MJB_F_COUNT, std::vector<double>());
	auto alloc = [&](int f) { host_fields_[f].assign((size_t)nenv_ * std::max(1, backend_->field_size(backend_->self, f)), 0.0); };
	for (int f : kStateFields) alloc(f);
	for (int f : kDerivedFields) alloc(f);
	// page-locked mirrors: the per-step copies around a callback round become DMA transfers (no staging, truly asynchronous)
	if (backend_->host_register)
		for (auto &v : host_fields_)
			if (!v.empty() && backend_->host_register(backend_->self, v.data(), (unsigned long long)v.size() * sizeof(double)) == 0)
				pinned_.push_back(v.data());
	views_.assign(nenv_, mjData());
	for (int e = 0; e < nenv_; e++) bindView(e, views_[e]);

	prepareReload();
	completeEnvSetup();
	load_error_.clear();
	sim_state_.model_valid = true;
}

void MujocoEnv::bindView(int env, mjData &d)
{
	auto p = [&](int f) { return host_fields_[f].data() + (size_t)env * std::max(1, backend_->field_size(backend_->self, f)); };
	d.env_id = env;
	d.time = 0;
	d.qpos = p(MJB_F_qpos); d.qvel = p(MJB_F_qvel); d.ctrl = p(MJB_F_ctrl); d.qacc = p(MJB_F_qacc);
	d.qacc_warmstart = p(MJB_F_qacc_warmstart); d.qfrc_applied = p(MJB_F_qfrc_applied);
	d.xfrc_applied = p(MJB_F_xfrc_applied); d.qfrc_passive = p(MJB_F_qfrc_passive); d.sensordata = p(MJB_F_sensordata);
	d.xpos = p(MJB_F_xpos); d.xquat = p(MJB_F_xquat); d.xmat = p(MJB_F_xmat); d.xipos = p(MJB_F_xipos);
	d.ximat = p(MJB_F_ximat); d.cvel = p(MJB_F_cvel); d.subtree_com = p(MJB_F_subtree_com);
	d.site_xpos = p(MJB_F_site_xpos); d.site_xmat = p(MJB_F_site_xmat); d.geom_xpos = p(MJB_F_geom_xpos);
	d.geom_xmat = p(MJB_F_geom_xmat); d.actuator_force = p(MJB_F_actuator_force); d.qfrc_bias = p(MJB_F_qfrc_bias);
	d.qfrc_actuator = p(MJB_F_qfrc_actuator);
	d.mocap_pos = p(MJB_F_mocap_pos); d.mocap_quat = p(MJB_F_mocap_quat);
	d.act = p(MJB_F_act);
}

// mujoco_env.cpp:404-415
void MujocoEnv::completeEnvSetup()
{
	loadInitialJointStates();
	if (ctrl_noise_std > 0) backend_->set_ctrl_noise(backend_->self, ctrl_noise_std, ctrl_noise_rate, noise_seed_, 0);
	loadPlugins();
}

// mujoco_env.cpp:417-434
void MujocoEnv::loadPlugins()
{
	cb_ready_plugins_.clear();
	cb_ready_plugins_.shrink_to_fit();
	ConfigValue plugin_config;
	if (plugin_utils::parsePlugins(&params_, plugin_config))
		plugin_utils::registerPlugins("~", plugin_config, plugins_, this, &params_, &plugin_warnings_);
	if (!plugins_.empty()) pullViews(0, 1, false);
	for (const auto &plugin : plugins_)
		if (plugin->safe_load(&model_, &views_[0])) cb_ready_plugins_.emplace_back(plugin.get());
	// what the callback-ready plugins declared: which callbacks exist, which view fields they read
	cb_mask_ = 0;
	cb_all_fields_ = false;
	cb_fields_.clear();
	for (const auto *plugin : cb_ready_plugins_) {
		cb_mask_ |= plugin->callbackMask();
		std::vector<int> fl;
		if (!plugin->viewFields(fl)) cb_all_fields_ = true;
		for (int f : fl)
			if (std::find(cb_fields_.begin(), cb_fields_.end(), f) == cb_fields_.end()) cb_fields_.push_back(f);
	}
	if (std::find(cb_fields_.begin(), cb_fields_.end(), (int)MJB_F_time) == cb_fields_.end()) cb_fields_.push_back(MJB_F_time);
}

// ------------------------------------------------------------------------------------ host <-> device views
// One transfer for all of `fields` of envs [lo, hi) when the backend offers it and the block is small (a latency-bound round:
// per-field copies cost ~8 us of driver time each; beyond kPackedMaxBytes the copies are bandwidth-bound and the extra host-side
// scatter would only add to them).  false = not taken, the caller falls back to one copy per field.
bool MujocoEnv::transferPacked(const std::vector<int> &fields, int lo, int hi, bool to_host)
{
	const size_t kPackedMaxBytes = 4u << 20;
	if ((to_host ? backend_->get_packed == nullptr : backend_->set_packed == nullptr) || fields.size() < 3 || fields.size() > 40 || hi <= lo) return false;
	size_t total = 0;
	for (int f : fields) total += (size_t)(hi - lo) * std::max(0, backend_->field_size(backend_->self, f));
	if (total == 0 || total * sizeof(double) > kPackedMaxBytes) return false;
	if (pack_host_.size() < kPackedMaxBytes / sizeof(double)) {
		pack_host_.assign(kPackedMaxBytes / sizeof(double), 0.0);
		if (backend_->host_register && backend_->host_register(backend_->self, pack_host_.data(), kPackedMaxBytes) == 0) pinned_.push_back(pack_host_.data());
	}
	size_t off = 0;
	if (!to_host) {
		for (int f : fields) {
			const size_t sz = (size_t)std::max(0, backend_->field_size(backend_->self, f)), nb = (size_t)(hi - lo) * sz;
			if (nb) std::memcpy(pack_host_.data() + off, host_fields_[f].data() + (size_t)lo * sz, nb * sizeof(double));
			off += nb;
		}
		return backend_->set_packed(backend_->self, (int)fields.size(), fields.data(), lo, hi, pack_host_.data()) == 0;
	}
	if (backend_->get_packed(backend_->self, (int)fields.size(), fields.data(), lo, hi, pack_host_.data()) != 0) return false;
	for (int f : fields) {
		const size_t sz = (size_t)std::max(0, backend_->field_size(backend_->self, f)), nb = (size_t)(hi - lo) * sz;
		if (nb) std::memcpy(host_fields_[f].data() + (size_t)lo * sz, pack_host_.data() + off, nb * sizeof(double));
		off += nb;
	}
	return true;
}

void MujocoEnv::pullFields(const int *fields, int n, int lo, int hi)
{
	std::vector<int> fl;
	std::vector<double *> ptr;
	for (int k = 0; k < n; k++) {
		const int f = fields[k], sz = backend_->field_size(backend_->self, f);
		if (sz <= 0) continue;
		fl.push_back(f);
		ptr.push_back(host_fields_[f].data() + (size_t)lo * sz);
	}
	if (transferPacked(fl, lo, hi, true)) {
		for (int e = lo; e < hi; e++) views_[e].time = host_fields_[MJB_F_time][e];
		return;
	}
	int rc = 0;
	if (backend_->get_many) {
		rc = backend_->get_many(backend_->self, (int)fl.size(), fl.data(), lo, hi, ptr.data());  // async copies, ONE synchronisation
	} else {
		for (size_t k = 0; k < fl.size() && rc == 0; k++) rc = backend_->get(backend_->self, fl[k], lo, hi, ptr[k]);
	}
	if (rc != 0) plugin_warnings_.push_back(std::string("view refresh failed: ") + backend_->last_error(backend_->self));
	for (int e = lo; e < hi; e++) views_[e].time = host_fields_[MJB_F_time][e];
}

void MujocoEnv::pullViews(int lo, int hi, bool derived)
{
	if (!backend_ || hi <= lo) return;
	std::vector<int> fl(std::begin(kStateFields), std::end(kStateFields));
	if (derived) fl.insert(fl.end(), std::begin(kDerivedFields), std::end(kDerivedFields));
	pullFields(fl.data(), (int)fl.size(), lo, hi);
}

void MujocoEnv::pushViews(int lo, int hi, bool with_passive)
{
	if (!backend_ || hi <= lo) return;
	std::vector<int> fl;
	std::vector<const double *> ptr;
	for (int f : kWritableFields) {
		const int n = backend_->field_size(backend_->self, f);
		if (n <= 0) continue;
		if (f == MJB_F_xfrc_applied) {  // only pay for the Cartesian-wrench path once somebody uses it
			bool any = false;
			const double *x = host_fields_[f].data() + (size_t)lo * n;
			for (size_t k = 0; k < (size_t)(hi - lo) * n && !any; k++) any = x[k] != 0;
			if (!any && !xfrc_used_) continue;
			xfrc_used_ = true;
		}
		fl.push_back(f);
		ptr.push_back(host_fields_[f].data() + (size_t)lo * n);
	}
	if (with_passive && backend_->field_size(backend_->self, MJB_F_qfrc_passive) > 0) {  // (between step1 and step2 only)
		fl.push_back(MJB_F_qfrc_passive);
		ptr.push_back(host_fields_[MJB_F_qfrc_passive].data() + (size_t)lo * backend_->field_size(backend_->self, MJB_F_qfrc_passive));
	}
	if (transferPacked(fl, lo, hi, false)) return;
	if (backend_->set_many) {
		backend_->set_many(backend_->self, (int)fl.size(), fl.data(), lo, hi, ptr.data());
	} else {
		for (size_t k = 0; k < fl.size(); k++) backend_->set(backend_->self, fl[k], lo, hi, ptr[k]);
	}
}

mjData *MujocoEnv::getDataPtr(int env)
{
	std::lock_guard<MujocoEnvMutex> lock(physics_thread_mutex_);
	if (!model_valid_ || env < 0 || env >= nenv_) return nullptr;
	pullViews(env, env + 1, false);
	return &views_[env];
}

void MujocoEnv::commitData(int env)
{
	std::lock_guard<MujocoEnvMutex> lock(physics_thread_mutex_);
	if (!model_valid_ || env < 0 || env >= nenv_) return;
	pushViews(env, env + 1);
	backend_->synchronize(backend_->self);  // the copies are asynchronous DMA out of the (page-locked) view: the caller may reuse it now
}

// ------------------------------------------------------------------------------------ callbacks fan-out
// callbacks.cpp:131-157; `cb_view_` is the env instance the current callback round is for
// (a plugin is called back only for what its callbackMask() declares: the runtime plans launches on the declaration -- fused,
//  split, chained -- so a callback outside it would see a view the plan did not refresh)
void MujocoEnv::runControlCbs()
{
	for (const auto &plugin : cb_ready_plugins_)
		if (plugin->callbackMask() & MujocoPlugin::CB_CONTROL) plugin->controlCallback(&model_, cb_view_);
}
void MujocoEnv::runPassiveCbs()
{
	for (const auto &plugin : cb_ready_plugins_)
		if (plugin->callbackMask() & MujocoPlugin::CB_PASSIVE) plugin->passiveCallback(&model_, cb_view_);
}
void MujocoEnv::runRenderCbs(mjvScene *scene)
{
	for (const auto &plugin : cb_ready_plugins_)
		if (plugin->callbackMask() & MujocoPlugin::CB_RENDER) plugin->renderCallback(&model_, cb_view_, scene);
}
void MujocoEnv::runLastStageCbs()
{
	for (const auto &plugin : cb_ready_plugins_)
		if (plugin->callbackMask() & MujocoPlugin::CB_LASTSTAGE) plugin->lastStageCallback(&model_, cb_view_);
}
void MujocoEnv::notifyGeomChanged(int geom_id)
{
	std::lock_guard<MujocoEnvMutex> lock(physics_thread_mutex_);
	if (!model_valid_) return;
	for (const auto &plugin : cb_ready_plugins_) plugin->onGeomChanged(&model_, &views_[0], geom_id);
}

// ------------------------------------------------------------------------------------ the step 5-tuple
// mujoco_env.cpp:498-520 / :552-574 / :593-612, for all envs of the batch.  Without callback-ready plugins
// the burst is ONE fused launch; with plugins every step is split at the point where mj_step would invoke
// mjcb_passive / mjcb_control so the plugins see exactly the fields MuJoCo would show them.
int MujocoEnv::stepBurst(int n, bool count_requests)
{
	if (n <= 0) return 0;
	const int ncb = cb_envs_ < 0 ? nenv_ : std::min(cb_envs_, nenv_);
	int done = 0;
	if (cb_ready_plugins_.empty() || ncb == 0) {
		if (backend_->step(backend_->self, n) != 0) {
			load_error_ = backend_->last_error(backend_->self);
			settings_.exit_request.store(1);
			return 0;
		}
		double t = 0;
		backend_->get(backend_->self, MJB_F_time, 0, 1, &t);
		publishSimTime(t);
		done = n;
		step_count_ += (unsigned long long)n;
		if (count_requests) settings_.env_steps_request.fetch_sub(n);
		if (num_steps_until_exit_.load() > 0) num_steps_until_exit_.store(std::max(0, num_steps_until_exit_.load() - n));
		return done;
	}
	// (the fused observer path below refreshes STATE fields only: a plugin that wants "any field" (viewFields() == false) or names a
	//  derived one -- xpos, geom_xpos, qfrc_* ... for lastStageCallback / renderCallback -- takes the split path, which pulls them)
	bool observers_read_state_only = !cb_all_fields_;
	for (int f : cb_fields_)
		if (std::find(std::begin(kStateFields), std::end(kStateFields), f) == std::end(kStateFields)) observers_read_state_only = false;
	if (!(cb_mask_ & (MujocoPlugin::CB_CONTROL | MujocoPlugin::CB_PASSIVE)) && observers_read_state_only) {
		// Only end-of-step observers (lastStageCallback / renderCallback, e.g. the sensors plugin): nothing can change the
		// step from the host, so every step is ONE fused launch (no split, no frame workspace) followed by one batched copy
		// of the fields the plugins read -- the state fields, or exactly the ones they named.
		const std::vector<int> &fl = cb_fields_;
		for (int s = 0; s < n; s++) {
			const double t_before = views_[0].time;
			if ((backend_->step_async ? backend_->step_async(backend_->self, 1) : backend_->step(backend_->self, 1)) != 0) break;
			pullFields(fl.data(), (int)fl.size(), 0, ncb);
			publishSimTime(views_[0].time);
			for (int e = 0; e < ncb; e++) {
				cb_view_ = &views_[e];
				runLastStageCbs();
				if (settings_.render_offscreen) runRenderCbs(&scn_);
			}
			cb_view_ = &views_[0];
			done++;
			step_count_ += 1;
			if (count_requests) settings_.env_steps_request.fetch_sub(1);
			if (num_steps_until_exit_ > 0) num_steps_until_exit_--;
			if (views_[0].time < t_before) break;  // "Break if reset"
			if (count_requests && settings_.env_steps_request.load() <= 0) break;
			if (settings_.exit_request.load() || num_steps_until_exit_ == 0) break;
		}
		return done;
	}
	// Only the callback envs [0, ncb) are split at the callback point; the others take the same step as one fused launch, enqueued
	// once the callback envs' fields are on the host so that it runs under the callbacks (backends without the prefix entry points
	// split every env).  With control / passive callbacks only -- nobody looks at the finished step before the next callback round
	// -- consecutive steps of a burst are CHAINED: the second half of step s and the first half of step s + 1 are one launch
	// (step21_prefix), one kernel and one device -> host round trip per step instead of two (round 3: 131 -> ~75 us per step).
	const bool prefix = backend_->step1_prefix && backend_->step_rest && backend_->step2_prefix;
	// RK4 (<option integrator="RK4">): mj_RungeKutta runs mj_forwardSkip -- and mjcb_passive / mjcb_control inside it -- for each of
	// its four evaluations, lastStageCallback once per step (plugin_utils.h:119-125 gives exactly this as the reason it exists):
	// the second half is cut at the evaluations (step2_rk) and the callbacks fire at each, on the view of that evaluation.
	const bool rk4cb = model_.opt.integrator == MJB_INT_RK4 && backend_->step2_rk != nullptr;
	const bool can_chain = prefix && backend_->step21_prefix && !(cb_mask_ & (MujocoPlugin::CB_LASTSTAGE | MujocoPlugin::CB_RENDER)) &&
	                       !settings_.render_offscreen && !rk4cb;
	bool primed = false;        // the first half of this step already ran (in the previous step's chained launch)
	bool stop_after = false;    // a reset was seen after the chain was committed: finish the primed step, then leave
	double t_prev = views_[0].time;
	for (int s = 0; s < n; s++) {
		const double t_before = primed ? t_prev : views_[0].time;
		if (!primed && (prefix ? backend_->step1_prefix(backend_->self, ncb) : backend_->step1(backend_->self)) != 0) break;
		pullViews(0, ncb, true);
		if (primed) {  // the pull above carries the finished step's state: what the unchained path publishes after step2
			publishSimTime(views_[0].time);
			if (views_[0].time < t_before) stop_after = true;  // "Break if reset" -- one step late: the chained half has run
		}
		if (prefix && backend_->step_rest(backend_->self, ncb) != 0) break;
		for (int e = 0; e < ncb; e++) {  // mjcb_passive then mjcb_control, in registration order per env
			cb_view_ = &views_[e];
			runPassiveCbs();
			runControlCbs();
		}
		pushViews(0, ncb, true);  // the writable state fields + qfrc_passive (what mjcb_passive adds to), one transfer when small
		// chain when the next trip of this loop is certain to run (everything but a reset is known now)
		const bool more = s + 1 < n && !stop_after && !(count_requests && settings_.env_steps_request.load() - 1 <= 0) &&
		                  !settings_.exit_request.load() && num_steps_until_exit_.load() != 1;
		if (can_chain && more) {
			if (backend_->step21_prefix(backend_->self, ncb) != 0) break;
			t_prev = views_[0].time;
			primed = true;
			cb_view_ = &views_[0];
			done++;
			step_count_ += 1;
			if (count_requests) settings_.env_steps_request.fetch_sub(1);
			if (num_steps_until_exit_ > 0) num_steps_until_exit_--;
			continue;
		}
		primed = false;
		if (rk4cb) {
			bool failed = false;
			for (int rk = 0; rk < 3 && !failed; rk++) {
				if (backend_->step2_rk(backend_->self, prefix ? ncb : -1, rk) != 0) {
					failed = true;
					break;
				}
				pullViews(0, ncb, true);  // evaluation rk + 1: its state, its derived fields, time = t0 + c h
				for (int e = 0; e < ncb; e++) {
					cb_view_ = &views_[e];
					runPassiveCbs();
					runControlCbs();
				}
				pushViews(0, ncb, true);
			}
			if (failed || backend_->step2_rk(backend_->self, prefix ? ncb : -1, 3) != 0) break;
		} else if ((prefix ? backend_->step2_prefix(backend_->self, ncb) : backend_->step2(backend_->self)) != 0) break;
		pullViews(0, ncb, false);
		publishSimTime(views_[0].time);
		for (int e = 0; e < ncb; e++) {
			cb_view_ = &views_[e];
			runLastStageCbs();
			if (settings_.render_offscreen) runRenderCbs(&scn_);  // hand-off point of mujoco_env.cpp:501-515
		}
		cb_view_ = &views_[0];
		done++;
		step_count_ += 1;
		if (count_requests) settings_.env_steps_request.fetch_sub(1);
		if (num_steps_until_exit_ > 0) num_steps_until_exit_--;
		if (stop_after || views_[0].time < t_before) break;  // "Break if reset"
		if (count_requests && settings_.env_steps_request.load() <= 0) break;
		if (settings_.exit_request.load() || num_steps_until_exit_ == 0) break;
	}
	return done;
}

// ------------------------------------------------------------------------------------ mujoco_env.cpp:436-639
void MujocoEnv::physicsLoop()
{
	is_physics_running_ = 1;
	Clock::time_point syncCPU{};
	mjtNum syncSim = 0;
	const int kFuseChunk = 256;  // fused steps per launch between two looks at the request flags

	while (!settings_.exit_request.load() && num_steps_until_exit_ != 0) {
		if (settings_.run.load() && settings_.busywait) std::this_thread::yield();
		else std::this_thread::sleep_for(std::chrono::milliseconds(1));

		if (!model_valid_) continue;
		if (!physics_thread_mutex_.try_lock()) continue;
		if (settings_.run.load()) {
			const auto startCPU = Clock::now();
			const auto elapsedCPU = startCPU - syncCPU;
			const double simtime = data_time_.load();
			const double elapsedSim = simtime - syncSim;
			// ctrl noise (mujoco_env.cpp:469-481) is generated on the device right before each step
			const double slowdown = settings_.real_time_index == 0 ? 1.0 : 100.0 / percentRealTime[settings_.real_time_index];
			const bool misaligned = std::fabs(Seconds(elapsedCPU).count() / slowdown - elapsedSim) > syncMisalign;
			if (elapsedSim < 0 || elapsedCPU.count() < 0 || syncCPU.time_since_epoch().count() == 0 || misaligned ||
			    settings_.speed_changed) {
				// out of sync: re-sync, ONE step (mujoco_env.cpp:490-521)
				syncCPU = startCPU;
				syncSim = simtime;
				settings_.speed_changed = false;
				stepBurst(1, false);
			} else {
				// in sync: step until ahead of the wall clock (unbounded when real_time_index == 0).
				// NOTE: the reference's loop condition `&& !connected_viewers_.empty()` (:531-537) disables this
				// branch when no viewer is attached; the intent documented in SURVEY.md §3.2 ("bounded by 1/30 s
				// only when a viewer is connected") is what is implemented here.
				bool measured = false;
				const double dt = model_.opt.timestep;
				while ((settings_.real_time_index == 0 ||
				        Seconds((data_time_.load() - syncSim) * slowdown) < Clock::now() - syncCPU) &&
				       Clock::now() - startCPU < Seconds(render_ui_rate_lower_bound_) && !settings_.exit_request.load() &&
				       num_steps_until_exit_ != 0 && settings_.run.load()) {
					if (!measured && elapsedSim != 0) {
						sim_state_.measured_slowdown = (float)(Seconds(elapsedCPU).count() / elapsedSim);
						measured = true;
					}
					int n = 1;
					if (settings_.real_time_index == 0) n = kFuseChunk;
					else {
						const double behind = Seconds(Clock::now() - syncCPU).count() / slowdown - (data_time_.load() - syncSim);
						n = std::max(1, std::min(kFuseChunk, (int)(behind / dt)));
					}
					if (num_steps_until_exit_ > 0) n = std::min(n, num_steps_until_exit_.load());
					const double prev = data_time_.load();
					if (stepBurst(n, false) == 0) break;
					if (data_time_.load() < prev) break;  // reset
				}
			}
		} else {
			// paused (mujoco_env.cpp:584-623)
			if (settings_.env_steps_request.load() > 0) {
				syncSim = data_time_.load();
				while (settings_.env_steps_request.load() > 0 && !settings_.exit_request.load()) {
					int n = std::min(settings_.env_steps_request.load(), kFuseChunk);
					if (num_steps_until_exit_ > 0) n = std::min(n, num_steps_until_exit_.load());
					if (n <= 0 || stepBurst(n, true) == 0) break;
					if (data_time_.load() < syncSim) break;
					if (num_steps_until_exit_ == 0) break;
				}
			} else {
				backend_->forward(backend_->self);  // mj_forward keeps derived quantities fresh (:619-622)
				double t = 0;
				backend_->get(backend_->self, MJB_F_time, 0, 1, &t);
				publishSimTime(t);
			}
		}
		physics_thread_mutex_.unlock();
	}
	is_physics_running_ = 0;
}

// ------------------------------------------------------------------------------------ mujoco_env.cpp:197-244
void MujocoEnv::eventLoop()
{
	is_event_running_ = 1;
	while (!settings_.exit_request.load()) {
		{
			std::unique_lock<MujocoEnvMutex> lock(physics_thread_mutex_);
			if (settings_.load_request.load() == 1) {
				loadWithModelAndData();
				settings_.load_request.store(0);
			} else if (settings_.load_request.load() >= 2) {
				if (initModelFromQueue()) settings_.load_request.store(1);
				else settings_.load_request.store(0);
			}
			if (settings_.reset_request.load()) resetSim();
		}
		std::this_thread::sleep_for(std::chrono::milliseconds(1));
	}
	is_event_running_ = 0;
}

// mujoco_env.cpp:246-264 (the 100 ms ROS-message drain sleep has no counterpart here)
void MujocoEnv::resetSim()
{
	if (model_valid_) {
		backend_->reset(backend_->self, nullptr);
		loadInitialJointStates();
		double t = 0;
		backend_->get(backend_->self, MJB_F_time, 0, 1, &t);
		publishSimTime(t);
		for (auto &plugin : plugins_) plugin->safe_reset();
	}
	settings_.reset_request.store(0);
}

// mujoco_env.cpp:391-402, applied to every env instance
void MujocoEnv::setJointPosition(double pos, int joint_id, int jnt_axis)
{
	const int qa = model_.jnt_qposadr[joint_id] + jnt_axis;
	const int da = model_.jnt_dofadr[joint_id] + jnt_axis;
	for (int e = 0; e < nenv_; e++) {
		init_qpos_[(size_t)e * model_.nq + qa] = pos;
		if (da < model_.jnt_dofadr[joint_id] + (model_.jnt_type[joint_id] == 0 ? 6 : (model_.jnt_type[joint_id] == 1 ? 3 : 1))) {
			init_qvel_[(size_t)e * model_.nv + da] = 0;
			init_qfrc_[(size_t)e * model_.nv + da] = 0;
		}
	}
}
void MujocoEnv::setJointVelocity(double vel, int joint_id, int jnt_axis)
{
	const int da = model_.jnt_dofadr[joint_id] + jnt_axis;
	for (int e = 0; e < nenv_; e++) {
		init_qvel_[(size_t)e * model_.nv + da] = vel;
		init_qfrc_[(size_t)e * model_.nv + da] = 0;
	}
}

// mujoco_env.cpp:266-389: joint_map entries are STRINGS "v0 v1 ..." with exactly 7/4/1 (pos) or 6/3/1 (vel) values
void MujocoEnv::loadInitialJointStates()
{
	const bool has_pos = params_.has("initial_joint_positions/joint_map");
	const bool has_vel = params_.has("initial_joint_velocities/joint_map");
	if (!has_pos && !has_vel) return;
	init_qpos_.assign((size_t)nenv_ * model_.nq, 0);
	init_qvel_.assign((size_t)nenv_ * model_.nv, 0);
	init_qfrc_.assign((size_t)nenv_ * model_.nv, 0);
	backend_->get(backend_->self, MJB_F_qpos, 0, nenv_, init_qpos_.data());
	backend_->get(backend_->self, MJB_F_qvel, 0, nenv_, init_qvel_.data());
	backend_->get(backend_->self, MJB_F_qfrc_applied, 0, nenv_, init_qfrc_.data());
	auto apply = [&](const ConfigValue &map, bool is_pos) {
		if (map.getType() != ConfigValue::TypeStruct) return;
		for (const auto &kv : map.members()) {
			const int id = mj_name2id(&model_, mjOBJ_JOINT, kv.first.c_str());
			if (id == -1) {
				plugin_warnings_.push_back("Joint with name '" + kv.first + "' could not be found. Initial joint state cannot be set!");
				continue;
			}
			int num_axes = 0;
			switch (model_.jnt_type[id]) {
			case MJB_JNT_FREE: num_axes = is_pos ? 7 : 6; break;
			case MJB_JNT_BALL: num_axes = is_pos ? 4 : 3; break;
			default: num_axes = 1;
			}
			if (kv.second.getType() != ConfigValue::TypeString) {
				plugin_warnings_.push_back("Initial joint states must be provided as strings (joint " + kv.first + ")");
				continue;
			}
			std::vector<double> vals;
			std::stringstream ss(kv.second.asString());
			std::string tok;
			bool bad = false;
			while (std::getline(ss, tok, ' ')) {
				if (tok.empty()) continue;
				try {
					vals.push_back(std::stod(tok));
				} catch (...) {
					bad = true;
				}
			}
			if (bad || (int)vals.size() != num_axes) {
				plugin_warnings_.push_back("Provided initial values for joint " + kv.first + " don't match the degrees of freedom of the joint");
				continue;
			}
			for (int a = 0; a < num_axes; a++) {
				if (is_pos) setJointPosition(vals[a], id, a);
				else setJointVelocity(vals[a], id, a);
			}
		}
	};
	if (has_pos) apply(params_.get("initial_joint_positions/joint_map"), true);
	if (has_vel) apply(params_.get("initial_joint_velocities/joint_map"), false);
	backend_->set(backend_->self, MJB_F_qpos, 0, nenv_, init_qpos_.data());
	backend_->set(backend_->self, MJB_F_qvel, 0, nenv_, init_qvel_.data());
	backend_->set(backend_->self, MJB_F_qfrc_applied, 0, nenv_, init_qfrc_.data());
	backend_->forward(backend_->self);  // "Apply changes in forward dynamics" (:329) -- also normalises quaternions
}

// ------------------------------------------------------------------------------------ requests
// mujoco_env.cpp:913-945
bool MujocoEnv::step(int num_steps, bool blocking)
{
	if (!model_valid_) return false;           // "No model loaded. Cannot step"
	if (settings_.run.load()) return false;    // "Simulation is already running. Ignoring request"
	if (num_steps <= 0) return false;          // "Number of steps must be positive"
	if (blocking && std::this_thread::get_id() == physics_thread_handle_.get_id()) return false;  // cannot block itself
	settings_.env_steps_request.store(num_steps);
	if (blocking)
		while (settings_.env_steps_request.load() > 0 && !settings_.exit_request.load())
			std::this_thread::sleep_for(std::chrono::milliseconds(1));
	return true;
}

// mujoco_env.cpp:716-731
bool MujocoEnv::togglePaused(bool paused, const std::string &admin_hash)
{
	if (settings_.eval_mode && paused) {
		if (std::string(settings_.admin_hash) != admin_hash) return false;  // "Unauthorized pause request detected"
	}
	settings_.run.store(!paused);
	if (settings_.run.load()) settings_.env_steps_request.store(0);
	return true;
}

// callbacks.cpp:94-129
MujocoEnv::StepResult MujocoEnv::onStepGoal(int num_steps, const std::atomic_bool *preempt, std::vector<int> *feedback)
{
	StepResult res;
	if (settings_.env_steps_request.load() > 0 || settings_.run.load()) {
		res.success = false;  // "Simulation is currently unpaused. Stepping makes no sense right now."
		res.preempted = true;
		return res;
	}
	if (feedback) feedback->push_back(num_steps + settings_.env_steps_request.load());
	settings_.env_steps_request.store(settings_.env_steps_request.load() + num_steps);
	res.success = true;
	while (settings_.env_steps_request.load() > 0) {
		if ((preempt && preempt->load()) || settings_.exit_request.load() > 0 || settings_.load_request.load() > 0 ||
		    settings_.reset_request.load() > 0) {
			res.success = false;
			res.preempted = true;
			settings_.env_steps_request.store(0);
			break;
		}
		if (feedback) feedback->push_back(settings_.env_steps_request.load());
		std::this_thread::sleep_for(std::chrono::milliseconds(1));
	}
	if (feedback) feedback->push_back(settings_.env_steps_request.load());
	return res;
}

MujocoEnv::ServiceResponse MujocoEnv::setPauseCB(bool paused, const std::string &admin_hash)
{
	ServiceResponse r;
	r.success = togglePaused(paused, admin_hash);
	return r;
}
MujocoEnv::ServiceResponse MujocoEnv::shutdownCB()
{
	settings_.exit_request.store(1);
	return ServiceResponse();
}
MujocoEnv::ServiceResponse MujocoEnv::resetCB()
{
	settings_.reset_request.store(1);
	return ServiceResponse();
}

}  // namespace mujoco_ros
