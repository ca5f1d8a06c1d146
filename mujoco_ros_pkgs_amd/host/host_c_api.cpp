// host_c_api.cpp — flat C face of mujoco_ros::MujocoEnv (include/mjr_host.h) + the libmjb-backed stepper.
#include <cstring>
#include <map>
#include <memory>
#include <algorithm>
#include <string>

#include "mujoco_env.h"
#include "test_plugin.h"
#include "sensors_plugin.h"

using namespace mujoco_ros;

namespace {
thread_local std::string g_err;

struct MjbBackend {
	mjb_model *model = nullptr;
	mjb_batch *batch = nullptr;
	mjr_backend vt{};
};
#define B(self) (static_cast<MjbBackend *>(self))
int be_nenv(void *s) { return mjb_nenv(B(s)->batch); }
int be_field_size(void *s, int f) { return mjb_field_size(B(s)->model, f); }
int be_step(void *s, int n) { int rc = mjb_step(B(s)->batch, n); return rc ? rc : mjb_synchronize(B(s)->batch); }
int be_step1(void *s) { return mjb_step1(B(s)->batch); }
int be_step2(void *s) { return mjb_step2(B(s)->batch); }
int be_forward(void *s) { return mjb_forward(B(s)->batch); }
int be_reset(void *s, const uint8_t *m) { return mjb_reset(B(s)->batch, m); }
int be_get(void *s, int f, int lo, int hi, double *h) { return mjb_get(B(s)->batch, f, lo, hi, h); }
int be_set(void *s, int f, int lo, int hi, const double *h) { return mjb_set(B(s)->batch, f, lo, hi, h); }
int be_noise(void *s, double a, double b, uint64_t seed, int64_t off) { return mjb_set_ctrl_noise(B(s)->batch, a, b, seed, off); }
int be_sync(void *s) { return mjb_synchronize(B(s)->batch); }
int be_get_many(void *s, int n, const int *f, int lo, int hi, double *const *h) { return mjb_get_many(B(s)->batch, n, f, lo, hi, h); }
int be_set_many(void *s, int n, const int *f, int lo, int hi, const double *const *h) { return mjb_set_many(B(s)->batch, n, f, lo, hi, h); }
int be_host_register(void *, void *h, unsigned long long bytes) { return mjb_host_register(h, bytes); }
int be_host_unregister(void *, void *h) { return mjb_host_unregister(h); }
int be_step_async(void *s, int n) { return mjb_step(B(s)->batch, n); }
int be_register_collision(void *s, int t1, int t2, int f) { return mjb_register_collision(B(s)->batch, t1, t2, f); }
int be_set_env_param(void *s, int what, int lo, int hi, const void *data)
{
	mjb_batch *b = B(s)->batch;
	switch (what) {
	case MJR_ENV_GRAVITY: return mjb_set_env_gravity(b, lo, hi, static_cast<const double *>(data));
	case MJR_ENV_GEOM_FRICTION: return mjb_set_env_geom_friction(b, lo, hi, static_cast<const double *>(data));
	case MJR_ENV_GEOM_SIZE: return mjb_set_env_geom_size(b, lo, hi, static_cast<const double *>(data));
	case MJR_ENV_GEOM_TYPE: return mjb_set_env_geom_type(b, lo, hi, static_cast<const int *>(data));
	case MJR_ENV_EQUALITY: return mjb_set_env_equality(b, lo, hi, static_cast<const double *>(data));
	case MJR_ENV_BODY_MASS: return mjb_set_env_body_mass(b, lo, hi, static_cast<const double *>(data), nullptr);
	default: return -1;
	}
}
int be_get_packed(void *s, int n, const int *f, int lo, int hi, double *h) { return mjb_get_packed(B(s)->batch, n, f, lo, hi, h); }
int be_set_packed(void *s, int n, const int *f, int lo, int hi, const double *h) { return mjb_set_packed(B(s)->batch, n, f, lo, hi, h); }
int be_step1_prefix(void *s, int ncb) { return mjb_step1_prefix(B(s)->batch, ncb); }
int be_step_rest(void *s, int ncb) { return mjb_step_rest(B(s)->batch, ncb); }
int be_step2_prefix(void *s, int ncb) { return mjb_step2_prefix(B(s)->batch, ncb); }
int be_step21_prefix(void *s, int ncb) { return mjb_step21_prefix(B(s)->batch, ncb); }
int be_step2_rk(void *s, int ncb, int rk) { return mjb_step2_rk_prefix(B(s)->batch, ncb < 0 ? mjb_nenv(B(s)->batch) : ncb, rk); }
const char *be_err(void *) { return mjb_last_error(); }
void be_destroy(void *s)
{
	MjbBackend *b = B(s);
	if (b->batch) mjb_free_batch(b->batch);
	if (b->model) mjb_free_model(b->model);
	delete b;
}
}  // namespace

// ---- composite backend: one child backend per env block / device --------------------------------------
struct ShardCfg {
	mjr_backend_factory inner = nullptr;
	void *inner_user = nullptr;
	std::vector<int> devices;
};
struct Sharded {
	std::vector<mjr_backend *> kid;
	std::vector<int> lo;  // first env of every block, plus the total at the end
	std::string err;
	size_t stride[6] = { 0, 0, 0, 0, 0, 0 };  // bytes per env of each MJR_ENV_* payload
	bool failed = false;                       // a launch failed on one shard: the blocks are out of step
	mjr_backend vt{};
};
#define SH(self) (static_cast<Sharded *>(self))
// a shard that fails leaves the blocks out of step with each other (the ones before it were already launched): the others are
// drained and the composite refuses every launch from then on
template <typename F> int sh_each(Sharded *s, F fn)
{
	if (s->failed) return -1;
	for (size_t k = 0; k < s->kid.size(); k++) {
		const int rc = fn(s->kid[k], (int)k);
		if (rc) {
			s->err = std::string("shard ") + std::to_string(k) + ": " + s->kid[k]->last_error(s->kid[k]->self) + " (the sharded backend is now disabled)";
			for (mjr_backend *q : s->kid) q->synchronize(q->self);
			s->failed = true;
			return rc;
		}
	}
	return 0;
}
// [lo, hi) of the whole batch -> per-block sub-ranges
template <typename F> int sh_range(Sharded *s, int lo, int hi, F fn)
{
	for (size_t k = 0; k < s->kid.size(); k++) {
		const int a = std::max(lo, s->lo[k]), b = std::min(hi, s->lo[k + 1]);
		if (a >= b) continue;
		const int rc = fn(s->kid[k], a - s->lo[k], b - s->lo[k], a - lo);
		if (rc) {
			s->err = s->kid[k]->last_error(s->kid[k]->self);
			return rc;
		}
	}
	return 0;
}
int sh_nenv(void *s) { return SH(s)->lo.back(); }
int sh_field_size(void *s, int f) { return SH(s)->kid[0]->field_size(SH(s)->kid[0]->self, f); }
int sh_sync(void *s) { return sh_each(SH(s), [](mjr_backend *k, int) { return k->synchronize(k->self); }); }
int sh_step(void *s, int n)
{
	// every block's launch is enqueued before any block is waited for
	int rc = sh_each(SH(s), [&](mjr_backend *k, int) { return k->step_async ? k->step_async(k->self, n) : k->step(k->self, n); });
	return rc ? rc : sh_sync(s);
}
int sh_step_async(void *s, int n)
{
	return sh_each(SH(s), [&](mjr_backend *k, int) { return k->step_async ? k->step_async(k->self, n) : k->step(k->self, n); });
}
int sh_step1(void *s) { return sh_each(SH(s), [](mjr_backend *k, int) { return k->step1(k->self); }); }
int sh_step2(void *s) { return sh_each(SH(s), [](mjr_backend *k, int) { return k->step2(k->self); }); }
int sh_forward(void *s) { return sh_each(SH(s), [](mjr_backend *k, int) { return k->forward(k->self); }); }
int sh_reset(void *s, const uint8_t *m)
{
	return sh_each(SH(s), [&](mjr_backend *k, int i) { return k->reset(k->self, m ? m + SH(s)->lo[i] : nullptr); });
}
int sh_get(void *s, int f, int lo, int hi, double *h)
{
	const int sz = sh_field_size(s, f);
	return sh_range(SH(s), lo, hi, [&](mjr_backend *k, int a, int b, int off) { return k->get(k->self, f, a, b, h + (size_t)off * sz); });
}
int sh_set(void *s, int f, int lo, int hi, const double *h)
{
	const int sz = sh_field_size(s, f);
	return sh_range(SH(s), lo, hi, [&](mjr_backend *k, int a, int b, int off) { return k->set(k->self, f, a, b, h + (size_t)off * sz); });
}
int sh_get_many(void *s, int n, const int *f, int lo, int hi, double *const *h)
{
	return sh_range(SH(s), lo, hi, [&](mjr_backend *k, int a, int b, int off) {
		std::vector<double *> p(n);
		for (int q = 0; q < n; q++) p[q] = h[q] + (size_t)off * std::max(0, sh_field_size(s, f[q]));
		if (k->get_many) return k->get_many(k->self, n, f, a, b, p.data());
		for (int q = 0; q < n; q++)
			if (int rc = k->get(k->self, f[q], a, b, p[q])) return rc;
		return 0;
	});
}
int sh_set_many(void *s, int n, const int *f, int lo, int hi, const double *const *h)
{
	return sh_range(SH(s), lo, hi, [&](mjr_backend *k, int a, int b, int off) {
		std::vector<const double *> p(n);
		for (int q = 0; q < n; q++) p[q] = h[q] + (size_t)off * std::max(0, sh_field_size(s, f[q]));
		if (k->set_many) return k->set_many(k->self, n, f, a, b, p.data());
		for (int q = 0; q < n; q++)
			if (int rc = k->set(k->self, f[q], a, b, p[q])) return rc;
		return 0;
	});
}
int sh_noise(void *s, double a, double b, uint64_t seed, int64_t off)
{
	// the Philox stream is keyed by the global env index: block i starts at off + lo[i]
	return sh_each(SH(s), [&](mjr_backend *k, int i) { return k->set_ctrl_noise(k->self, a, b, seed, off + SH(s)->lo[i]); });
}
int sh_host_register(void *s, void *h, unsigned long long bytes)
{
	mjr_backend *k = SH(s)->kid[0];
	return k->host_register ? k->host_register(k->self, h, bytes) : -1;
}
int sh_host_unregister(void *s, void *h)
{
	mjr_backend *k = SH(s)->kid[0];
	return k->host_unregister ? k->host_unregister(k->self, h) : 0;
}
int sh_register_collision(void *s, int t1, int t2, int fn)
{
	return sh_each(SH(s), [&](mjr_backend *k, int) { return k->register_collision ? k->register_collision(k->self, t1, t2, fn) : -1; });
}
int sh_set_env_param(void *s, int what, int lo, int hi, const void *data)
{
	// per-env stride of the payload, from the model's sizes (every block holds the same model)
	Sharded *sh = SH(s);
	const size_t per = sh->stride[what < 0 || what > 5 ? 0 : what];
	return sh_range(sh, lo, hi, [&](mjr_backend *k, int a, int b, int off) {
		return k->set_env_param ? k->set_env_param(k->self, what, a, b, static_cast<const char *>(data) + (size_t)off * per) : -1;
	});
}
// the callback prefix [0, ncb) of the whole batch, seen from block i: its first clamp(ncb - lo[i], 0, n_i) envs
template <typename F> int sh_prefix(Sharded *s, int ncb, F fn)
{
	return sh_each(s, [&](mjr_backend *k, int i) {
		const int n = s->lo[i + 1] - s->lo[i], local = std::max(0, std::min(n, ncb - s->lo[i]));
		return fn(k, local);
	});
}
int sh_step1_prefix(void *s, int ncb)
{
	return sh_prefix(SH(s), ncb, [](mjr_backend *k, int c) { return k->step1_prefix ? k->step1_prefix(k->self, c) : k->step1(k->self); });
}
int sh_step_rest(void *s, int ncb)
{
	return sh_prefix(SH(s), ncb, [](mjr_backend *k, int c) { return k->step_rest ? k->step_rest(k->self, c) : 0; });
}
int sh_step2_prefix(void *s, int ncb)
{
	return sh_prefix(SH(s), ncb, [](mjr_backend *k, int c) { return k->step2_prefix ? k->step2_prefix(k->self, c) : k->step2(k->self); });
}
const char *sh_err(void *s) { return SH(s)->err.c_str(); }
void sh_destroy(void *s)
{
	for (mjr_backend *k : SH(s)->kid) k->destroy(k->self);
	delete SH(s);
}
mjr_backend *sharded_factory(const mjb_model_desc *desc, int nenv, int, void *user)
{
	const ShardCfg *cfg = static_cast<const ShardCfg *>(user);
	const int nd = (int)cfg->devices.size();
	Sharded *sh = new Sharded;
	sh->lo.push_back(0);
	for (int i = 0; i < nd; i++) {
		const int n = nenv / nd + (i < nenv % nd ? 1 : 0);  // contiguous blocks, the remainder on the first ones
		if (n == 0) continue;
		mjr_backend *k = cfg->inner(desc, n, cfg->devices[i], cfg->inner_user);
		if (!k) {
			for (mjr_backend *q : sh->kid) q->destroy(q->self);
			delete sh;
			return nullptr;
		}
		sh->kid.push_back(k);
		sh->lo.push_back(sh->lo.back() + n);
	}
	sh->stride[MJR_ENV_GRAVITY] = 3 * sizeof(double);
	sh->stride[MJR_ENV_GEOM_FRICTION] = sh->stride[MJR_ENV_GEOM_SIZE] = (size_t)3 * desc->ngeom * sizeof(double);
	sh->stride[MJR_ENV_GEOM_TYPE] = (size_t)desc->ngeom * sizeof(int);
	sh->stride[MJR_ENV_EQUALITY] = (size_t)19 * desc->neq * sizeof(double);
	sh->stride[MJR_ENV_BODY_MASS] = (size_t)desc->nbody * sizeof(double);
	sh->vt = mjr_backend{ sh, sh_nenv, sh_field_size, sh_step, sh_step1, sh_step2, sh_forward, sh_reset, sh_get, sh_set, sh_noise,
		                  sh_sync, sh_err, sh_destroy, sh_get_many, sh_set_many, sh_host_register, sh_host_unregister, sh_step_async,
		                  sh_register_collision, sh_set_env_param, nullptr, nullptr, sh_step1_prefix, sh_step_rest, sh_step2_prefix, nullptr, nullptr };
	return &sh->vt;
}

struct mjr_env {
	MujocoEnv *env = nullptr;
	// one ShardCfg PER queued request (the deferred factory reads it on the event thread; a second queue_model_devices before the
	// first is served must not change what the first one sees); they live as long as the env
	std::vector<std::unique_ptr<ShardCfg>> shards;
};

extern "C" {

const char *mjr_last_error(void) { return g_err.c_str(); }
int mjr_model_desc_size(void) { return (int)sizeof(mjb_model_desc); }

mjr_backend *mjr_make_mjb_backend(const mjb_model_desc *desc, int nenv, int device, void *)
{
	MjbBackend *b = new MjbBackend;
	b->model = mjb_compile(desc);
	if (!b->model) {
		g_err = mjb_last_error();
		delete b;
		return nullptr;
	}
	b->batch = mjb_make_batch(b->model, nenv, device);
	if (!b->batch) {
		g_err = mjb_last_error();
		mjb_free_model(b->model);
		delete b;
		return nullptr;
	}
	b->vt = mjr_backend{ b, be_nenv, be_field_size, be_step, be_step1, be_step2, be_forward, be_reset, be_get, be_set,
		                 be_noise, be_sync, be_err, be_destroy, be_get_many, be_set_many, be_host_register, be_host_unregister,
		                 be_step_async, be_register_collision, be_set_env_param, be_get_packed, be_set_packed, be_step1_prefix, be_step_rest,
		                 be_step2_prefix, be_step21_prefix, be_step2_rk };
	return &b->vt;
}

mjr_env *mjr_env_create(const char *admin_hash, const char *params_json)
{
	try {
		ParamServer ps;
		if (params_json && params_json[0]) {
			ConfigValue v = ConfigValue::fromJson(params_json);
			if (v.getType() != ConfigValue::TypeStruct) throw std::runtime_error("params must be a JSON object");
			for (const auto &kv : v.members()) ps.set(kv.first, kv.second);
		}
		mjr_env *e = new mjr_env;
		e->env = new MujocoEnv(admin_hash ? admin_hash : "", &ps);
		return e;
	} catch (const std::exception &ex) {
		g_err = ex.what();
		return nullptr;
	}
}

void mjr_env_destroy(mjr_env *e)
{
	if (!e) return;
	delete e->env;
	delete e;
}

int mjr_env_set_param(mjr_env *e, const char *key, const char *json_value)
{
	try {
		e->env->params_.set(key, ConfigValue::fromJson(json_value));
		return 0;
	} catch (const std::exception &ex) {
		g_err = ex.what();
		return -1;
	}
}
int mjr_env_delete_param(mjr_env *e, const char *key)
{
	e->env->params_.erase(key);
	return 0;
}

int mjr_env_queue_model(mjr_env *e, const mjb_model_desc *desc, const mjr_names *names, int nenv, int device,
                        mjr_backend_factory factory, void *factory_user)
{
	if (!e || !desc || nenv <= 0) return -1;
	ModelNames n;
	auto fill = [](std::vector<std::string> &dst, const char *const *src, int cnt) {
		for (int i = 0; i < cnt; i++) dst.emplace_back(src && src[i] ? src[i] : "");
	};
	if (names) {
		fill(n.body, names->body, desc->nbody);
		fill(n.joint, names->joint, desc->njnt);
		fill(n.geom, names->geom, desc->ngeom);
		fill(n.site, names->site, desc->nsite);
		fill(n.sensor, names->sensor, desc->nsensor);
		fill(n.actuator, names->actuator, desc->nu);
		if (names->equality) fill(n.equality, names->equality, desc->neq);
		if (names->tendon) fill(n.tendon, names->tendon, desc->ntendon);
	}
	e->env->queueModel(desc, n, nenv, device, factory, factory_user);
	return 0;
}

int mjr_env_queue_model_devices(mjr_env *e, const mjb_model_desc *desc, const mjr_names *names, int nenv, const int *devices, int ndev,
                                mjr_backend_factory factory, void *factory_user)
{
	if (!e || !desc || !devices || ndev <= 0 || nenv < ndev) return -1;
	e->shards.emplace_back(new ShardCfg);
	ShardCfg *cfg = e->shards.back().get();
	cfg->inner = factory ? factory : mjr_make_mjb_backend;
	cfg->inner_user = factory_user;
	cfg->devices.assign(devices, devices + ndev);
	return mjr_env_queue_model(e, desc, names, nenv, devices[0], sharded_factory, cfg);
}

int mjr_env_start(mjr_env *e)
{
	e->env->startPhysicsLoop();
	e->env->startEventLoop();
	return 0;
}
int mjr_env_shutdown(mjr_env *e)
{
	e->env->shutdown();
	return 0;
}
int mjr_env_operational_status(mjr_env *e) { return e->env->getOperationalStatus(); }
int mjr_env_pending_steps(mjr_env *e) { return e->env->getPendingSteps(); }
int mjr_env_is_physics_running(mjr_env *e) { return e->env->isPhysicsRunning(); }
int mjr_env_is_event_running(mjr_env *e) { return e->env->isEventRunning(); }
int mjr_env_model_valid(mjr_env *e) { return e->env->getModelPtr() != nullptr; }
const char *mjr_env_load_error(mjr_env *e) { return e->env->loadError().c_str(); }
int mjr_env_step(mjr_env *e, int n, int blocking) { return e->env->step(n, blocking != 0) ? 1 : 0; }
int mjr_env_toggle_paused(mjr_env *e, int paused, const char *hash) { return e->env->togglePaused(paused != 0, hash ? hash : "") ? 1 : 0; }
int mjr_env_step_goal(mjr_env *e, int n, int *preempted)
{
	MujocoEnv::StepResult r = e->env->onStepGoal(n);
	if (preempted) *preempted = r.preempted;
	return r.success ? 1 : 0;
}
int mjr_env_reset_request(mjr_env *e)
{
	e->env->resetCB();
	return 1;
}
int mjr_env_set_pause(mjr_env *e, int paused, const char *hash) { return e->env->setPauseCB(paused != 0, hash ? hash : "").success ? 1 : 0; }

static std::atomic_int *setting(mjr_env *e, const char *name)
{
	auto &s = e->env->settings_;
	if (!strcmp(name, "run")) return &s.run;
	if (!strcmp(name, "exit_request")) return &s.exit_request;
	if (!strcmp(name, "load_request")) return &s.load_request;
	if (!strcmp(name, "reset_request")) return &s.reset_request;
	if (!strcmp(name, "env_steps_request")) return &s.env_steps_request;
	return nullptr;
}
int mjr_env_get_setting(mjr_env *e, const char *name)
{
	auto *a = setting(e, name);
	return a ? a->load() : -1;
}
int mjr_env_set_setting(mjr_env *e, const char *name, int value)
{
	auto *a = setting(e, name);
	if (!a) return -1;
	a->store(value);
	return 0;
}
int mjr_env_set_ctrl_noise(mjr_env *e, double std, double rate)
{
	std::lock_guard<MujocoEnvMutex> lock(e->env->physics_thread_mutex_);
	e->env->ctrl_noise_std = std;
	e->env->ctrl_noise_rate = rate;
	if (e->env->backend()) return e->env->backend()->set_ctrl_noise(e->env->backend()->self, std, rate, 12345, 0);
	return 0;
}
double mjr_env_sim_time(mjr_env *e) { return e->env->simTime(); }
double mjr_env_data_time(mjr_env *e) { return e->env->dataTime(); }
unsigned long long mjr_env_step_count(mjr_env *e) { return e->env->stepCount(); }
int mjr_env_nenv(mjr_env *e) { return e->env->nenv(); }
int mjr_env_name2id(mjr_env *e, int objtype, const char *name)
{
	const mjModel *m = e->env->getModelPtr();
	return m ? mj_name2id(m, objtype, name) : -1;
}
int mjr_env_get_field(mjr_env *e, int field, int env, double *out)
{
	std::lock_guard<MujocoEnvMutex> lock(e->env->physics_thread_mutex_);
	mjr_backend *b = e->env->backend();
	if (!b) return -1;
	return b->get(b->self, field, env, env + 1, out);
}
int mjr_env_set_field(mjr_env *e, int field, int env, const double *in)
{
	std::lock_guard<MujocoEnvMutex> lock(e->env->physics_thread_mutex_);
	mjr_backend *b = e->env->backend();
	if (!b) return -1;
	return b->set(b->self, field, env, env + 1, in);
}
int mjr_env_num_plugins(mjr_env *e) { return (int)e->env->getPlugins().size(); }
int mjr_env_num_cb_ready_plugins(mjr_env *e)
{
	int n = 0;
	for (const auto &p : e->env->getPlugins()) n += p->loadingSuccessful() ? 1 : 0;
	return n;
}
int mjr_env_test_plugin_flag(mjr_env *e, int i, const char *name, int clear)
{
	const auto &pl = e->env->getPlugins();
	if (i < 0 || i >= (int)pl.size()) return -1;
	TestPlugin *t = dynamic_cast<TestPlugin *>(pl[i].get());
	if (!t) return -1;
	std::map<std::string, std::atomic_bool *> flags = {
		{ "ran_reset", &t->ran_reset }, { "ran_control_cb", &t->ran_control_cb }, { "ran_passive_cb", &t->ran_passive_cb },
		{ "ran_render_cb", &t->ran_render_cb }, { "ran_last_cb", &t->ran_last_cb },
		{ "ran_on_geom_changed_cb", &t->ran_on_geom_changed_cb }, { "got_config_param", &t->got_config_param },
		{ "got_lvl1_nested_array", &t->got_lvl1_nested_array }, { "got_lvl2_nested_array", &t->got_lvl2_nested_array },
		{ "got_lvl1_nested_struct", &t->got_lvl1_nested_struct }, { "got_lvl2_nested_struct", &t->got_lvl2_nested_struct },
		{ "should_fail", &t->should_fail }
	};
	auto it = flags.find(name);
	if (it != flags.end()) {
		int v = it->second->load() ? 1 : 0;
		if (clear) it->second->store(false);
		return v;
	}
	if (!strcmp(name, "control_calls")) {
		int v = t->control_calls.load();
		if (clear) t->control_calls.store(0);
		return v;
	}
	if (!strcmp(name, "passive_calls")) {
		int v = t->passive_calls.load();
		if (clear) t->passive_calls.store(0);
		return v;
	}
	if (!strcmp(name, "last_calls")) {
		int v = t->last_calls.load();
		if (clear) t->last_calls.store(0);
		return v;
	}
	if (!strcmp(name, "last_env")) return t->last_env.load();
	return -1;
}
int mjr_env_notify_geom_changed(mjr_env *e, int geom_id)
{
	e->env->notifyGeomChanged(geom_id);
	return 0;
}
int mjr_env_register_collision_function(mjr_env *e, int t1, int t2, int func) { return e->env->registerCollisionFunction(t1, t2, func); }
int mjr_env_set_callback_envs(mjr_env *e, int n)
{
	e->env->setCallbackEnvs(n);
	return 0;
}

// ---- model / body-state services (host/services.cpp) ----
static void put_msg(char *msg, int cap, const std::string &s)
{
	if (msg && cap > 0) snprintf(msg, (size_t)cap, "%s", s.c_str());
}
static ModelNames names_of(const mjb_model_desc *desc, const mjr_names *names)
{
	ModelNames n;
	auto fill = [](std::vector<std::string> &dst, const char *const *src, int cnt) {
		for (int i = 0; i < cnt; i++) dst.emplace_back(src && src[i] ? src[i] : "");
	};
	if (names) {
		fill(n.body, names->body, desc->nbody);
		fill(n.joint, names->joint, desc->njnt);
		fill(n.geom, names->geom, desc->ngeom);
		fill(n.site, names->site, desc->nsite);
		fill(n.sensor, names->sensor, desc->nsensor);
		fill(n.actuator, names->actuator, desc->nu);
		if (names->equality) fill(n.equality, names->equality, desc->neq);
		if (names->tendon) fill(n.tendon, names->tendon, desc->ntendon);
	}
	return n;
}
int mjr_env_set_body_state(mjr_env *e, const mjr_body_state *st, int set_pose, int set_twist, int set_mass, int reset_qpos,
                           const char *admin_hash, int env_lo, int env_hi, char *msg, int msg_cap)
{
	if (!e || !st) return -1;
	BodyState b;
	b.name = st->name;
	b.mass = st->mass;
	for (int k = 0; k < 7; k++) b.pose[k] = st->pose[k];
	for (int k = 0; k < 6; k++) b.twist[k] = st->twist[k];
	b.pose_frame = st->pose_frame;
	b.twist_frame = st->twist_frame;
	auto r = e->env->setBodyStateCB(b, set_pose != 0, set_twist != 0, set_mass != 0, reset_qpos != 0, admin_hash ? admin_hash : "", env_lo, env_hi);
	put_msg(msg, msg_cap, r.status_message);
	return r.success ? 1 : 0;
}
int mjr_env_get_body_state(mjr_env *e, const char *name, const char *admin_hash, int env, mjr_body_state *out, char *msg, int msg_cap)
{
	if (!e || !name || !out) return -1;
	auto r = e->env->getBodyStateCB(name, admin_hash ? admin_hash : "", env);
	put_msg(msg, msg_cap, r.status_message);
	memset(out, 0, sizeof(*out));
	snprintf(out->name, sizeof(out->name), "%s", r.state.name.c_str());
	snprintf(out->pose_frame, sizeof(out->pose_frame), "%s", r.state.pose_frame.c_str());
	snprintf(out->twist_frame, sizeof(out->twist_frame), "%s", r.state.twist_frame.c_str());
	out->mass = r.state.mass;
	for (int k = 0; k < 7; k++) out->pose[k] = r.state.pose[k];
	for (int k = 0; k < 6; k++) out->twist[k] = r.state.twist[k];
	return r.success ? 1 : 0;
}
int mjr_env_set_geom_properties(mjr_env *e, const mjr_geom_properties *p, int set_type, int set_mass, int set_friction, int set_size,
                                const char *admin_hash, int env_lo, int env_hi, char *msg, int msg_cap)
{
	if (!e || !p) return -1;
	GeomProperties g;
	g.name = p->name;
	g.type = p->type;
	g.body_mass = p->body_mass;
	for (int k = 0; k < 3; k++) { g.friction[k] = p->friction[k]; g.size[k] = p->size[k]; }
	auto r = e->env->setGeomPropertiesCB(g, set_type != 0, set_mass != 0, set_friction != 0, set_size != 0, admin_hash ? admin_hash : "", env_lo, env_hi);
	put_msg(msg, msg_cap, r.status_message);
	return r.success ? 1 : 0;
}
int mjr_env_get_geom_properties(mjr_env *e, const char *geom_name, const char *admin_hash, int env, mjr_geom_properties *out, char *msg,
                                int msg_cap)
{
	if (!e || !geom_name || !out) return -1;
	auto r = e->env->getGeomPropertiesCB(geom_name, admin_hash ? admin_hash : "", env);
	put_msg(msg, msg_cap, r.status_message);
	memset(out, 0, sizeof(*out));
	snprintf(out->name, sizeof(out->name), "%s", r.properties.name.c_str());
	out->type = r.properties.type;
	out->body_mass = r.properties.body_mass;
	for (int k = 0; k < 3; k++) { out->friction[k] = r.properties.friction[k]; out->size[k] = r.properties.size[k]; }
	return r.success ? 1 : 0;
}
int mjr_env_set_gravity(mjr_env *e, const double *g, const char *admin_hash, int env_lo, int env_hi, char *msg, int msg_cap)
{
	if (!e || !g) return -1;
	auto r = e->env->setGravityCB(g, admin_hash ? admin_hash : "", env_lo, env_hi);
	put_msg(msg, msg_cap, r.status_message);
	return r.success ? 1 : 0;
}
int mjr_env_get_gravity(mjr_env *e, const char *admin_hash, int env, double *g, char *msg, int msg_cap)
{
	if (!e || !g) return -1;
	auto r = e->env->getGravityCB(admin_hash ? admin_hash : "", env);
	put_msg(msg, msg_cap, r.status_message);
	for (int k = 0; k < 3; k++) g[k] = r.gravity[k];
	return r.success ? 1 : 0;
}
static EqualityConstraintParameters eq_in(const mjr_eq_parameters &p)
{
	EqualityConstraintParameters q;
	q.name = p.name; q.element1 = p.element1; q.element2 = p.element2;
	q.type = p.type;
	q.active = p.active != 0;
	for (int k = 0; k < 3; k++) q.anchor[k] = p.anchor[k];
	for (int k = 0; k < 7; k++) q.relpose[k] = p.relpose[k];
	q.torquescale = p.torquescale;
	for (int k = 0; k < 5; k++) q.polycoef[k] = p.polycoef[k];
	q.solverParameters.dmin = p.dmin; q.solverParameters.dmax = p.dmax; q.solverParameters.width = p.width;
	q.solverParameters.midpoint = p.midpoint; q.solverParameters.power = p.power; q.solverParameters.timeconst = p.timeconst;
	q.solverParameters.dampratio = p.dampratio;
	return q;
}
static void eq_out(const EqualityConstraintParameters &q, mjr_eq_parameters *p)
{
	memset(p, 0, sizeof(*p));
	snprintf(p->name, sizeof(p->name), "%s", q.name.c_str());
	snprintf(p->element1, sizeof(p->element1), "%s", q.element1.c_str());
	snprintf(p->element2, sizeof(p->element2), "%s", q.element2.c_str());
	p->type = q.type;
	p->active = q.active ? 1 : 0;
	for (int k = 0; k < 3; k++) p->anchor[k] = q.anchor[k];
	for (int k = 0; k < 7; k++) p->relpose[k] = q.relpose[k];
	p->torquescale = q.torquescale;
	for (int k = 0; k < 5; k++) p->polycoef[k] = q.polycoef[k];
	p->dmin = q.solverParameters.dmin; p->dmax = q.solverParameters.dmax; p->width = q.solverParameters.width;
	p->midpoint = q.solverParameters.midpoint; p->power = q.solverParameters.power; p->timeconst = q.solverParameters.timeconst;
	p->dampratio = q.solverParameters.dampratio;
}
int mjr_env_set_eq_parameters(mjr_env *e, const mjr_eq_parameters *params, int n, const char *admin_hash, int env_lo, int env_hi, char *msg,
                              int msg_cap)
{
	if (!e || (n > 0 && !params)) return -1;
	std::vector<EqualityConstraintParameters> v;
	for (int k = 0; k < n; k++) v.push_back(eq_in(params[k]));
	auto r = e->env->setEqualityConstraintParametersArrayCB(v, admin_hash ? admin_hash : "", env_lo, env_hi);
	put_msg(msg, msg_cap, r.status_message);
	return r.success ? 1 : 0;
}
int mjr_env_get_eq_parameters(mjr_env *e, const char *const *names, int n, const char *admin_hash, int env, mjr_eq_parameters *out, int *nout,
                              char *msg, int msg_cap)
{
	if (!e || (n > 0 && (!names || !out))) return -1;
	std::vector<std::string> v;
	for (int k = 0; k < n; k++) v.emplace_back(names[k] ? names[k] : "");
	auto r = e->env->getEqualityConstraintParametersArrayCB(v, admin_hash ? admin_hash : "", env);
	put_msg(msg, msg_cap, r.status_message);
	for (size_t k = 0; k < r.parameters.size() && (int)k < n; k++) eq_out(r.parameters[k], out + k);
	if (nout) *nout = (int)r.parameters.size();
	return r.success ? 1 : 0;
}
int mjr_env_reload(mjr_env *e, const mjb_model_desc *desc, const mjr_names *names, int nenv, int device, mjr_backend_factory factory,
                   void *factory_user, char *msg, int msg_cap)
{
	if (!e) return -1;
	ModelNames n = desc ? names_of(desc, names) : ModelNames();
	auto r = e->env->reloadCB(desc, n, nenv, device, factory, factory_user);
	put_msg(msg, msg_cap, r.status_message);
	return r.success ? 1 : 0;
}
int mjr_env_loading_request_state(mjr_env *e, char *description, int cap)
{
	if (!e) return -1;
	auto st = e->env->getLoadingRequestState();
	put_msg(description, cap, st.description);
	return st.value;
}
int mjr_env_load_initial_joint_states(mjr_env *e)
{
	if (!e) return -1;
	return e->env->loadInitialJointStatesCB().success ? 1 : 0;
}

// ---- MujocoRosSensorsPlugin accessors (sensors_plugin.h) ----
static mujoco_ros::sensors::MujocoRosSensorsPlugin *sensors_plugin(mjr_env *e, int i)
{
	const auto &pl = e->env->getPlugins();
	if (i < 0 || i >= (int)pl.size()) return nullptr;
	return dynamic_cast<mujoco_ros::sensors::MujocoRosSensorsPlugin *>(pl[i].get());
}
int mjr_sensors_num_records(mjr_env *e, int plugin, int env)
{
	auto *p = sensors_plugin(e, plugin);
	return p ? (int)p->records(env).size() : -1;
}
int mjr_sensors_get_record(mjr_env *e, int plugin, int env, int k, mjr_sensor_record *out)
{
	auto *p = sensors_plugin(e, plugin);
	if (!p || !out) return -1;
	const auto &r = p->records(env);
	if (k < 0 || k >= (int)r.size()) return -1;
	memset(out, 0, sizeof(*out));
	snprintf(out->name, sizeof(out->name), "%s", r[k].name.c_str());
	snprintf(out->frame_id, sizeof(out->frame_id), "%s", r[k].frame_id.c_str());
	out->kind = r[k].kind;
	out->env = r[k].env;
	out->has_truth = r[k].has_truth ? 1 : 0;
	out->stamp = r[k].stamp;
	for (int c = 0; c < 4; c++) {
		out->value[c] = r[k].value[c];
		out->truth[c] = r[k].truth[c];
	}
	return 0;
}
int mjr_sensors_register_noise(mjr_env *e, int plugin, const char *sensor_name, int set_flag, const double *mean, const double *std,
                               const char *admin_hash)
{
	auto *p = sensors_plugin(e, plugin);
	if (!p) return -1;
	return p->registerNoiseModel(sensor_name ? sensor_name : "", (unsigned char)set_flag, mean, std, admin_hash ? admin_hash : "") ? 1 : 0;
}

}  // extern "C"
