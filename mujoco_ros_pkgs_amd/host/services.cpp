// services.cpp — the reference's model- and body-state service handlers for a BATCH of env instances.
//
// Mirrors /root/reference mujoco_ros/src/callbacks.cpp: reloadCB :177-201, setBodyStateCB :210-370, getBodyStateCB :372-460,
// setGravityCB / getGravityCB :462-506, setGeomPropertiesCB / getGeomPropertiesCB :508-639, set / get
// EqualityConstraintParameters(ArrayCB) :641-897, the get_loading_request_state / load_initial_joint_states lambdas :66-87.
// Same order of checks, same `success` / `status_message` outcomes (handlers never fail the call itself), same eval-mode
// admin-hash gate.  What differs by construction: the reference edits ONE mjModel / mjData in place; here every handler
// addresses an env range and the edits travel to the device as per-env parameter overrides (mjr_backend::set_env_param ->
// mjb_set_env_*), `mj_setConst` included (the backend derives the constants, mjb_set_env_body_mass).  Frames other than "world"
// cannot be transformed (no tf buffer): they take the reference's failed-transform branch.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <thread>

#include "mujoco_env.h"

namespace mujoco_ros {

namespace {
enum { mjEQ_CONNECT = 0, mjEQ_WELD = 1, mjEQ_JOINT = 2, mjEQ_TENDON = 3 };
const int mjNEQDATA = 11, mjNIMP = 5, mjNREF = 2, kEqStride = 19;  // active | eq_data[11] | solref[2] | solimp[5]
}  // namespace

void MujocoEnv::ensureMirrors()
{
	const mjb_model_desc &d = current_.desc;
	const size_t n = (size_t)nenv_;
	if (env_gravity_.empty()) {
		env_gravity_.resize(n * 3);
		for (size_t e = 0; e < n; e++)
			for (int k = 0; k < 3; k++) env_gravity_[3 * e + k] = d.gravity[k];
	}
	if (env_body_mass_.empty() && d.nbody > 0) {
		env_body_mass_.resize(n * d.nbody);
		for (size_t e = 0; e < n; e++) std::copy(d.body_mass, d.body_mass + d.nbody, env_body_mass_.begin() + e * d.nbody);
	}
	if (env_geom_friction_.empty() && d.ngeom > 0) {
		env_geom_friction_.resize(n * d.ngeom * 3);
		env_geom_size_.resize(n * d.ngeom * 3);
		env_geom_type_.resize(n * d.ngeom);
		for (size_t e = 0; e < n; e++) {
			std::copy(d.geom_friction, d.geom_friction + 3 * d.ngeom, env_geom_friction_.begin() + e * 3 * d.ngeom);
			std::copy(d.geom_size, d.geom_size + 3 * d.ngeom, env_geom_size_.begin() + e * 3 * d.ngeom);
			std::copy(d.geom_type, d.geom_type + d.ngeom, env_geom_type_.begin() + e * d.ngeom);
		}
	}
	if (env_equality_.empty() && d.neq > 0) {
		env_equality_.resize(n * d.neq * kEqStride);
		for (size_t e = 0; e < n; e++)
			for (int q = 0; q < d.neq; q++) {
				double *o = env_equality_.data() + (e * d.neq + q) * kEqStride;
				o[0] = d.eq_active[q] ? 1.0 : 0.0;
				for (int k = 0; k < mjNEQDATA; k++) o[1 + k] = d.eq_data[mjNEQDATA * q + k];
				for (int k = 0; k < mjNREF; k++) o[12 + k] = d.eq_solref[mjNREF * q + k];
				for (int k = 0; k < mjNIMP; k++) o[14 + k] = d.eq_solimp[mjNIMP * q + k];
			}
	}
}

int MujocoEnv::pushEnvParam(int what, int lo, int hi, const void *data, std::string &err)
{
	if (!backend_->set_env_param) {
		err = "the step backend has no per-env model parameters";
		return -1;
	}
	const int rc = backend_->set_env_param(backend_->self, what, lo, hi, data);
	if (rc != 0) err = backend_->last_error(backend_->self);
	return rc;
}

// `model_->body_mass[id] = mass; mj_setConst(...)` with qpos saved and restored around it (callbacks.cpp:244-258, :577-587): the
// state never leaves the device here, so there is nothing to save; the backend derives the constants from the new masses.
void MujocoEnv::applyMassChange(int lo, int hi, std::string &err)
{
	pushEnvParam(MJR_ENV_BODY_MASS, lo, hi, env_body_mass_.data() + (size_t)lo * current_.desc.nbody, err);
}

// ------------------------------------------------------------------------------------ reload / loading state (:66-87, :177-201)
MujocoEnv::ServiceResponse MujocoEnv::reloadCB(const mjb_model_desc *desc, const ModelNames &names, int nenv, int device,
                                               mjr_backend_factory factory, void *factory_user)
{
	ServiceResponse res;
	if (!desc || nenv <= 0) {  // (the reference's only request-shape check is the filename length, :181-188)
		res.success = false;
		res.status_message = "Model description missing or empty batch";
		return res;
	}
	queueModel(desc, names, nenv, device, factory, factory_user);  // settings_.load_request = 2
	while (getOperationalStatus() > 0 && !settings_.exit_request.load()) std::this_thread::sleep_for(std::chrono::milliseconds(5));
	res.success = sim_state_.model_valid && load_error_.empty();
	res.status_message = load_error_;
	return res;
}

MujocoEnv::LoadingRequestState MujocoEnv::getLoadingRequestState()
{
	LoadingRequestState st;
	st.value = getOperationalStatus();
	if (st.value == 0) st.description = "Sim ready";
	else if (st.value == 1) st.description = "Loading in progress";
	else st.description = "Loading issued";
	return st;
}

MujocoEnv::ServiceResponse MujocoEnv::loadInitialJointStatesCB()
{
	std::lock_guard<MujocoEnvMutex> lock(physics_thread_mutex_);
	if (model_valid_) loadInitialJointStates();
	return ServiceResponse();
}

// ------------------------------------------------------------------------------------ setBodyStateCB (:210-370)
MujocoEnv::ServiceResponse MujocoEnv::setBodyStateCB(BodyState state, bool set_pose, bool set_twist, bool set_mass, bool reset_qpos,
                                                     const std::string &admin_hash, int env_lo, int env_hi)
{
	ServiceResponse resp;
	if (!authorized(admin_hash)) {
		resp.success = false;
		resp.status_message = "Hash mismatch, no permission to set body state!";
		return resp;
	}
	if (!model_valid_) {
		resp.success = false;
		resp.status_message = "No model loaded";
		return resp;
	}
	std::string full_error_msg;
	resp.success = true;
	int body_id = mj_name2id(&model_, mjOBJ_BODY, state.name.c_str());
	if (body_id == -1) {  // "Trying to find geom..."
		const int geom_id = mj_name2id(&model_, mjOBJ_GEOM, state.name.c_str());
		if (geom_id == -1) {
			resp.status_message = "Could not find model (not body nor geom) with name " + state.name;
			resp.success = false;
			return resp;
		}
		body_id = model_.geom_bodyid[geom_id];
	}
	std::lock_guard<MujocoEnvMutex> lk_sim(physics_thread_mutex_);
	const int lo = std::max(0, env_lo), hi = env_hi < 0 ? nenv_ : std::min(env_hi, nenv_);
	if (lo >= hi) {
		resp.success = false;
		resp.status_message = "Empty env range";
		return resp;
	}
	if (set_mass) {
		ensureMirrors();
		for (int e = lo; e < hi; e++) env_body_mass_[(size_t)e * model_.nbody + body_id] = state.mass;
		std::string err;
		applyMassChange(lo, hi, err);
		if (!err.empty()) {
			full_error_msg += "Could not apply the new mass: " + err + '\n';
			resp.success = false;
		}
	}
	const int num_jnt = model_.body_jntnum[body_id];
	const int jnt_adr = num_jnt > 0 ? model_.body_jntadr[body_id] : -1;  // (MuJoCo stores -1 for a body without joints)
	const int jnt_type = jnt_adr >= 0 ? model_.jnt_type[jnt_adr] : -1;

	if (set_pose || set_twist || reset_qpos) {
		if (jnt_adr == -1) {
			full_error_msg += std::string("Body has no joints, cannot move body!") + '\n';
			resp.success = false;
		} else if (jnt_type != MJB_JNT_FREE) {
			full_error_msg += "Body " + state.name + " has no joint of type 'freetype'. This service call does not support any other types!" + '\n';
			resp.success = false;
		} else if (num_jnt > 1) {
			full_error_msg += "Body " + state.name + " has more than one joint ('" + std::to_string(num_jnt) +
			                  "'), pose/twist changes to bodies with more than one joint are not supported!" + '\n';
			resp.success = false;
		} else {
			const int jnt_qposadr = model_.jnt_qposadr[jnt_adr], jnt_dofadr = model_.jnt_dofadr[jnt_adr];
			std::vector<double> qpos((size_t)(hi - lo) * model_.nq), qvel((size_t)(hi - lo) * model_.nv);
			backend_->get(backend_->self, MJB_F_qpos, lo, hi, qpos.data());
			backend_->get(backend_->self, MJB_F_qvel, lo, hi, qvel.data());
			bool write_pos = false, write_vel = false;
			if (set_pose && !reset_qpos) {
				bool valid_pose = true;
				if (!state.pose_frame.empty() && state.pose_frame != "world") {  // would need tf_bufferPtr_->transform (:297-304)
					full_error_msg += "Could not transform frame '" + state.pose_frame + "' to frame world" + '\n';
					resp.success = false;
					valid_pose = false;
				}
				if (valid_pose) {
					double quat[4] = { state.pose[3], state.pose[4], state.pose[5], state.pose[6] };
					double nrm = std::sqrt(quat[0] * quat[0] + quat[1] * quat[1] + quat[2] * quat[2] + quat[3] * quat[3]);
					if (nrm < 1e-15) {  // mju_normalize4 of a zero quaternion gives the identity
						quat[0] = 1; quat[1] = quat[2] = quat[3] = 0;
					} else {
						for (double &q : quat) q /= nrm;
					}
					for (int e = 0; e < hi - lo; e++) {
						double *q = qpos.data() + (size_t)e * model_.nq + jnt_qposadr;
						q[0] = state.pose[0]; q[1] = state.pose[1]; q[2] = state.pose[2];
						q[3] = quat[0]; q[4] = quat[1]; q[5] = quat[2]; q[6] = quat[3];
					}
					write_pos = true;
				}
			}
			if (reset_qpos && num_jnt > 0) {  // "reset_qpos will overwrite the custom pose"
				for (int e = 0; e < hi - lo; e++)
					for (int k = 0; k < 7; k++) qpos[(size_t)e * model_.nq + jnt_qposadr + k] = model_.qpos0[jnt_qposadr + k];
				write_pos = true;
				if (!set_twist) {  // "Reset twist if no desired twist is given"
					set_twist = true;
					for (double &t : state.twist) t = 0;
					state.twist_frame.clear();
				}
			}
			if (set_twist) {
				if (!state.twist_frame.empty() && state.twist_frame != "world") {
					full_error_msg += std::string("Transforming twists from other frames is not supported! Not setting twist.") + '\n';
					resp.success = false;
				} else {
					for (int e = 0; e < hi - lo; e++)
						for (int k = 0; k < 6; k++) qvel[(size_t)e * model_.nv + jnt_dofadr + k] = state.twist[k];
					write_vel = true;
				}
			}
			if (write_pos) backend_->set(backend_->self, MJB_F_qpos, lo, hi, qpos.data());
			if (write_vel) backend_->set(backend_->self, MJB_F_qvel, lo, hi, qvel.data());
		}
	}
	resp.status_message = full_error_msg;
	return resp;
}

// ------------------------------------------------------------------------------------ getBodyStateCB (:372-460)
MujocoEnv::GetBodyStateResponse MujocoEnv::getBodyStateCB(const std::string &name, const std::string &admin_hash, int env)
{
	GetBodyStateResponse resp;
	if (!authorized(admin_hash)) {
		resp.status_message = "Hash mismatch, no permission to get body state!";
		resp.success = false;
		return resp;
	}
	if (!model_valid_ || env < 0 || env >= nenv_) {
		resp.success = false;
		resp.status_message = "No model loaded or env out of range";
		return resp;
	}
	resp.success = true;
	int body_id = mj_name2id(&model_, mjOBJ_BODY, name.c_str());
	if (body_id == -1) {
		const int geom_id = mj_name2id(&model_, mjOBJ_GEOM, name.c_str());
		if (geom_id == -1) {
			resp.status_message = "Could not find model (not body nor geom) with name " + name;
			resp.success = false;
			return resp;
		}
		body_id = model_.geom_bodyid[geom_id];
	}
	std::lock_guard<MujocoEnvMutex> lk_sim(physics_thread_mutex_);  // "Stop sim to get data out of the same point in time"
	ensureMirrors();
	resp.state.name = model_.body_names[body_id];
	resp.state.mass = env_body_mass_[(size_t)env * model_.nbody + body_id];
	const int num_jnt = model_.body_jntnum[body_id];
	const int jnt_adr = num_jnt > 0 ? model_.body_jntadr[body_id] : -1;
	const int jnt_type = jnt_adr >= 0 ? model_.jnt_type[jnt_adr] : -1;
	resp.state.pose_frame = "world";
	resp.state.twist_frame = "world";
	if (jnt_adr == -1 || jnt_type != MJB_JNT_FREE || num_jnt > 1) {
		// Cartesian pose / velocity of the body frame.  (The reference indexes xquat with a stride of 3, callbacks.cpp:420-423 --
		// a slip: xquat is nbody x 4; the quaternion of THIS body is what the message is documented to carry.)
		backend_->forward(backend_->self);
		const int fl[] = { MJB_F_xpos, MJB_F_xquat, MJB_F_cvel };
		pullFields(fl, 3, env, env + 1);
		const mjData &d = views_[env];
		for (int k = 0; k < 3; k++) resp.state.pose[k] = d.xpos[3 * body_id + k];
		for (int k = 0; k < 4; k++) resp.state.pose[3 + k] = d.xquat[4 * body_id + k];
		for (int k = 0; k < 6; k++) resp.state.twist[k] = d.cvel[6 * body_id + k];
	} else {
		const int qa = model_.jnt_qposadr[jnt_adr], da = model_.jnt_dofadr[jnt_adr];
		std::vector<double> qpos(model_.nq), qvel(model_.nv);
		backend_->get(backend_->self, MJB_F_qpos, env, env + 1, qpos.data());
		backend_->get(backend_->self, MJB_F_qvel, env, env + 1, qvel.data());
		for (int k = 0; k < 7; k++) resp.state.pose[k] = qpos[qa + k];
		for (int k = 0; k < 6; k++) resp.state.twist[k] = qvel[da + k];
	}
	return resp;
}

// ------------------------------------------------------------------------------------ gravity (:462-506)
MujocoEnv::ServiceResponse MujocoEnv::setGravityCB(const double gravity[3], const std::string &admin_hash, int env_lo, int env_hi)
{
	ServiceResponse resp;
	if (!authorized(admin_hash)) {
		resp.status_message = "Hash mismatch, no permission to set gravity!";
		resp.success = false;
		return resp;
	}
	std::lock_guard<MujocoEnvMutex> lk_sim(physics_thread_mutex_);
	if (!model_valid_) {
		resp.success = false;
		resp.status_message = "No model loaded";
		return resp;
	}
	const int lo = std::max(0, env_lo), hi = env_hi < 0 ? nenv_ : std::min(env_hi, nenv_);
	ensureMirrors();
	for (int e = lo; e < hi; e++)
		for (int k = 0; k < 3; k++) env_gravity_[3 * (size_t)e + k] = gravity[k];
	if (lo == 0 && hi == nenv_)
		for (int k = 0; k < 3; k++) model_.opt.gravity[k] = gravity[k];  // the shared view shows what every env has
	std::string err;
	if (lo < hi && pushEnvParam(MJR_ENV_GRAVITY, lo, hi, env_gravity_.data() + 3 * (size_t)lo, err) != 0) {
		resp.success = false;
		resp.status_message = err;
	}
	return resp;
}

MujocoEnv::GetGravityResponse MujocoEnv::getGravityCB(const std::string &admin_hash, int env)
{
	GetGravityResponse resp;
	if (!authorized(admin_hash)) {
		resp.status_message = "Hash mismatch, no permission to get gravity!";
		resp.success = false;
		return resp;
	}
	std::lock_guard<MujocoEnvMutex> lk_sim(physics_thread_mutex_);
	if (!model_valid_ || env < 0 || env >= nenv_) {
		resp.success = false;
		resp.status_message = "No model loaded or env out of range";
		return resp;
	}
	ensureMirrors();
	for (int k = 0; k < 3; k++) resp.gravity[k] = env_gravity_[3 * (size_t)env + k];
	return resp;
}

// ------------------------------------------------------------------------------------ geom properties (:508-639)
MujocoEnv::ServiceResponse MujocoEnv::setGeomPropertiesCB(const GeomProperties &p, bool set_type, bool set_mass, bool set_friction,
                                                          bool set_size, const std::string &admin_hash, int env_lo, int env_hi)
{
	ServiceResponse resp;
	if (!authorized(admin_hash)) {
		resp.status_message = "Hash mismatch, no permission to set geom properties!";
		resp.success = false;
		return resp;
	}
	if (!model_valid_) {
		resp.success = false;
		resp.status_message = "No model loaded";
		return resp;
	}
	const int geom_id = mj_name2id(&model_, mjOBJ_GEOM, p.name.c_str());
	if (geom_id == -1) {
		resp.status_message = "Could not find model (mujoco geom) with name " + p.name;
		resp.success = false;
		return resp;
	}
	const int body_id = model_.geom_bodyid[geom_id];
	std::string err, note;
	{
		std::lock_guard<MujocoEnvMutex> lk_sim(physics_thread_mutex_);
		const int lo = std::max(0, env_lo), hi = env_hi < 0 ? nenv_ : std::min(env_hi, nenv_);
		if (lo >= hi) {
			resp.success = false;
			resp.status_message = "Empty env range";
			return resp;
		}
		ensureMirrors();
		const int ng = model_.ngeom;
		if (set_mass)
			for (int e = lo; e < hi; e++) env_body_mass_[(size_t)e * model_.nbody + body_id] = p.body_mass;
		if (set_friction) {
			for (int e = lo; e < hi; e++)
				for (int k = 0; k < 3; k++) env_geom_friction_[((size_t)e * ng + geom_id) * 3 + k] = p.friction[k];
			if (pushEnvParam(MJR_ENV_GEOM_FRICTION, lo, hi, env_geom_friction_.data() + (size_t)lo * ng * 3, err) != 0) note += err + '\n';
		}
		if (set_type) {
			for (int e = lo; e < hi; e++) env_geom_type_[(size_t)e * ng + geom_id] = p.type;
			if (pushEnvParam(MJR_ENV_GEOM_TYPE, lo, hi, env_geom_type_.data() + (size_t)lo * ng, err) != 0) note += err + '\n';
			else if (p.type != MJB_GEOM_PLANE && p.type != MJB_GEOM_SPHERE && p.type != MJB_GEOM_CAPSULE && p.type != MJB_GEOM_BOX)
				note += "geom type " + std::to_string(p.type) + " has no pair function in the step engine: the geom yields no contacts\n";
		}
		if (set_size) {
			// ("New geom size is bigger than the old size. AABBs are not recomputed" -- neither are the engine's bounding radii)
			for (int e = lo; e < hi; e++)
				for (int k = 0; k < 3; k++) env_geom_size_[((size_t)e * ng + geom_id) * 3 + k] = p.size[k];
			if (pushEnvParam(MJR_ENV_GEOM_SIZE, lo, hi, env_geom_size_.data() + (size_t)lo * ng * 3, err) != 0) note += err + '\n';
			backend_->forward(backend_->self);  // mj_forward (:573)
		}
		if (set_type || set_mass) {  // mj_setConst (:577-587); a type change alone leaves the masses, hence the constants, as they are
			if (set_mass) {
				std::string e2;
				applyMassChange(lo, hi, e2);
				if (!e2.empty()) note += e2 + '\n';
			}
		}
	}
	notifyGeomChanged(geom_id);
	resp.success = true;
	resp.status_message = note;
	return resp;
}

MujocoEnv::GetGeomPropertiesResponse MujocoEnv::getGeomPropertiesCB(const std::string &geom_name, const std::string &admin_hash, int env)
{
	GetGeomPropertiesResponse resp;
	if (!authorized(admin_hash)) {
		resp.status_message = "Hash mismatch, no permission to get geom properties!";
		resp.success = false;
		return resp;
	}
	if (!model_valid_ || env < 0 || env >= nenv_) {
		resp.success = false;
		resp.status_message = "No model loaded or env out of range";
		return resp;
	}
	const int geom_id = mj_name2id(&model_, mjOBJ_GEOM, geom_name.c_str());
	if (geom_id == -1) {
		resp.status_message = "Could not find model (mujoco geom) with name " + geom_name;
		resp.success = false;
		return resp;
	}
	const int body_id = model_.geom_bodyid[geom_id];
	std::lock_guard<MujocoEnvMutex> lk_sim(physics_thread_mutex_);
	ensureMirrors();
	const size_t g = (size_t)env * model_.ngeom + geom_id;
	resp.properties.name = geom_name;
	resp.properties.body_mass = env_body_mass_[(size_t)env * model_.nbody + body_id];
	for (int k = 0; k < 3; k++) {
		resp.properties.friction[k] = env_geom_friction_[3 * g + k];
		resp.properties.size[k] = env_geom_size_[3 * g + k];
	}
	resp.properties.type = env_geom_type_[g];
	return resp;
}

// ------------------------------------------------------------------------------------ equality constraints (:641-897)
bool MujocoEnv::setEqualityConstraintParameters(const EqualityConstraintParameters &parameters, int lo, int hi, std::string &note)
{
	const int eq_id = mj_name2id(&model_, mjOBJ_EQUALITY, parameters.name.c_str());
	if (eq_id == -1) return false;  // "Could not find specified equality constraint"
	const mjb_model_desc &d = current_.desc;
	// The reference re-points eq_obj1id / eq_obj2id when element names resolve (:651-725).  The engine's per-env overrides carry the
	// PARAMETERS of an equality, not its topology: a request naming other elements than the model's is reported, not applied.
	const int objtype = parameters.type == mjEQ_TENDON ? mjOBJ_TENDON : (parameters.type == mjEQ_JOINT ? mjOBJ_JOINT : mjOBJ_XBODY);
	if (!parameters.element1.empty()) {
		const int id1 = mj_name2id(&model_, objtype, parameters.element1.c_str());
		if (id1 != -1 && id1 != d.eq_obj1id[eq_id]) note += "equality '" + parameters.name + "': element1 differs from the model's; the constrained elements cannot be changed per env\n";
	}
	if (!parameters.element2.empty()) {
		const int id2 = mj_name2id(&model_, objtype, parameters.element2.c_str());
		if (id2 != -1 && id2 != d.eq_obj2id[eq_id]) note += "equality '" + parameters.name + "': element2 differs from the model's; the constrained elements cannot be changed per env\n";
	}
	for (int e = lo; e < hi; e++) {
		double *o = env_equality_.data() + ((size_t)e * d.neq + eq_id) * kEqStride;
		double *data = o + 1;
		switch (parameters.type) {
		case mjEQ_TENDON:
		case mjEQ_JOINT:
			for (int k = 0; k < 5; k++) data[k] = parameters.polycoef[k];
			break;
		case mjEQ_WELD:
			for (int k = 0; k < 3; k++) data[k] = parameters.anchor[k];
			for (int k = 0; k < 3; k++) data[3 + k] = parameters.relpose[k];
			for (int k = 0; k < 4; k++) data[6 + k] = parameters.relpose[3 + k];  // orientation w, x, y, z
			data[10] = parameters.torquescale;
			break;
		case mjEQ_CONNECT:
			for (int k = 0; k < 3; k++) data[k] = parameters.anchor[k];
			break;
		default: break;
		}
		o[0] = parameters.active ? 1.0 : 0.0;
		o[14] = parameters.solverParameters.dmin;
		o[15] = parameters.solverParameters.dmax;
		o[16] = parameters.solverParameters.width;
		o[17] = parameters.solverParameters.midpoint;
		o[18] = parameters.solverParameters.power;
		o[12] = parameters.solverParameters.timeconst;
		o[13] = parameters.solverParameters.dampratio;
	}
	return true;
}

MujocoEnv::ServiceResponse MujocoEnv::setEqualityConstraintParametersArrayCB(const std::vector<EqualityConstraintParameters> &parameters,
                                                                            const std::string &admin_hash, int env_lo, int env_hi)
{
	ServiceResponse resp;
	if (!authorized(admin_hash)) {
		resp.status_message = "Hash mismatch, no permission to get geom properties!";  // (sic: the reference's text, :758)
		resp.success = false;
		return resp;
	}
	std::lock_guard<MujocoEnvMutex> lk_sim(physics_thread_mutex_);
	if (!model_valid_) {
		resp.success = false;
		resp.status_message = "No model loaded";
		return resp;
	}
	resp.success = true;
	const int lo = std::max(0, env_lo), hi = env_hi < 0 ? nenv_ : std::min(env_hi, nenv_);
	ensureMirrors();
	bool failed_any = false, succeeded_any = false;
	std::string note;
	for (const auto &p : parameters) {
		const bool ok = current_.desc.neq > 0 && lo < hi && setEqualityConstraintParameters(p, lo, hi, note);
		failed_any = failed_any || !ok;
		succeeded_any = succeeded_any || ok;
	}
	if (succeeded_any) {
		std::string err;
		if (pushEnvParam(MJR_ENV_EQUALITY, lo, hi, env_equality_.data() + (size_t)lo * current_.desc.neq * kEqStride, err) != 0) note += err + '\n';
	}
	if (succeeded_any && failed_any) {
		resp.status_message = "Not all constraints could be set";
		resp.success = false;
	} else if (failed_any) {
		resp.status_message = "Could not set any constraints";
		resp.success = false;
	} else {
		resp.status_message = note;
	}
	return resp;
}

bool MujocoEnv::getEqualityConstraintParameters(EqualityConstraintParameters &parameters, int env)
{
	const int eq_id = mj_name2id(&model_, mjOBJ_EQUALITY, parameters.name.c_str());
	if (eq_id == -1) return false;
	const mjb_model_desc &d = current_.desc;
	const double *o = env_equality_.data() + ((size_t)env * d.neq + eq_id) * kEqStride, *data = o + 1;
	parameters.type = d.eq_type[eq_id];
	auto name_of = [&](const std::vector<std::string> &tab, int id) { return id >= 0 && id < (int)tab.size() ? tab[id] : std::string(); };
	const int id1 = d.eq_obj1id[eq_id], id2 = d.eq_obj2id[eq_id];
	switch (parameters.type) {
	case mjEQ_CONNECT:  // (the reference looks element1 up among the JOINT names here, :794 -- a slip; connect constrains bodies)
		parameters.element1 = name_of(model_.body_names, id1);
		parameters.element2 = name_of(model_.body_names, id2);
		for (int k = 0; k < 3; k++) parameters.anchor[k] = data[k];
		break;
	case mjEQ_WELD:
		parameters.element1 = name_of(model_.body_names, id1);
		parameters.element2 = name_of(model_.body_names, id2);
		for (int k = 0; k < 3; k++) parameters.anchor[k] = data[k];
		for (int k = 0; k < 3; k++) parameters.relpose[k] = data[3 + k];
		for (int k = 0; k < 4; k++) parameters.relpose[3 + k] = data[6 + k];
		parameters.torquescale = data[10];
		break;
	case mjEQ_JOINT:
		parameters.element1 = name_of(model_.joint_names, id1);
		parameters.element2 = name_of(model_.joint_names, id2);
		for (int k = 0; k < 5; k++) parameters.polycoef[k] = data[k];
		break;
	case mjEQ_TENDON:
		parameters.element1 = name_of(model_.tendon_names, id1);
		parameters.element2 = name_of(model_.tendon_names, id2);
		for (int k = 0; k < 5; k++) parameters.polycoef[k] = data[k];
		break;
	default: break;
	}
	parameters.active = o[0] != 0;
	parameters.solverParameters.dmin = o[14];
	parameters.solverParameters.dmax = o[15];
	parameters.solverParameters.width = o[16];
	parameters.solverParameters.midpoint = o[17];
	parameters.solverParameters.power = o[18];
	parameters.solverParameters.timeconst = o[12];
	parameters.solverParameters.dampratio = o[13];
	return true;
}

MujocoEnv::GetEqualityResponse MujocoEnv::getEqualityConstraintParametersArrayCB(const std::vector<std::string> &names,
                                                                                const std::string &admin_hash, int env)
{
	GetEqualityResponse resp;
	if (!authorized(admin_hash)) {
		resp.status_message = "Hash mismatch, no permission to get geom properties!";  // (sic, :871)
		resp.success = false;
		return resp;
	}
	std::lock_guard<MujocoEnvMutex> lk_sim(physics_thread_mutex_);
	if (!model_valid_ || env < 0 || env >= nenv_) {
		resp.success = false;
		resp.status_message = "No model loaded or env out of range";
		return resp;
	}
	resp.success = true;
	ensureMirrors();
	bool failed_any = false, succeeded_any = false;
	for (const auto &name : names) {
		EqualityConstraintParameters eqc;
		eqc.name = name;
		const bool ok = current_.desc.neq > 0 && getEqualityConstraintParameters(eqc, env);
		failed_any = failed_any || !ok;
		succeeded_any = succeeded_any || ok;
		if (ok) resp.parameters.emplace_back(eqc);
	}
	if (succeeded_any && failed_any) {
		resp.status_message = "Not all constraints could be fetched";
		resp.success = false;
	} else if (failed_any) {
		resp.status_message = "Could not fetch any constraints";
		resp.success = false;
	}
	return resp;
}

}  // namespace mujoco_ros
