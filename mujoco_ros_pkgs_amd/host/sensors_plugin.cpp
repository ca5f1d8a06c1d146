// sensors_plugin.cpp -- see sensors_plugin.h.  Semantics follow the reference line by line where it has any:
//   value (no noise model)    = sensordata / cutoff'            (cutoff' = cutoff > 0 ? cutoff : 1)
//   value (noise model)       = sensordata + noise_k / cutoff'  (NOT divided: kept as the reference computes it)
//   quaternions               = normalize(setRPY(noise) * normalize(float32 reading))
//   ground truth              = sensordata / cutoff', absent in eval mode
#include "sensors_plugin.h"

#include <cmath>

#include "mujoco_env.h"

namespace mujoco_ros {
namespace sensors {

namespace {
const std::string &nameOf(const std::vector<std::string> &names, int id)
{
	static const std::string empty;
	return (id >= 0 && id < (int)names.size()) ? names[id] : empty;
}
}  // namespace

bool MujocoRosSensorsPlugin::load(const mjModel *m, mjData * /*d*/)
{
	eval_mode_ = env_ptr_ ? env_ptr_->settings_.eval_mode : false;
	if (rosparam_config_.hasMember("seed")) rand_generator_.seed((unsigned int)rosparam_config_["seed"].asInt());
	else rand_generator_.seed(std::random_device{}());  // as the reference (mujoco_sensor_handler_plugin.h:94)
	initSensors(m);
	model_ = m;
	const size_t n = env_ptr_ ? (size_t)env_ptr_->nenv() : 1;
	records_.assign(n, std::vector<SensorRecord>());
	pending_.assign(n, Pending());
	sensor_cfg_.assign((size_t)m->nsensor, nullptr);
	for (int k = 0; k < m->nsensor; k++) {
		auto it = sensor_map_.find(nameOf(m->sensor_names, k));
		if (it != sensor_map_.end() && !it->first.empty()) sensor_cfg_[k] = &it->second;
	}
	return true;
}

void MujocoRosSensorsPlugin::initSensors(const mjModel *model)
{
	sensor_map_.clear();
	for (int n = 0; n < model->nsensor; n++) {
		const std::string &sensor_name = nameOf(model->sensor_names, n);
		if (sensor_name.empty()) continue;  // "Sensor name resolution error. Skipping" (:452-456)
		const int type = model->sensor_type[n], objid = model->sensor_objid[n];
		std::string frame_id = "world";
		SensorConfig cfg;
		bool global_frame = false, done = false;
		auto ref_frame = [&]() {  // relative frame of frame* sensors (:468-478): a site reference maps to its body
			int refid = model->sensor_refid[n];
			if (refid != -1) {
				int reftype = model->sensor_reftype[n];
				if (reftype == MJB_OBJ_SITE) refid = model->site_bodyid[refid];
				frame_id = nameOf(model->body_names, refid);
			}
		};
		switch (type) {
		case MJB_SENS_FRAMEXAXIS: case MJB_SENS_FRAMEYAXIS: case MJB_SENS_FRAMEZAXIS: case MJB_SENS_FRAMELINVEL:
		case MJB_SENS_FRAMELINACC: case MJB_SENS_FRAMEANGACC:
			ref_frame();
			cfg.kind = VECTOR3_STAMPED;
			done = true;
			break;
		case MJB_SENS_SUBTREECOM: case MJB_SENS_SUBTREELINVEL: case MJB_SENS_SUBTREEANGMOM:  // (:490-501)
			cfg.kind = VECTOR3_STAMPED;
			done = global_frame = true;
			break;
		case MJB_SENS_FRAMEPOS:
			ref_frame();
			cfg.kind = POINT_STAMPED;
			done = global_frame = true;
			break;
		case MJB_SENS_BALLQUAT: case MJB_SENS_FRAMEQUAT:
			cfg.kind = QUATERNION_STAMPED;
			done = global_frame = true;
			break;
		default: break;
		}
		if (done) {
			cfg.frame_id = frame_id;
			sensor_map_[sensor_name] = cfg;
		}
		if (global_frame || frame_id != "world") continue;  // (:540-545)
		// site-attached sensors report in the frame of the site's parent body (:547)
		const int parent_id = (objid >= 0 && objid < model->nsite) ? model->site_bodyid[objid] : 0;
		frame_id = nameOf(model->body_names, parent_id);
		switch (type) {
		case MJB_SENS_ACCELEROMETER: case MJB_SENS_VELOCIMETER: case MJB_SENS_GYRO: case MJB_SENS_FORCE: case MJB_SENS_TORQUE:
		case MJB_SENS_MAGNETOMETER: case MJB_SENS_BALLANGVEL:
			cfg.kind = VECTOR3_STAMPED;
			cfg.frame_id = frame_id;
			sensor_map_[sensor_name] = cfg;
			break;
		case MJB_SENS_TOUCH: case MJB_SENS_RANGEFINDER: case MJB_SENS_JOINTPOS: case MJB_SENS_JOINTVEL: case MJB_SENS_TENDONPOS: case MJB_SENS_TENDONVEL:
		case MJB_SENS_ACTUATORPOS: case MJB_SENS_ACTUATORVEL: case MJB_SENS_ACTUATORFRC: case MJB_SENS_JOINTACTFRC: case MJB_SENS_JOINTLIMITPOS:
		case MJB_SENS_JOINTLIMITVEL: case MJB_SENS_JOINTLIMITFRC: case MJB_SENS_TENDONLIMITPOS: case MJB_SENS_TENDONLIMITVEL: case MJB_SENS_TENDONLIMITFRC:  // (:575-590)
			cfg.kind = SCALAR_STAMPED;
			cfg.frame_id = frame_id;
			sensor_map_[sensor_name] = cfg;
			break;
		default: break;  // "is unknown! Cannot publish" (:601-604), e.g. frameangvel / clock
		}
	}
}

bool MujocoRosSensorsPlugin::registerNoiseModel(const std::string &sensor_name, unsigned char set_flag, const double *mean,
                                                const double *std, const std::string &admin_hash)
{
	if (env_ptr_ && env_ptr_->settings_.eval_mode && admin_hash != env_ptr_->settings_.admin_hash) return false;  // (:126-135)
	auto pos = sensor_map_.find(sensor_name);
	if (pos == sensor_map_.end()) return true;  // "No sensor with name ... Can not apply noise model", still success (:143-148)
	SensorConfig &config = pos->second;
	int noise_idx = 0;
	for (int k = 0; k < 3; k++)
		if (set_flag & (1 << k)) {  // (:153-167): the n-th set bit reads entry n
			config.mean[noise_idx] = mean[noise_idx];
			config.sigma[noise_idx] = std[noise_idx];
			noise_idx++;
		}
	config.is_set = config.is_set | set_flag;
	return true;
}

const std::vector<SensorRecord> &MujocoRosSensorsPlugin::records(int env) const
{
	static const std::vector<SensorRecord> none;
	if (env < 0 || env >= (int)records_.size()) return none;
	if (env_ptr_) {  // the mirrors are written by the physics thread between steps
		std::lock_guard<MujocoEnvMutex> lock(env_ptr_->physics_thread_mutex_);
		if (pending_[env].stale) build(env);
	} else if (pending_[env].stale) build(env);
	return records_[env];
}

// lastStageCallback (:175-437): per env and step, O(1) -- the messages are produced by build() when somebody reads them
void MujocoRosSensorsPlugin::lastStageCallback(const mjModel * /*model*/, mjData *data)
{
	const int env = data->env_id;
	if (env < 0) return;
	if (env >= (int)pending_.size()) {
		pending_.resize((size_t)env + 1);
		records_.resize((size_t)env + 1);
	}
	Pending &p = pending_[env];
	p.sensordata = data->sensordata;
	p.stamp = data->time;
	p.stale = true;
}

void MujocoRosSensorsPlugin::build(int env) const
{
	const mjModel *model = model_;
	Pending &p = pending_[env];
	p.stale = false;
	std::vector<SensorRecord> &out = records_[env];
	out.clear();
	if (!model || !p.sensordata) return;
	for (int n = 0; n < model->nsensor; n++) {
		const SensorConfig *cfgp = n < (int)sensor_cfg_.size() ? sensor_cfg_[n] : nullptr;
		if (!cfgp) continue;
		const SensorConfig &config = *cfgp;
		const std::string &sensor_name = nameOf(model->sensor_names, n);
		const int adr = model->sensor_adr[n];
		const double cutoff = model->sensor_cutoff[n] > 0 ? model->sensor_cutoff[n] : 1;
		const double *sd = p.sensordata + adr;
		SensorRecord rec;
		rec.name = sensor_name;
		rec.frame_id = config.frame_id;
		rec.kind = config.kind;
		rec.env = env;
		rec.stamp = p.stamp;
		rec.has_truth = !eval_mode_;
		const int dim = config.kind == SCALAR_STAMPED ? 1 : (config.kind == QUATERNION_STAMPED ? 4 : 3);
		for (int k = 0; k < dim; k++) rec.truth[k] = static_cast<float>(sd[k] / cutoff);
		if (config.is_set == 0) {
			for (int k = 0; k < dim; k++) rec.value[k] = rec.truth[k];
		} else if (config.kind == SCALAR_STAMPED) {
			const double noise = noise_dist_(rand_generator_) * config.sigma[0] + config.mean[0];
			rec.value[0] = static_cast<float>(sd[0] + noise / cutoff);
		} else if (config.kind == QUATERNION_STAMPED) {
			double q[4] = { rec.truth[0], rec.truth[1], rec.truth[2], rec.truth[3] }, rpy[3] = { 0, 0, 0 };
			const double nq = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
			for (int k = 0; k < 4; k++) q[k] /= nq;
			int noise_idx = 0;
			for (int k = 0; k < 3; k++)
				if (config.is_set & (1 << k)) {
					rpy[k] = noise_dist_(rand_generator_) * config.sigma[noise_idx] + config.mean[noise_idx];
					noise_idx++;
				}
			const double cr = std::cos(0.5 * rpy[0]), sr = std::sin(0.5 * rpy[0]), cp = std::cos(0.5 * rpy[1]), sp = std::sin(0.5 * rpy[1]);
			const double cy = std::cos(0.5 * rpy[2]), sy = std::sin(0.5 * rpy[2]);
			double r[4] = { cr * cp * cy + sr * sp * sy, sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy };
			const double nr = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
			for (int k = 0; k < 4; k++) r[k] /= nr;
			double o[4] = { r[0] * q[0] - r[1] * q[1] - r[2] * q[2] - r[3] * q[3], r[0] * q[1] + r[1] * q[0] + r[2] * q[3] - r[3] * q[2],
				            r[0] * q[2] - r[1] * q[3] + r[2] * q[0] + r[3] * q[1], r[0] * q[3] + r[1] * q[2] - r[2] * q[1] + r[3] * q[0] };
			const double no = std::sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
			for (int k = 0; k < 4; k++) rec.value[k] = static_cast<float>(o[k] / no);
		} else {
			int noise_idx = 0;
			for (int k = 0; k < 3; k++) {
				double noise = 0;
				if (config.is_set & (1 << k)) {
					noise = noise_dist_(rand_generator_) * config.sigma[noise_idx] + config.mean[noise_idx];
					noise_idx++;
				}
				rec.value[k] = static_cast<float>(sd[k] + noise / cutoff);
			}
		}
		out.push_back(rec);
	}
}

MUJOCO_REGISTER_PLUGIN("mujoco_ros_sensors/MujocoRosSensorsPlugin", MujocoRosSensorsPlugin);

}  // namespace sensors
}  // namespace mujoco_ros
