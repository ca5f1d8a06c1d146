// sensors_plugin.h -- ROS-free mirror of the reference's MujocoRosSensorsPlugin
// (/root/reference mujoco_ros_sensors/include/mujoco_ros_sensors/mujoco_sensor_handler_plugin.h:53-117,
//  src/mujoco_sensor_handler_plugin.cpp:52-640): per named sensor a typed record (scalar / vector3 / point /
// quaternion, frame id, value with the registered noise model, ground truth) is produced in lastStageCallback from
// mjData.sensordata exactly where the reference publishes its messages.  Publishers become the `records()` list.
#pragma once

#include <map>
#include <random>
#include <string>
#include <vector>

#include "plugin_utils.h"

namespace mujoco_ros {
namespace sensors {

enum MessageKind { SCALAR_STAMPED = 0, VECTOR3_STAMPED = 1, POINT_STAMPED = 2, QUATERNION_STAMPED = 3 };

struct SensorConfig {   // mujoco_sensor_handler_plugin.h:53-82
	std::string frame_id;
	int kind = SCALAR_STAMPED;
	unsigned char is_set = 0;  // bit k: noise on component k (set_flag of RegisterSensorNoiseModels)
	double mean[3] = { 0, 0, 0 }, sigma[3] = { 0, 0, 0 };
};

struct SensorRecord {   // what value_pub / gt_pub carry (float32 payloads, as the ROS messages)
	std::string name, frame_id;
	int kind = SCALAR_STAMPED, env = 0;
	double stamp = 0;            // sim time of the step (the reference stamps ros::Time::now() == sim time)
	float value[4] = { 0, 0, 0, 0 };
	float truth[4] = { 0, 0, 0, 0 };
	bool has_truth = true;       // false in eval mode: no ground-truth publisher (mujoco_sensor_handler_plugin.cpp:454-461 ...)
};

class MujocoRosSensorsPlugin : public MujocoPlugin {
public:
	~MujocoRosSensorsPlugin() override = default;
	bool load(const mjModel *m, mjData *d) override;
	void reset() override {}
	void lastStageCallback(const mjModel *model, mjData *data) override;
	// the plugin observes the end of a step and reads sensordata + time only (lastStageCallback, :175-437)
	unsigned callbackMask() const override { return CB_LASTSTAGE; }
	bool viewFields(std::vector<int> &fields) const override
	{
		fields = { MJB_F_sensordata, MJB_F_time };
		return true;
	}

	// registerNoiseModelsCB (:123-173): returns false ("success = false") on an admin-hash mismatch in eval mode
	bool registerNoiseModel(const std::string &sensor_name, unsigned char set_flag, const double *mean, const double *std,
	                        const std::string &admin_hash);
	const std::map<std::string, SensorConfig> &sensorMap() const { return sensor_map_; }
	// records of the last lastStageCallback of every env, in sensor order.  They are BUILT HERE, on demand: lastStageCallback only
	// notes that env's step (time stamp + where its sensordata mirror lives) -- building 4096 x nsensor typed records every step
	// for nobody was the whole cost of the plugin on a batch (1.9 M env-steps/s, profiles/r02_callback_path.txt); the noise of a
	// record is drawn when it is built, once per step and env at most.
	const std::vector<SensorRecord> &records(int env) const;

private:
	void initSensors(const mjModel *model);
	void build(int env) const;
	std::map<std::string, SensorConfig> sensor_map_;
	struct Pending {
		const double *sensordata = nullptr;  // the env's host mirror (valid until the next reload)
		double stamp = 0;
		bool stale = false;
	};
	mutable std::vector<Pending> pending_;
	mutable std::vector<std::vector<SensorRecord>> records_;
	const mjModel *model_ = nullptr;
	std::vector<const SensorConfig *> sensor_cfg_;  // per sensor id: its config or nullptr (resolved once: no map lookup per step)
	mutable std::mt19937 rand_generator_;
	mutable std::normal_distribution<double> noise_dist_{ 0.0, 1.0 };
	bool eval_mode_ = false;
};

}  // namespace sensors
}  // namespace mujoco_ros
