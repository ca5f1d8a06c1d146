// test_plugin.h — flag-setting dummy plugin, the build's counterpart of the reference's
// mujoco_ros/test/test_plugin/test_plugin.{h,cpp}: every callback raises an atomic flag so tests can pin
// which callbacks fire (mujoco_ros_plugin_test.cpp:97-128) and what a failed load() implies (:182-319).
#pragma once

#include <atomic>

#include "plugin_utils.h"

namespace mujoco_ros {

class TestPlugin : public MujocoPlugin {
public:
	~TestPlugin() override = default;
	bool load(const mjModel *m, mjData *d) override;
	void reset() override;
	void controlCallback(const mjModel *model, mjData *data) override;
	void passiveCallback(const mjModel *model, mjData *data) override;
	void renderCallback(const mjModel *model, mjData *data, mjvScene *scene) override;
	void lastStageCallback(const mjModel *model, mjData *data) override;
	void onGeomChanged(const mjModel *model, mjData *data, const int geom_id) override;
	// config "callbacks": "control" declares the control / passive callbacks only (what a ros_control-style plugin implements,
	// mujoco_ros_control_plugin.cpp: controlCallback alone) -- the host runtime then chains consecutive split steps
	unsigned callbackMask() const override { return mask_; }

	std::atomic_bool ran_reset = { false }, ran_control_cb = { false }, ran_passive_cb = { false },
	                 ran_render_cb = { false }, ran_last_cb = { false }, ran_on_geom_changed_cb = { false };
	std::atomic_bool got_config_param = { false }, got_lvl1_nested_array = { false }, got_lvl2_nested_array = { false },
	                 got_lvl1_nested_struct = { false }, got_lvl2_nested_struct = { false }, should_fail = { false };
	std::atomic_int control_calls = { 0 }, passive_calls = { 0 }, last_calls = { 0 }, last_env = { -1 };
	// optional behaviour for data-path tests: ctrl[i] += ctrl_bias, qfrc_passive[i] += passive_bias
	double ctrl_bias = 0, passive_bias = 0;
	// ... and from sim time xfrc_time on, xfrc_applied[6 * xfrc_body + 2] = xfrc_z (a plugin that starts pushing a body in mid-run)
	double xfrc_time = -1, xfrc_z = 0;
	int xfrc_body = 1;

private:
	unsigned mask_ = CB_ALL;
	const mjModel *m_ = nullptr;
	mjData *d_ = nullptr;
};

}  // namespace mujoco_ros
