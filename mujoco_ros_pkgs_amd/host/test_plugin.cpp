// test_plugin.cpp — see test_plugin.h (reference: mujoco_ros/test/test_plugin/test_plugin.cpp:44-108)
#include "test_plugin.h"

namespace mujoco_ros {

bool TestPlugin::load(const mjModel *m, mjData *d)
{
	if (rosparam_config_.hasMember("example_param")) got_config_param.store(true);
	if (rosparam_config_.hasMember("nested_array_param_1") &&
	    rosparam_config_["nested_array_param_1"].getType() == ConfigValue::TypeArray) {
		got_lvl1_nested_array.store(true);
		if (rosparam_config_["nested_array_param_1"][0].hasMember("nested_array_param_2")) got_lvl2_nested_array.store(true);
	}
	if (rosparam_config_.hasMember("nested_struct_param_1") &&
	    rosparam_config_["nested_struct_param_1"].getType() == ConfigValue::TypeStruct) {
		got_lvl1_nested_struct.store(true);
		if (rosparam_config_["nested_struct_param_1"].hasMember("nested_struct_param_2")) got_lvl2_nested_struct.store(true);
	}
	if (rosparam_config_.hasMember("callbacks") && rosparam_config_["callbacks"].asString() == "control") mask_ = CB_CONTROL | CB_PASSIVE;
	ctrl_bias = rosparam_config_["ctrl_bias"].asDouble(0.0);
	passive_bias = rosparam_config_["passive_bias"].asDouble(0.0);
	xfrc_time = rosparam_config_["xfrc_time"].asDouble(-1.0);
	xfrc_z = rosparam_config_["xfrc_z"].asDouble(0.0);
	xfrc_body = (int)rosparam_config_["xfrc_body"].asDouble(1.0);
	bool tmp_fail = false;
	if (node_handle_) node_handle_->param<bool>("should_fail", tmp_fail, false);
	should_fail.store(tmp_fail);
	if (tmp_fail) return false;
	m_ = m;
	d_ = d;
	return true;
}

void TestPlugin::reset() { ran_reset.store(true); }

void TestPlugin::controlCallback(const mjModel *model, mjData *data)
{
	ran_control_cb.store(true);
	control_calls++;
	last_env.store(data->env_id);
	if (ctrl_bias != 0)
		for (int i = 0; i < model->nu; i++) data->ctrl[i] += ctrl_bias;
	if (xfrc_time >= 0 && data->time >= xfrc_time - 1e-12 && xfrc_body > 0 && xfrc_body < model->nbody) data->xfrc_applied[6 * xfrc_body + 2] = xfrc_z;
}

void TestPlugin::passiveCallback(const mjModel *model, mjData *data)
{
	ran_passive_cb.store(true);
	passive_calls++;
	if (passive_bias != 0)
		for (int i = 0; i < model->nv; i++) data->qfrc_passive[i] += passive_bias;
}

void TestPlugin::renderCallback(const mjModel *, mjData *, mjvScene *) { ran_render_cb.store(true); }
void TestPlugin::lastStageCallback(const mjModel *, mjData *)
{
	ran_last_cb.store(true);
	last_calls++;
}
void TestPlugin::onGeomChanged(const mjModel *, mjData *, const int) { ran_on_geom_changed_cb.store(true); }

MUJOCO_REGISTER_PLUGIN("mujoco_ros/TestPlugin", TestPlugin);

}  // namespace mujoco_ros
