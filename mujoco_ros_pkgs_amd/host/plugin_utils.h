// plugin_utils.h — ROS-free restatement of the reference's plugin surface.
//
// Same class shape, virtuals, call order and safe_load / safe_reset semantics as
// /root/reference mujoco_ros/include/mujoco_ros/plugin_utils.h:45-161 and the loader in
// mujoco_ros/src/plugin_utils.cpp:41-118, with ROS types replaced:
//   XmlRpc::XmlRpcValue  -> ConfigValue  (nested bool/int/double/string/array/struct tree, parsed from JSON)
//   ros::NodeHandle      -> ParamServer* (string-keyed parameter store owned by the env) + namespace string
//   pluginlib class name -> static registry keyed by the same type strings ("mujoco_ros/TestPlugin")
#pragma once

#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "common_types.h"

namespace mujoco_ros {

// --------------------------------------------------------------------------------- config values
class ConfigValue {
public:
	enum Type { TypeInvalid, TypeBoolean, TypeInt, TypeDouble, TypeString, TypeArray, TypeStruct };
	ConfigValue() = default;
	explicit ConfigValue(bool b) : type_(TypeBoolean), b_(b) {}
	explicit ConfigValue(int i) : type_(TypeInt), i_(i) {}
	explicit ConfigValue(double d) : type_(TypeDouble), d_(d) {}
	explicit ConfigValue(const std::string &s) : type_(TypeString), s_(s) {}

	Type getType() const { return type_; }
	bool valid() const { return type_ != TypeInvalid; }
	bool hasMember(const std::string &k) const { return type_ == TypeStruct && struct_.count(k) > 0; }
	int size() const { return type_ == TypeArray ? (int)array_.size() : (type_ == TypeStruct ? (int)struct_.size() : 0); }
	const ConfigValue &operator[](const std::string &k) const;
	const ConfigValue &operator[](int i) const;
	ConfigValue &member(const std::string &k);  // creates (turns an invalid value into a struct)
	void push_back(const ConfigValue &v);       // turns an invalid value into an array
	bool asBool(bool dflt = false) const;
	int asInt(int dflt = 0) const;
	double asDouble(double dflt = 0) const;
	std::string asString(const std::string &dflt = std::string()) const;
	const std::map<std::string, ConfigValue> &members() const { return struct_; }
	std::string toString() const;

	// minimal JSON reader (objects, arrays, strings, numbers, true/false/null); throws std::runtime_error
	static ConfigValue fromJson(const std::string &text);
	static ConfigValue emptyOf(Type t);  // empty array / struct

private:
	Type type_ = TypeInvalid;
	bool b_ = false;
	int i_ = 0;
	double d_ = 0;
	std::string s_;
	std::vector<ConfigValue> array_;
	std::map<std::string, ConfigValue> struct_;
};

// string-keyed parameter store (what the reference reads through ros::NodeHandle::param)
class ParamServer {
public:
	void set(const std::string &key, const ConfigValue &v) { params_[key] = v; }
	void erase(const std::string &key) { params_.erase(key); }
	bool has(const std::string &key) const { return params_.count(key) > 0; }
	const ConfigValue &get(const std::string &key) const;
	template <typename T> void param(const std::string &key, T &out, const T &dflt) const;
	void clear() { params_.clear(); }

private:
	std::map<std::string, ConfigValue> params_;
};

// --------------------------------------------------------------------------------- plugin base
class MujocoPlugin {
public:
	virtual ~MujocoPlugin() = default;

	// Called directly after plugin creation (plugin_utils.h:51-57)
	void init(const ConfigValue &config, const std::string &nh_namespace, MujocoEnvPtr env_ptr, const ParamServer *params)
	{
		rosparam_config_ = config;
		nh_namespace_ = nh_namespace;
		env_ptr_ = env_ptr;
		node_handle_ = params;
		type_ = rosparam_config_["type"].asString();
	}

	std::string type_;

	// Wrapper that records whether loading succeeded (plugin_utils.h:69-78)
	bool safe_load(const mjModel *m, mjData *d)
	{
		loading_successful_ = load(m, d);
		if (!loading_successful_) last_warning_ = "Plugin of type '" + type_ + "' failed to load. It will be ignored until the next load attempt.";
		return loading_successful_;
	}

	// Only resets plugins whose load succeeded (plugin_utils.h:83-87)
	void safe_reset()
	{
		if (loading_successful_) reset();
	}

	bool loadingSuccessful() const { return loading_successful_; }
	const std::string &lastWarning() const { return last_warning_; }

	// To apply control, write into mjData.ctrl, mjData.qfrc_applied and/or mjData.xfrc_applied (plugin_utils.h:97)
	virtual void controlCallback(const mjModel * /*model*/, mjData * /*data*/) {}
	// Should ADD to mjData.qfrc_passive (plugin_utils.h:107)
	virtual void passiveCallback(const mjModel * /*model*/, mjData * /*data*/) {}
	// Visualisation hook (plugin_utils.h:116); rendering itself is out of scope, the hook is honoured
	virtual void renderCallback(const mjModel * /*model*/, mjData * /*data*/, mjvScene * /*scene*/) {}
	// End of a full env step, never inside integrator sub-steps (plugin_utils.h:126)
	virtual void lastStageCallback(const mjModel * /*model*/, mjData * /*data*/) {}
	// A geom was changed in the model (plugin_utils.h:135)
	virtual void onGeomChanged(const mjModel * /*model*/, mjData * /*data*/, const int /*geom_id*/) {}

	// ---- batched-runtime extension (no counterpart in the reference, whose mjData lives in host memory) ----
	// Which callbacks the plugin overrides, and which mjData fields it reads there.  The defaults -- everything -- keep the
	// reference's behaviour for a plugin that says nothing: every step is split at the control-callback point and the
	// whole T1 view is mirrored.  A plugin that only observes the end of a step (MujocoRosSensorsPlugin) lets the runtime
	// keep fused launches and move only the fields it names.
	enum : unsigned { CB_CONTROL = 1u, CB_PASSIVE = 2u, CB_RENDER = 4u, CB_LASTSTAGE = 8u, CB_ALL = 15u };
	virtual unsigned callbackMask() const { return CB_ALL; }
	// mjb_field ids the plugin reads through mjData in its callbacks; false = "any field of the view"
	virtual bool viewFields(std::vector<int> & /*fields*/) const { return false; }

protected:
	virtual bool load(const mjModel *m, mjData *d) = 0;  // plugin_utils.h:146
	virtual void reset() = 0;                            // plugin_utils.h:151

private:
	bool loading_successful_ = false;
	std::string last_warning_;

protected:
	MujocoPlugin() = default;
	ConfigValue rosparam_config_;
	const ParamServer *node_handle_ = nullptr;
	std::string nh_namespace_;
	MujocoEnvPtr env_ptr_ = nullptr;
};

namespace plugin_utils {

typedef std::function<MujocoPlugin *()> PluginFactory;

// static registry that takes the place of pluginlib::ClassLoader (plugin_utils.cpp:99,114-118)
bool registerPluginType(const std::string &type, PluginFactory factory);
bool isPluginTypeRegistered(const std::string &type);

// Reads the plugin list from the parameter store (plugin_utils.cpp:41-62): key MUJOCO_PLUGIN_PARAM_NAME must
// hold an ARRAY of structs.  Returns false when absent or not an array.
bool parsePlugins(const ParamServer *params, ConfigValue &plugin_config);

// registerPlugins / registerPlugin (plugin_utils.cpp:64-112): every entry must be a struct with a `type`
// member naming a registered class; offenders are skipped with a warning, the rest are appended.
void registerPlugins(const std::string &nh_namespace, const ConfigValue &config, std::vector<MujocoPluginPtr> &plugins,
                     MujocoEnv *env, const ParamServer *params, std::vector<std::string> *warnings = nullptr);
bool registerPlugin(const std::string &nh_namespace, const ConfigValue &config, std::vector<MujocoPluginPtr> &plugins,
                    MujocoEnv *env, const ParamServer *params, std::vector<std::string> *warnings = nullptr);

const static std::string MUJOCO_PLUGIN_PARAM_NAME = "MujocoPlugins";

}  // namespace plugin_utils

#define MUJOCO_REGISTER_PLUGIN(TYPE_STRING, CLASS)                                                         \
	static const bool CLASS##_registered_ =                                                                \
	    ::mujoco_ros::plugin_utils::registerPluginType(TYPE_STRING, []() -> ::mujoco_ros::MujocoPlugin * { return new CLASS(); })

}  // namespace mujoco_ros
