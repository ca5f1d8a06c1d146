// plugin_utils.cpp — config values, parameter store and the plugin registry / loader
// (restates /root/reference mujoco_ros/src/plugin_utils.cpp:41-118 without pluginlib / XmlRpc / roscpp).
#include "plugin_utils.h"

#include <cctype>
#include <cstdlib>
#include <sstream>
#include <stdexcept>

namespace mujoco_ros {

static const ConfigValue kInvalid;

const ConfigValue &ConfigValue::operator[](const std::string &k) const
{
	if (type_ != TypeStruct) return kInvalid;
	auto it = struct_.find(k);
	return it == struct_.end() ? kInvalid : it->second;
}
const ConfigValue &ConfigValue::operator[](int i) const
{
	if (type_ != TypeArray || i < 0 || i >= (int)array_.size()) return kInvalid;
	return array_[i];
}
ConfigValue &ConfigValue::member(const std::string &k)
{
	if (type_ == TypeInvalid) type_ = TypeStruct;
	if (type_ != TypeStruct) throw std::runtime_error("ConfigValue::member on a non-struct");
	return struct_[k];
}
void ConfigValue::push_back(const ConfigValue &v)
{
	if (type_ == TypeInvalid) type_ = TypeArray;
	if (type_ != TypeArray) throw std::runtime_error("ConfigValue::push_back on a non-array");
	array_.push_back(v);
}
bool ConfigValue::asBool(bool dflt) const
{
	switch (type_) {
	case TypeBoolean: return b_;
	case TypeInt: return i_ != 0;
	case TypeString: return s_ == "true" || s_ == "True" || s_ == "1";
	default: return dflt;
	}
}
int ConfigValue::asInt(int dflt) const
{
	switch (type_) {
	case TypeInt: return i_;
	case TypeBoolean: return b_ ? 1 : 0;
	case TypeDouble: return (int)d_;
	case TypeString: return std::atoi(s_.c_str());
	default: return dflt;
	}
}
double ConfigValue::asDouble(double dflt) const
{
	switch (type_) {
	case TypeDouble: return d_;
	case TypeInt: return i_;
	case TypeString: return std::atof(s_.c_str());
	default: return dflt;
	}
}
std::string ConfigValue::asString(const std::string &dflt) const
{
	switch (type_) {
	case TypeString: return s_;
	case TypeInt: return std::to_string(i_);
	case TypeDouble: { std::ostringstream o; o << d_; return o.str(); }
	case TypeBoolean: return b_ ? "true" : "false";
	default: return dflt;
	}
}
std::string ConfigValue::toString() const
{
	std::ostringstream o;
	switch (type_) {
	case TypeArray:
		o << "[";
		for (size_t i = 0; i < array_.size(); i++) o << (i ? "," : "") << array_[i].toString();
		o << "]";
		break;
	case TypeStruct: {
		o << "{";
		bool first = true;
		for (const auto &kv : struct_) {
			o << (first ? "" : ",") << "\"" << kv.first << "\":" << kv.second.toString();
			first = false;
		}
		o << "}";
		break;
	}
	case TypeString: o << "\"" << s_ << "\""; break;
	case TypeInvalid: o << "null"; break;
	default: o << asString();
	}
	return o.str();
}

// ---- tiny recursive-descent JSON reader
namespace {
struct Reader {
	const std::string &t;
	size_t p = 0;
	explicit Reader(const std::string &s) : t(s) {}
	void ws() { while (p < t.size() && std::isspace((unsigned char)t[p])) p++; }
	[[noreturn]] void err(const char *m) { throw std::runtime_error(std::string("config JSON: ") + m + " at offset " + std::to_string(p)); }
	ConfigValue value()
	{
		ws();
		if (p >= t.size()) err("unexpected end");
		char c = t[p];
		if (c == '{') {
			p++;
			ConfigValue st = ConfigValue::emptyOf(ConfigValue::TypeStruct);
			while (true) {
				ws();
				if (p < t.size() && t[p] == '}') { p++; break; }
				if (p >= t.size() || t[p] != '"') err("expected key");
				std::string k = str();
				ws();
				if (p >= t.size() || t[p] != ':') err("expected ':'");
				p++;
				st.member(k) = value();
				ws();
				if (p < t.size() && t[p] == ',') { p++; continue; }
				if (p < t.size() && t[p] == '}') { p++; break; }
				err("expected ',' or '}'");
			}
			return st;
		}
		if (c == '[') {
			p++;
			ConfigValue a = ConfigValue::emptyOf(ConfigValue::TypeArray);
			while (true) {
				ws();
				if (p < t.size() && t[p] == ']') { p++; break; }
				a.push_back(value());
				ws();
				if (p < t.size() && t[p] == ',') { p++; continue; }
				if (p < t.size() && t[p] == ']') { p++; break; }
				err("expected ',' or ']'");
			}
			return a;
		}
		if (c == '"') return ConfigValue(str());
		if (!t.compare(p, 4, "true")) { p += 4; return ConfigValue(true); }
		if (!t.compare(p, 5, "false")) { p += 5; return ConfigValue(false); }
		if (!t.compare(p, 4, "null")) { p += 4; return ConfigValue(); }
		// number
		size_t s0 = p;
		bool isdbl = false;
		while (p < t.size() && (std::isdigit((unsigned char)t[p]) || t[p] == '-' || t[p] == '+' || t[p] == '.' || t[p] == 'e' || t[p] == 'E')) {
			if (t[p] == '.' || t[p] == 'e' || t[p] == 'E') isdbl = true;
			p++;
		}
		if (p == s0) err("unexpected character");
		std::string num = t.substr(s0, p - s0);
		if (isdbl) return ConfigValue(std::atof(num.c_str()));
		return ConfigValue(std::atoi(num.c_str()));
	}
	std::string str()
	{
		std::string o;
		p++;  // opening quote
		while (p < t.size() && t[p] != '"') {
			if (t[p] == '\\' && p + 1 < t.size()) {
				p++;
				char e = t[p];
				o += e == 'n' ? '\n' : (e == 't' ? '\t' : e);
			} else {
				o += t[p];
			}
			p++;
		}
		if (p >= t.size()) err("unterminated string");
		p++;
		return o;
	}
};
}  // namespace

ConfigValue ConfigValue::emptyOf(Type t)
{
	ConfigValue v;
	v.type_ = t;
	return v;
}

ConfigValue ConfigValue::fromJson(const std::string &text)
{
	Reader r(text);
	ConfigValue v = r.value();
	r.ws();
	if (r.p != text.size()) r.err("trailing characters");
	return v;
}

const ConfigValue &ParamServer::get(const std::string &key) const
{
	auto it = params_.find(key);
	return it == params_.end() ? kInvalid : it->second;
}
template <> void ParamServer::param<bool>(const std::string &key, bool &out, const bool &dflt) const
{
	out = has(key) ? get(key).asBool(dflt) : dflt;
}
template <> void ParamServer::param<int>(const std::string &key, int &out, const int &dflt) const
{
	out = has(key) ? get(key).asInt(dflt) : dflt;
}
template <> void ParamServer::param<double>(const std::string &key, double &out, const double &dflt) const
{
	out = has(key) ? get(key).asDouble(dflt) : dflt;
}
template <> void ParamServer::param<std::string>(const std::string &key, std::string &out, const std::string &dflt) const
{
	out = has(key) ? get(key).asString(dflt) : dflt;
}

int mj_name2id(const mjModel *m, int type, const char *name)
{
	const std::vector<std::string> *tab = nullptr;
	switch (type) {
	case mjOBJ_BODY: tab = &m->body_names; break;
	case mjOBJ_XBODY: tab = &m->body_names; break;
	case mjOBJ_EQUALITY: tab = &m->equality_names; break;
	case mjOBJ_TENDON: tab = &m->tendon_names; break;
	case mjOBJ_JOINT: tab = &m->joint_names; break;
	case mjOBJ_GEOM: tab = &m->geom_names; break;
	case mjOBJ_SITE: tab = &m->site_names; break;
	case mjOBJ_ACTUATOR: tab = &m->actuator_names; break;
	case mjOBJ_SENSOR: tab = &m->sensor_names; break;
	default: return -1;
	}
	// (unnamed elements are kept as "" in the tables: mj_name2id never matches them -- an empty name is "not found", as in MuJoCo,
	//  where setBodyStateCB / getGeomPropertiesCB then fail instead of acting on the first unnamed body or geom)
	if (!name || !*name) return -1;
	for (size_t i = 0; i < tab->size(); i++)
		if ((*tab)[i] == name) return (int)i;
	return -1;
}

namespace plugin_utils {

static std::map<std::string, PluginFactory> &registry()
{
	static std::map<std::string, PluginFactory> r;
	return r;
}

bool registerPluginType(const std::string &type, PluginFactory factory)
{
	registry()[type] = std::move(factory);
	return true;
}
bool isPluginTypeRegistered(const std::string &type) { return registry().count(type) > 0; }

bool parsePlugins(const ParamServer *params, ConfigValue &plugin_config)
{
	if (!params || !params->has(MUJOCO_PLUGIN_PARAM_NAME)) return false;
	plugin_config = params->get(MUJOCO_PLUGIN_PARAM_NAME);
	// the reference insists on an array of structs (plugin_utils.cpp:56-62)
	return plugin_config.getType() == ConfigValue::TypeArray;
}

void registerPlugins(const std::string &nh_namespace, const ConfigValue &config, std::vector<MujocoPluginPtr> &plugins,
                     MujocoEnv *env, const ParamServer *params, std::vector<std::string> *warnings)
{
	for (int i = 0; i < config.size(); i++) {
		if (config[i].getType() != ConfigValue::TypeStruct) {
			if (warnings) warnings->push_back("Error while parsing MujocoPlugins rosparam: wrong type; entry " + std::to_string(i) + " is not a struct");
			continue;
		}
		registerPlugin(nh_namespace, config[i], plugins, env, params, warnings);
	}
}

bool registerPlugin(const std::string &nh_namespace, const ConfigValue &config, std::vector<MujocoPluginPtr> &plugins,
                    MujocoEnv *env, const ParamServer *params, std::vector<std::string> *warnings)
{
	if (!config.hasMember("type")) {
		if (warnings) warnings->push_back("Error while parsing MujocoPlugins rosparam: Every listed plugin should provide a 'type'");
		return false;
	}
	const std::string type = config["type"].asString();
	auto it = registry().find(type);
	if (it == registry().end()) {
		if (warnings) warnings->push_back("The plugin failed to load (found no plugin of type " + type + ")");
		return false;
	}
	MujocoPluginPtr p(it->second());
	p->init(config, nh_namespace, env, params);
	plugins.emplace_back(std::move(p));
	return true;
}

}  // namespace plugin_utils
}  // namespace mujoco_ros
