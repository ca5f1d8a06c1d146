// common_types.h — types shared by the ROS-free host mirror of the reference's core runtime.
//
// Mirrors /root/reference mujoco_ros/include/mujoco_ros/common_types.h:74-88 (mjModelPtr / mjDataPtr /
// MujocoPluginPtr / MujocoEnvPtr) for a BATCH of env instances.  `mjModel` / `mjData` here are the
// per-env VIEW structs plugins receive: same member names as MuJoCo 2.3.7's structs for every field the
// reference's own code touches outside rendering; pointers reference a host
// mirror of one env's frame and are valid for the duration of a callback (or until the next reload for
// the env-0 view handed to MujocoPlugin::load, plugin_utils.h:146).
#pragma once

#include <memory>
#include <string>
#include <vector>

#include "../../include/mjb.h"

typedef double mjtNum;

namespace mujoco_ros {

struct mjOption {
	mjtNum timestep;
	mjtNum gravity[3];
	mjtNum tolerance;
	mjtNum impratio;
	int integrator, cone, solver, iterations, disableflags;
};

// constant model view (shared by all envs)
struct mjModel {
	int nq, nv, nu, na, nbody, njnt, ngeom, nsite, nsensor, nsensordata;
	mjOption opt;
	const mjb_model_desc *desc;  // all arrays, by their mjModel names (desc->jnt_type, desc->body_mass, ...)
	// the arrays the reference's plugins / services read, by their mjModel names
	const int *jnt_type, *jnt_qposadr, *jnt_dofadr, *jnt_bodyid, *body_jntadr, *body_jntnum, *geom_bodyid, *geom_type,
	    *site_bodyid, *sensor_type, *sensor_adr, *sensor_dim, *sensor_objid, *sensor_objtype, *sensor_refid,
	    *sensor_reftype;
	const mjtNum *qpos0, *body_mass, *geom_size, *geom_friction, *sensor_cutoff;
	// name tables (mj_name2id / mj_id2name)
	std::vector<std::string> joint_names, body_names, geom_names, site_names, sensor_names, actuator_names, equality_names, tendon_names;
};

// per-env data view
struct mjData {
	int env_id;     // index of this env instance in the batch
	mjtNum time;
	mjtNum *qpos, *qvel, *ctrl, *qacc, *qacc_warmstart;
	mjtNum *act;    // actuator activations [na] (read-only for plugins: the engine integrates them)
	mjtNum *qfrc_applied, *xfrc_applied, *qfrc_passive;  // callback-writable force fields (plugin_utils.h:91,101)
	mjtNum *sensordata;
	mjtNum *mocap_pos, *mocap_quat;  // written by the mocap plugin (mocap_plugin.cpp:102-103)
	mjtNum *xpos, *xquat, *xmat, *xipos, *ximat, *cvel, *subtree_com, *site_xpos, *site_xmat, *geom_xpos, *geom_xmat;
	mjtNum *actuator_force, *qfrc_bias, *qfrc_actuator;
};

// joint / object name lookup (mj_name2id restated over the name tables); -1 if absent
enum mjtObjKind { mjOBJ_BODY = 1, mjOBJ_XBODY = 2, mjOBJ_JOINT = 3, mjOBJ_GEOM = 5, mjOBJ_SITE = 6, mjOBJ_TENDON = 17, mjOBJ_ACTUATOR = 18, mjOBJ_SENSOR = 19,
	              mjOBJ_EQUALITY = 16 };
int mj_name2id(const mjModel *m, int type, const char *name);

class MujocoEnv;
class MujocoPlugin;
typedef MujocoEnv *MujocoEnvPtr;  // non-owning, as handed to MujocoPlugin::init
typedef std::unique_ptr<MujocoPlugin> MujocoPluginPtr;

// opaque stand-in for mjvScene (rendering itself is out of scope; the hook is kept)
struct mjvScene {
	int ngeom = 0;
};

}  // namespace mujoco_ros
