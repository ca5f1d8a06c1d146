"""Host side of the lane = env kernel (csrc/mjb_lane_env.hip), no GPU: which models it takes, and that the generated topology header is the
one tools/gen_lane_env_topo.py writes from the assets today."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def classify(model):
    from mujoco_ros_pkgs_amd import engine
    cm = engine.CompiledModel(model)
    v = int(cm.lib.mjb_model_lane_env(cm.ptr))
    cm.close()
    return v


def test_which_models_the_kernel_takes():
    from mujoco_ros_pkgs_amd import mjcf
    from test_gpu_lane_env import JIT_ARM, two_arm_xml
    assert classify(mjcf.load_asset("franka_like")) == 0            # compiled in (csrc/lane_env_topos.h)
    assert classify(mjcf.load_asset("lane_env_tree")) == 1
    assert classify(mjcf.compile_xml_string(two_arm_xml())) == -2   # eligible: built by hiprtc at the first eligible launch
    arm = JIT_ARM.replace('actuator="3"', 'actuator="act3"').replace('<motor joint="j4" forcelimited', '<motor name="act3" joint="j4" forcelimited')
    assert classify(mjcf.compile_xml_string(arm)) == -2
    assert classify(mjcf.load_asset("franka_table")) == -1          # constraint rows, a free joint
    assert classify(mjcf.load_asset("shadow_hand_grasp")) == -1
    for world in ("pendulum_world", "sensors_world", "mocap_world", "equality_world"):
        assert classify(mjcf.compile_xml_file(os.path.join(ROOT, "tests", "golden", world + ".xml"))) == -1, world
    # RK4, a gyro sensor and a second joint on a body each take a model out
    base = two_arm_xml(3)
    assert classify(mjcf.compile_xml_string(base)) == -2
    assert classify(mjcf.compile_xml_string(base.replace('integrator="Euler"', 'integrator="RK4"'))) == -1


def test_which_models_the_split_step_takes():
    """mjb_model_split_step: the smooth kernel's compiled-in topologies (csrc/smooth_topos.h) whose constraint half is kernel variant 9's -- plain PGS,
    pyramidal cones, nv <= 16."""
    from mujoco_ros_pkgs_amd import engine, mjcf

    def split(model):
        cm = engine.CompiledModel(model)
        v = int(cm.lib.mjb_model_split_step(cm.ptr))
        cm.close()
        return v
    assert split(mjcf.load_asset("franka_table")) == 0
    assert split(mjcf.load_asset("split_step_tree")) == 1
    assert split(mjcf.load_asset("franka_like")) == -1         # no constraint rows: the lane = env kernel proper takes it
    assert split(mjcf.load_asset("shadow_hand_grasp")) == -1   # Newton, 30 dofs
    xml = open(os.path.join(mjcf.ASSET_DIR, "franka_table.xml")).read()
    assert split(mjcf.compile_xml_string(xml.replace('solver="PGS"', 'solver="Newton"'))) == -1
    assert split(mjcf.compile_xml_string(xml.replace('cone="pyramidal"', 'cone="elliptic"'))) == -1


def test_generated_topology_header_is_current(tmp_path):
    """csrc/lane_env_topos.h is generated from the asset XMLs; a changed asset with a stale header would silently stop matching (the
    model would still run, on hiprtc's build or the generic kernels)."""
    hdrs = [os.path.join(ROOT, "mujoco_ros_pkgs_amd", "csrc", n) for n in ("lane_env_topos.h", "smooth_topos.h")]
    before = [open(h).read() for h in hdrs]
    try:
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_lane_env_topo.py")], stdout=subprocess.DEVNULL)
        after = [open(h).read() for h in hdrs]
    finally:
        for h, t in zip(hdrs, before):
            open(h, "w").write(t)
    assert after == before, "run tools/gen_lane_env_topo.py and rebuild"
