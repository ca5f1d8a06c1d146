#!/bin/bash
# lane_env_probe_build.sh [extra -D flags]: csrc/libmjb_xprobe.so = libmjb.so with the lane = env object compiled with -DMJB_LE_PROBE (tools/lane_env_probe.py)
set -e
cd "$(dirname "$0")/../mujoco_ros_pkgs_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -ffp-contract=fast -Wno-pass-failed -mllvm -disable-machine-licm -DMJB_LE_PROBE "$@" -c -o /tmp/mjb_lane_env_probe.o mjb_lane_env.hip 2>&1 | grep -v hip-link || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libmjb_xprobe.so mjb_step_g0.o mjb_step_g1.o mjb_step_g2.o mjb_step_g3.o mjb_step_g4.o mjb_step_g5.o mjb_api.o mjb_sensor_pack.o /tmp/mjb_lane_env_probe.o 2>&1 | grep -v hip-link || true
