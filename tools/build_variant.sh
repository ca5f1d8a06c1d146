#!/bin/bash
# build_variant.sh <tag> <group> <extra -D flags...>: libmjb_x<tag>.so = libmjb.so with the step-kernel slice <group> (csrc/mjb_step.hip, MJB_GROUP)
# recompiled with extra defines -- for A/B measurements of compile-time knobs (MJB_LIBRARY=.../libmjb_x<tag>.so python bench.py ...).
set -e
tag=$1; grp=$2; shift 2
cd "$(dirname "$0")/../mujoco_ros_pkgs_amd/csrc"
HIPCC=/opt/rocm/bin/hipcc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -ffp-contract=fast -Wno-pass-failed -mllvm -disable-machine-licm"
$HIPCC $FLAGS "$@" -DMJB_GROUP=$grp -c -o mjb_x${tag}_g$grp.o mjb_step.hip 2>&1 | grep -v hip-link || true
objs=""
for g in 0 1 2 3 4 5 6; do if [ $g = $grp ]; then objs="$objs mjb_x${tag}_g$grp.o"; else objs="$objs mjb_step_g$g.o"; fi; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libmjb_x${tag}.so $objs mjb_api.o mjb_sensor_pack.o mjb_lane_env.o mjb_smooth.o 2>&1 | grep -v hip-link || true
ls -la libmjb_x${tag}.so
