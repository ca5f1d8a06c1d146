#!/bin/bash
# ab_bench.sh <bench args> -- <lib tags...>: the same bench command on libmjb.so and on each libmjb_x<tag>.so (tools/build_variant.sh), value + kernel ms
args=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do args+=("$1"); shift; done; shift
for t in "" "$@"; do
  lib=$PWD/mujoco_ros_pkgs_amd/csrc/libmjb${t:+_x$t}.so
  MJB_LIBRARY=$lib python bench.py --no-cpu-baseline --no-other-configs "${args[@]}" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-10s %9.3f M env-steps/s  kernel %.2f ms' % ('${t:-base}', d['value']/1e6, d['roofline']['kernel_ms']))"
done
