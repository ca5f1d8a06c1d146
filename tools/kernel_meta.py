#!/usr/bin/env python3
"""Register / scratch summary of every step-kernel variant from the ISA the lint reads (csrc/mjb_step_g*.s)."""
import glob
import os
import re
import sys

root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mujoco_ros_pkgs_amd", "csrc")
rows = []
for f in sorted(glob.glob(os.path.join(root, "mjb_step_g*.s"))):
    txt = open(f).read()
    for blk in re.split(r"\n  - \.agpr_count", txt)[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name or "mjb_step_kernel" not in name.group(1):
            continue
        mm = re.search(r"mjb_step_kernelILi(\d+)ELi(\d+)ELi(\d+)E", name.group(1))
        get = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))  # noqa: E731
        rows.append((int(mm.group(2)), int(mm.group(1)), int(mm.group(3)), get("vgpr_count"), get("vgpr_spill_count"), get("sgpr_spill_count"),
                     get("private_segment_fixed_size"), os.path.basename(f)))
print(f"{'kernel <G,CON,DENSE>':24s} {'vgpr':>5s} {'vgpr spills':>11s} {'sgpr spills':>11s} {'private B':>9s}  slice")
for con, g, dense, v, vs, ss, p, f in sorted(set(rows)):
    print(f"<{g},{con},{dense}>".ljust(24) + f" {v:5d} {vs:11d} {ss:11d} {p:9d}  {f}")
# the lane = env kernels (csrc/mjb_lane_env.s): one per topology and LDS budget
f = os.path.join(root, "mjb_lane_env.s")
if os.path.exists(f):
    print(f"{'lane = env <topology, LDS KB>':44s} {'vgpr':>5s} {'agpr':>5s} {'vgpr spills':>11s} {'sgpr spills':>11s} {'private B':>9s}")
    txt = open(f).read()
    for blk in re.split(r"\n  - \.agpr_count", txt)[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        mm = name and re.search(r"mjb_lane_env_(duo2_|duo_|trio_|)kernelI\d+LeTopo_(\w+?)(?:Li(\d+))?E", name.group(1))
        if not mm:
            continue
        form = {"": "", "duo_": ", two halves", "duo2_": ", pipelined", "trio_": ", three wavefronts"}[mm.group(1)]
        mm = (None, mm.group(2), (mm.group(3) or "-") + form)
        mm = type("M", (), {"group": lambda self, i, _m=mm: _m[i]})()
        get = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))  # noqa: E731
        agpr = int(re.match(r":\s+(\d+)", blk).group(1))
        print(f"<{mm.group(1)}, {mm.group(2)}>".ljust(44) + f" {get('vgpr_count'):5d} {agpr:5d} {get('vgpr_spill_count'):11d} {get('sgpr_spill_count'):11d} {get('private_segment_fixed_size'):9d}")
