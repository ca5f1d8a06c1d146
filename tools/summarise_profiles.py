#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (written by tools/collect_profiles.sh on the GPU box) into profiles/:
   <tag>_bench_line.json, <tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), <tag>_pmc_summary.json
   (mean counter value per dispatch of the step kernel; FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 reports them).
usage: summarise_profiles.py <tag> [name-suffix]"""
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
suffix = sys.argv[2] if len(sys.argv) > 2 else ""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
name = tag + suffix


def first(pattern):
    g = glob.glob(os.path.join(src, pattern), recursive=True)
    return g[0] if g else None


for f, t in (("bench_line.json", "_bench_line.json"), ("bench_line_under_rocprof.json", "_bench_line_under_rocprof.json")):
    p = os.path.join(src, f)
    if os.path.exists(p):
        lines = [l for l in open(p).read().splitlines() if l.startswith("{")]
        if lines:
            open(os.path.join(dst, name + t), "w").write(lines[-1] + "\n")
ks = first("trace/**/*kernel_stats.csv")
if ks:
    shutil.copy(ks, os.path.join(dst, name + "_kernel_stats.csv"))
summary = {}
meta = None
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_f64"):
    cc = first(d + "/**/*counter_collection.csv")
    if not cc:
        continue
    acc = {}
    for row in csv.DictReader(open(cc)):
        if "mjb_step_kernel" not in row["Kernel_Name"] and "mjb_lane_env_" not in row["Kernel_Name"]:
            continue
        acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        if meta is None:
            meta = {k: row[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size",
                                        "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
    for k, v in acc.items():
        # (`mean_per_dispatch` keeps its name for bench.py, but is the MEDIAN since round 5: the first long launch of a batch on kernel
        #  variant 4 still runs on the 64-row frame -- the frame policy looks at the previous launch -- and on the power grasp that one launch
        #  moves 111 GB through the HBM row blocks against 0.5 GB for every later one)
        sv = sorted(v)
        summary[k] = {"dispatches": len(v), "mean_per_dispatch": sv[len(sv) // 2], "mean": sum(v) / len(v), "min": sv[0], "max": sv[-1]}
if summary:
    pk = os.path.join(src, "fp64_peak.json")
    if os.path.exists(pk):
        try:
            summary["fp64_peak_measured"] = json.loads(open(pk).read().splitlines()[-1])
        except Exception:
            pass
    f64 = {k: summary[k]["mean_per_dispatch"] for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64",
                                                       "SQ_INSTS_VALU_TRANS_F64") if k in summary}
    if len(f64) == 4:
        # wave-level instruction counts; one instruction = 64 lanes (idle lanes of a partially filled stage included)
        summary["fp64_executed_flops_per_dispatch"] = 64.0 * (f64["SQ_INSTS_VALU_ADD_F64"] + f64["SQ_INSTS_VALU_MUL_F64"] +
                                                              2.0 * f64["SQ_INSTS_VALU_FMA_F64"] + f64["SQ_INSTS_VALU_TRANS_F64"])
    summary["dispatch_meta"] = meta
    sys.path.insert(0, root)
    from mujoco_ros_pkgs_amd import provenance  # noqa: E402
    summary["csrc_sha"] = provenance.csrc_sha()  # the kernel sources these counters belong to (bench.py checks it)
    bl = os.path.join(dst, name + "_bench_line_under_rocprof.json")
    if os.path.exists(bl):
        cfg = json.loads(open(bl).read())["config"]
        summary["bench_config"] = cfg
        summary["envs"], summary["substeps"] = cfg.get("envs_per_gpu"), cfg.get("physics_steps_per_launch")
    summary["notes"] = ("rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE and two SQ passes, each its own run, no tracing "
                        "domains) of `python bench.py --steps 3 --warmup 1 --no-cpu-baseline ...`; means over the step-kernel "
                        "dispatches.  FETCH_SIZE / WRITE_SIZE are KiB (HBM bytes = value * 1024); per MI355X_MICROARCH.md "
                        "the gfx950 x2 correction applies to wide (16 B/lane) streaming reads only -- this kernel reads 8 B/lane, "
                        "uncalibrated, so the raw value is reported.")
    json.dump(summary, open(os.path.join(dst, name + "_pmc_summary.json"), "w"), indent=1)
print("wrote", sorted(f for f in os.listdir(dst) if f.startswith(name)))
