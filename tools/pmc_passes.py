#!/usr/bin/env python3
"""pmc_passes.py <out.json> <kernel-substring> -- <command...>   (runs ON THE GPU BOX)
Runs <command> under rocprofv3 once per counter group (each its own pass, no tracing domains beside --pmc) and writes the mean
counter value per dispatch of the kernels whose name contains <kernel-substring>, keyed by (kernel, grid size)."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

GROUPS = [
    ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY"],
    ["SQ_INSTS_SMEM", "SQ_INSTS_VMEM", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM", "SQ_INSTS_FLAT"],
    ["SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_IFETCH", "SQ_INST_LEVEL_VMEM", "SQ_INST_LEVEL_SMEM", "SQ_INST_CYCLES_VMEM"],
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
]


def main():
    out, sub = sys.argv[1], sys.argv[2]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    os.environ["TMPDIR"] = "/tmp"
    res = {}
    meta = {}
    for gi, grp in enumerate(GROUPS):
        d = "/tmp/pmc_pass_%d" % gi
        shutil.rmtree(d, ignore_errors=True)
        p = subprocess.run(["rocprofv3", "--pmc", *grp, "--output-format", "csv", "-d", d, "-o", "p", "--", *cmd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
        if not files:
            res.setdefault("_errors", []).append({"group": grp, "rc": p.returncode, "stderr": p.stderr[-400:]})
            continue
        acc = {}
        for row in csv.DictReader(open(files[0])):
            if sub not in row["Kernel_Name"]:
                continue
            key = row["Kernel_Name"].split("(")[0][-60:] + " grid=" + row["Grid_Size"]
            acc.setdefault(key, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            meta.setdefault(key, {k: row[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")})
        for key, cs in acc.items():
            for c, v in cs.items():
                res.setdefault(key, {})[c] = {"n": len(v), "mean": sum(v) / len(v)}
    for key in meta:
        res[key]["_dispatch"] = meta[key]
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1)[:6000])


if __name__ == "__main__":
    main()
