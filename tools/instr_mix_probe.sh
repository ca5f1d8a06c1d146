#!/bin/bash
set -u
tag=$1; shift
out=$PWD/gpurun_out/mix_$tag; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
short=(--steps 2 --warmup 1 --no-cpu-baseline --no-other-configs "$@")
i=0
for pmc in "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32" "SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_VSKIPPED SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --output-format csv -d "$out/p$i" -o p -- python bench.py "${short[@]}" > /dev/null 2> "$out/p$i.log"
done
python - "$out" $tag <<'PY' > $PWD/gpurun_out/mix_$tag.txt 2>&1
import csv, glob, sys, collections
out, cfg = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); n = collections.Counter()
for fn in glob.glob(f"{out}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "mjb_step_kernel" in r["Kernel_Name"] or "mjb_lane_env" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(acc): print(f"{cfg} {k:32s} {acc[k]/max(n[k],1):18.2f} per launch ({n[k]} launches)")
PY
cat $PWD/gpurun_out/mix_$tag.txt; tail -2 "$out"/p1.log
