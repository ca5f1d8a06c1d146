#!/bin/bash
# Runs ON THE GPU BOX: instruction-cache / issue counters of the step kernel for ONE bench command line.
# usage: tools/icache_probe2.sh <tag> <bench args...>   -> gpurun_out/icache_<tag>.txt
set -u
tag=$1; shift
out=$PWD/gpurun_out/icache_$tag; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
short=(--steps 2 --warmup 1 --no-cpu-baseline --no-other-configs "$@")
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d "$out/c" -o p -- python bench.py "${short[@]}" > /dev/null 2> "$out/c.log"
rocprofv3 --pmc SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d "$out/d" -o p -- python bench.py "${short[@]}" > /dev/null 2> "$out/d.log"
rocprofv3 --pmc SQ_WAIT_ANY SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d "$out/e" -o p -- python bench.py "${short[@]}" > /dev/null 2> "$out/e.log"
python - "$out" $tag <<'PY' > $PWD/gpurun_out/icache_$tag.txt 2>&1
import csv, glob, sys, collections
out, cfg = sys.argv[1], sys.argv[2]
for tag in "cde":
    acc = collections.defaultdict(float); n = collections.Counter()
    for fn in glob.glob(f"{out}/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if "mjb_step_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in acc: print(f"{cfg} {k:32s} {acc[k]/max(n[k],1):16.0f} per launch ({n[k]} launches)")
PY
cat $PWD/gpurun_out/icache_$tag.txt; tail -2 "$out"/e.log
