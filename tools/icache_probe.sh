#!/bin/bash
# Runs ON THE GPU BOX: instruction-cache counters of the step kernel per config (is the 130 .. 680 KB step loop fetch bound?).
# usage: tools/icache_probe.sh   -> gpurun_out/icache/<cfg>.txt
set -u
out=$PWD/gpurun_out/icache; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQC_[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT_INST[A-Z_]*" | sort -u > "$out/available.txt"
for cfg in 2 3 5; do
  short=(--steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --config $cfg)
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d "$out/c$cfg" -o p -- python bench.py "${short[@]}" > /dev/null 2> "$out/c$cfg.log"
  rocprofv3 --pmc SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d "$out/d$cfg" -o p -- python bench.py "${short[@]}" > /dev/null 2> "$out/d$cfg.log"
  python - "$out" $cfg <<'PY'
import csv, glob, sys, collections
out, cfg = sys.argv[1], sys.argv[2]
for tag in "cd":
    acc = collections.defaultdict(float); n = collections.Counter()
    for fn in glob.glob(f"{out}/{tag}{cfg}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if "mjb_step_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in acc: print(f"cfg{cfg} {k:32s} {acc[k]/max(n[k],1):16.0f} per launch ({n[k]} launches)")
PY
done > "$out/summary.txt" 2>&1
cat "$out/summary.txt"; tail -3 "$out"/c5.log
