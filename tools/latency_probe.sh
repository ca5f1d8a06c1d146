#!/bin/bash
# Runs ON THE GPU BOX: average issue-to-return latencies of the step kernel's instruction fetches, LDS, vector-memory and scalar-memory
# instructions (rocprofv3's derived InstrFetchLatency / LdsLatency / VmemLatency / SmemLatency, one pass each) and its branch count.
# usage: tools/latency_probe.sh <tag> <bench args...>   -> gpurun_out/latency_<tag>.txt
set -u
tag=$1; shift
out=$PWD/gpurun_out/latency_$tag; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
short=(--steps 2 --warmup 1 --no-cpu-baseline --no-other-configs "$@")
i=0
for pmc in "InstrFetchLatency" "LdsLatency" "VmemLatency" "SmemLatency" "SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_IFETCH"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --output-format csv -d "$out/p$i" -o p -- python bench.py "${short[@]}" > /dev/null 2> "$out/p$i.log"
done
python - "$out" $tag <<'PY' > $PWD/gpurun_out/latency_$tag.txt 2>&1
import csv, glob, sys, collections
out, cfg = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); n = collections.Counter()
for fn in glob.glob(f"{out}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "mjb_step_kernel" in r["Kernel_Name"] or "mjb_lane_env" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(acc): print(f"{cfg} {k:32s} {acc[k]/max(n[k],1):18.2f} per launch ({n[k]} launches)")
PY
cat $PWD/gpurun_out/latency_$tag.txt; tail -2 "$out"/p1.log
