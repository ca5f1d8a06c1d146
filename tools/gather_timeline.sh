#!/bin/bash
# Runs ON THE GPU BOX: kernel timeline of bench.py's RCCL path with ONE rank (MJB_BENCH_FORCE_GATHER=1): the step kernel on the
# engine's stream, the staging copies, and the all-gather / all-reduce kernels on the side stream UNDER the next launch.
# Writes gpurun_out/gather_timeline/timeline.txt (condensed by the python below).
set -u
out=$PWD/gpurun_out/gather_timeline
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
MJB_BENCH_FORCE_GATHER=1 MASTER_PORT=29577 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$out/trace" -o t -- \
  python bench.py --steps 4 --warmup 1 --substeps 1000 --no-cpu-baseline --no-other-configs > "$out/bench_line.json" 2> "$out/stderr.log"
python - "$out" <<'PY'
import csv, glob, sys
out = sys.argv[1]
kt = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)
rows = []
for f in kt:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", r.get("Queue_Id", "?")), r["Kernel_Name"][:70]))
for f in glob.glob(out + "/trace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy", r.get("Direction", "copy")))
rows.sort()
steps = [r for r in rows if "mjb_step_kernel" in r[3]]
with open(out + "/timeline.txt", "w") as fo:
    if len(steps) < 4:
        fo.write("too few step kernels in the trace\n")
    else:
        t0 = steps[-4][0]
        fo.write("# kernel / copy timeline of the last 4 fused launches (1000 steps each), times in ms from the first of them;\n")
        fo.write("# stream = rocprofv3 Stream_Id / Queue_Id.  The all-gather / all-reduce kernels of launch k run while the step kernel of launch k+1 does.\n")
        for s, e, q, n in rows:
            if s < t0 - 2_000_000:
                continue
            if "mjb_step_kernel" in n:
                n = "mjb_step_kernel<16,0,12>  (engine stream)"
            fo.write(f"{(s - t0) / 1e6:10.3f} .. {(e - t0) / 1e6:10.3f} ms  [{q:>6}]  {n}\n")
        # overlap evidence
        coll = [r for r in rows if ("ccl" in r[3].lower() or "AllGather" in r[3] or "AllReduce" in r[3] or "Generic" in r[3]) and r[0] >= t0]
        inside = sum(1 for c in coll for s in steps[-4:] if c[0] >= s[0] and c[1] <= s[1])
        fo.write(f"# collective kernels after t0: {len(coll)}; of them running entirely INSIDE a step kernel's interval: {inside}\n")
print(open(out + "/timeline.txt").read()[-3000:])
PY
