#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): bench line, rocprofv3 kernel-trace stats and the PMC passes the bench's
# `roofline.traffic` comes from.  Raw output under gpurun_out/prof_<tag>/; tools/summarise_profiles.py (run back in
# the repo) condenses it into profiles/.   usage: tools/collect_profiles.sh <tag> [bench args...]
set -u
tag=${1:-r01}; shift || true
args=("$@")
out=$PWD/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
python bench.py "${args[@]}" > "$out/bench_line.json" 2> "$out/bench_stderr.log"
short=(--steps 3 --warmup 1 --no-cpu-baseline "${args[@]}")
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o bench -- python bench.py "${short[@]}" > "$out/bench_line_under_rocprof.json" 2> "$out/rocprof_trace.log"
# counters in their own passes (no tracing domains alongside --pmc)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o p -- python bench.py "${short[@]}" > /dev/null 2> "$out/pmc_fetch.log"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o p -- python bench.py "${short[@]}" > /dev/null 2> "$out/pmc_write.log"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d "$out/pmc_sq" -o p -- python bench.py "${short[@]}" > /dev/null 2> "$out/pmc_sq.log"
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d "$out/pmc_sq2" -o p -- python bench.py "${short[@]}" > /dev/null 2> "$out/pmc_sq2.log"
# executed fp64 vector work (SURVEY.md 8d asks for the fp64 roofline next to the HBM one) + the box's measured fma peak
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d "$out/pmc_f64" -o p -- python bench.py "${short[@]}" > /dev/null 2> "$out/pmc_f64.log"
[ -x tools/ubench/fp64_peak ] && tools/ubench/fp64_peak > "$out/fp64_peak.json" 2>/dev/null
find "$out" -name "*.csv" | head -40
tail -2 "$out"/pmc_sq2.log
cat "$out/bench_line.json"
