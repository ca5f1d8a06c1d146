#!/bin/bash
# Config 5 (1024 Shadow-Hand-like envs, Newton): slots per CU x launch length.  4 envs per CU = 1024 slots = one per env (a launch lasts
# as long as its slowest env's chain); 3 per CU = 768 slots (the work queue evens the chunks out).  Run on the GPU box from the repo root.
out=${1:-gpurun_out/cfg5_residency.txt}
{
echo "# config 5, one MI355X: env-steps/s by (envs, steps per launch, resident envs per CU).  3 per CU is forced with MJB_DEBUG_LDS_BYTES=53760,"
echo "# 4 per CU on short launches with MJB_DEBUG_NO_SLOT_CAP=1 (the host caps launches under 400 steps at 3 per CU when every env would get its own slot)."
for envs in 1024 2048; do for k in 100 1000; do for occ in 3 4; do
  if [ $occ = 3 ]; then export MJB_DEBUG_LDS_BYTES=53760; unset MJB_DEBUG_NO_SLOT_CAP; else unset MJB_DEBUG_LDS_BYTES; export MJB_DEBUG_NO_SLOT_CAP=1; fi
  v=$(python bench.py --config 5 --envs $envs --substeps $k --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.3f M env-steps/s, kernel %.2f ms' % (d['value']/1e6, d['roofline']['kernel_ms']))")
  echo "envs $envs  steps/launch $k  envs/CU $occ :  $v"
done; done; done
unset MJB_DEBUG_LDS_BYTES MJB_DEBUG_NO_SLOT_CAP
echo "# defaults (no knobs):"
for k in 100 1000; do
  v=$(python bench.py --config 5 --substeps $k --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.3f M env-steps/s' % (d['value']/1e6))")
  echo "envs 1024  steps/launch $k  default :  $v"
done
} | tee $out
