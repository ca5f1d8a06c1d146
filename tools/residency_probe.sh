#!/bin/bash
# Config 5 (Shadow-Hand-like envs, Newton): resident envs per CU x launch length x chunked work queue.  4 envs per CU = 1024 slots on
# an MI355X; with envs <= slots every env has a slot of its own (a launch lasts as long as its slowest env's chain; the host then
# launches unchunked, mjb_api.hip: launch).  Run on the GPU box from the repo root.
out=${1:-gpurun_out/cfg5_residency.txt}
run() { python bench.py --config 5 --envs $1 --substeps $2 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.3f M' % (d['value']/1e6))"; }
{
echo "# config 5, one MI355X: M env-steps/s by (envs, steps per launch, resident envs per CU, chunking).  3 per CU is forced with"
echo "# MJB_DEBUG_LDS_BYTES=53760; 'chunked' forces the work queue with MJB_DEBUG_CHUNK=max(5, steps/20); 'default' is the host's own choice."
for envs in 768 1024 1280 2048; do for k in 100 1000; do
  c=$(( k / 20 )); [ $c -lt 5 ] && c=5
  a=$(MJB_DEBUG_LDS_BYTES=53760 MJB_DEBUG_CHUNK=$c run $envs $k)
  b=$(MJB_DEBUG_LDS_BYTES=53760 MJB_DEBUG_NO_CHUNKS=1 run $envs $k)
  d=$(MJB_DEBUG_CHUNK=$c run $envs $k)
  e=$(MJB_DEBUG_NO_CHUNKS=1 run $envs $k)
  f=$(run $envs $k)
  echo "envs $envs steps/launch $k : 3/CU chunked $a | 3/CU unchunked $b | 4/CU chunked $d | 4/CU unchunked $e | default $f"
done; done
} | tee $out
