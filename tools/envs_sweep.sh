#!/bin/bash
# env-steps/s by batch size (config N: tools/envs_sweep.sh N e1 e2 ...); run on the GPU box
c=$1; shift
for e in "$@"; do
  python bench.py --config $c --envs $e --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('config $c envs %6d: %7.2f M env-steps/s, kernel %.2f ms' % ($e, d['value']/1e6, d['roofline']['kernel_ms']))"
done
