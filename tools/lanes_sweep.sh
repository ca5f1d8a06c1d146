#!/bin/bash
# config 2 by lanes per env (16 = the dense register-resident kernel, the others the generic LDS-staged ones); run on the GPU box
for l in 8 16 32 64; do
  python bench.py --lanes $l --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('lanes %2d: %6.1f M env-steps/s, kernel %.2f ms' % ($l, d['value']/1e6, d['roofline']['kernel_ms']))"
done
