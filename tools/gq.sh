#!/bin/bash
# quick GPU check: gpu tests (optional -k filter as $1) + the three bench configs, one line each
K="${1:-}"
if [ -n "$K" ]; then python -m pytest tests -m gpu -x -q -k "$K" 2>&1 | tail -2; else python -m pytest tests -m gpu -x -q 2>&1 | tail -2; fi
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('cfg2 %.1f M  kernel %.2f ms' % (d['value']/1e6, d['roofline']['kernel_ms']))
for k,v in d.get('other_configs',{}).items():
    print('cfg%s %.2f M  kernel %.2f ms  %s' % (k, v.get('value',0)/1e6, v.get('roofline',{}).get('kernel_ms',0), v.get('error','')))"
