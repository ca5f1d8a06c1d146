#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit code 3 = nothing charged).  usage: tools/grun.sh [--timeout S] -- 'cmd'
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
