#!/bin/bash
# usage: scripts/gpu_try.sh TIMEOUT 'command'  — retries only while gpurun answers "no box / slot free" (rc 3, nothing charged)
T=$1; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 120
done
exit 3
