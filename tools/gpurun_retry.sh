#!/bin/bash
# Usage: tools/gpurun_retry.sh <timeout-seconds> '<command>'   -- retries while gpurun reports "no slot / no box" (exit 3)
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 120
done
exit 3
