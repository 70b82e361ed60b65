#!/bin/bash
# usage: tools/gpurun_retry.sh <timeout_s> '<command>'   -- retries while the pool has no free slot (exit 3)
T=$1; shift
for i in $(seq 1 10); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 60
done
exit 3
