#!/bin/bash
# gpurun with retries ONLY for "no slot / no box" (exit 3: nothing ran, nothing charged).  Usage: tools/gpurun_retry.sh <timeout> '<command>'
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 60
done
exit 3
