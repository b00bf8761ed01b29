#!/bin/bash
# usage: tools/gpu_submit.sh <tag> <timeout_s> '<command>'
# Submits one gpurun call; when the pool has no free box (exit 3, nothing ran) waits and asks
# again.  Output of the client goes to /tmp/gpurun_<tag>.out
tag=$1; limit=$2; shift 2
out=/tmp/gpurun_${tag}.out
# the exact PID of this loop, so that it can be stopped with `kill $(cat /tmp/gpurun_<tag>.pid)`
echo $$ > /tmp/gpurun_${tag}.pid
for attempt in $(seq 1 80); do
  /usr/local/graft/bin/gpurun --timeout "$limit" -- "$@" > "$out" 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then echo "done rc=$rc" >> "$out"; exit $rc; fi
  sleep 100
done
echo "gave up: no box" >> "$out"
exit 3
