#!/bin/bash
# usage: tools/gpu_submit.sh <tag> <timeout_s> '<command>'
# Submits one gpurun call; when the pool has no free box (exit 3, nothing ran) waits and asks
# again.  Output of the client goes to /tmp/gpurun_<tag>.out
tag=$1; limit=$2; shift 2
out=/tmp/gpurun_${tag}.out
for attempt in 1 2 3 4 5 6 7 8 9 10 11 12; do
  /usr/local/graft/bin/gpurun --timeout "$limit" -- "$@" > "$out" 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then echo "done rc=$rc" >> "$out"; exit $rc; fi
  sleep 60
done
echo "gave up: no box" >> "$out"
exit 3
