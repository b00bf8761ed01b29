#!/bin/bash
# usage: tools/gpu_prof2.sh <name> <script.py> <args...> : kernel-trace profile of a short python run
name=$1; script=$2; shift 2
mkdir -p gpurun_out/prof_$name
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$name -- python3 $GRAFT_REPO_ROOT/$script "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$name/stdout.log 2>&1
rc=$?
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_$name -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -30 "$f" > gpurun_out/prof_$name/kernel_stats_top.csv
find gpurun_out/prof_$name -name '*kernel_trace.csv' -delete
exit $rc
