#!/bin/bash
# usage: tools/gpu_prof_py.sh <name> <script.py> <args...> : kernel-trace statistics of a python script (top kernels kept)
name=$1; shift
mkdir -p gpurun_out/prof_$name
cd /tmp && export TMPDIR=/tmp
script=$GRAFT_REPO_ROOT/$1; shift
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$name -- python3 $script "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$name/stdout.log 2>&1
rc=$?
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_$name -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -16 "$f" > gpurun_out/prof_$name/kernel_stats_top.csv
find gpurun_out/prof_$name -name '*kernel_trace.csv' -delete
find gpurun_out/prof_$name -name '*agent_info.csv' -delete
exit $rc
