#!/bin/bash
# usage: tools/gpu_trace.sh <name> <steps> <bench args...> : kernel trace of a bench run, summarised per step
name=$1; steps=$2; shift; shift
out=$GRAFT_REPO_ROOT/gpurun_out/trace_$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps $steps "$@" > $out/stdout.log 2>&1
rc=$?
cd $GRAFT_REPO_ROOT
f=$(find $out -name '*kernel_trace.csv' | head -1)
[ -n "$f" ] && python3 tools/trace_summary.py "$f" $steps 0.6 > $out/summary.txt 2>&1
find $out -name '*kernel_trace.csv' -delete
exit $rc
