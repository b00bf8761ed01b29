#!/bin/bash
# folded CMA-ES tell: tests, benchmark line, kernel trace summary
out=$GRAFT_REPO_ROOT/gpurun_out/r3
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_ops_gpu.py -x -q --timeout 120 --timeout-method thread -k "cma or newton or rank_sort" > $out/fast_tests.log 2>&1 || { tail -30 $out/fast_tests.log; exit 1; }
tail -3 $out/fast_tests.log
timeout -k 10 200 python bench.py --no-secondary > $out/bench_fast.json 2> $out/bench_fast.err || { tail -20 $out/bench_fast.err; exit 1; }
cut -c1-330 $out/bench_fast.json
EVOX_AMD_CMA_WIDE=0 timeout -k 10 200 python bench.py --no-secondary > $out/bench_fast_narrow.json 2> $out/bench_fast_wide.err || exit 1
cut -c1-330 $out/bench_fast_narrow.json
EVOX_AMD_CMA_DRAW_AHEAD=0 timeout -k 10 200 python bench.py --no-secondary > $out/bench_fast_nodraw.json 2>/dev/null || exit 1
cut -c1-330 $out/bench_fast_nodraw.json
bash tools/gpu_trace.sh fast 200 --warmup 20 --no-secondary || exit 1
cat gpurun_out/trace_fast/summary.txt
