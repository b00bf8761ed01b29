#!/bin/bash
# radix sort: tests, NSGA-II 10^5 and SHADE 50k timings and kernel tables
out=$GRAFT_REPO_ROOT/gpurun_out/r3
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -u -m pytest tests/test_ops_gpu.py tests/test_algorithms_gpu.py -x -q --timeout 120 --timeout-method thread -k "radix or sort or nds or nsga or crowd or select or dominated" > $out/sort_tests.log 2>&1 || { tail -30 $out/sort_tests.log; exit 1; }
tail -3 $out/sort_tests.log
timeout -k 10 200 python benchmarks/run.py nsga2_dtlz2 --pop 100000 --steps 60 --warmup 10 > $out/nsga2_100k.json 2>$out/nsga2_100k.err || { tail $out/nsga2_100k.err; exit 1; }
cat $out/nsga2_100k.json
timeout -k 10 200 python benchmarks/run.py island_shade --steps 300 --warmup 50 > $out/shade.json 2>$out/shade.err || { tail $out/shade.err; exit 1; }
cat $out/shade.json
timeout -k 10 200 python benchmarks/run.py island_shade --steps 300 --warmup 50 --graph > $out/shade_graph.json 2>$out/shade_graph.err || { tail $out/shade_graph.err; exit 1; }
cat $out/shade_graph.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_nsga2 -- python3 $GRAFT_REPO_ROOT/benchmarks/run.py nsga2_dtlz2 --pop 100000 --steps 50 --warmup 5 > $out/stats_nsga2.log 2>&1
f=$(find $out/stats_nsga2 -name '*kernel_stats.csv' | head -1)
head -45 "$f" | cut -c1-200 > $out/kernel_stats_nsga2_100k.csv
find $out/stats_nsga2 -name '*kernel_trace.csv' -delete
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_shade -- python3 $GRAFT_REPO_ROOT/benchmarks/run.py island_shade --steps 200 --warmup 20 --graph > $out/stats_shade.log 2>&1
f=$(find $out/stats_shade -name '*kernel_stats.csv' | head -1)
head -30 "$f" | cut -c1-200 > $out/kernel_stats_shade.csv
find $out/stats_shade -name '*kernel_trace.csv' -delete
cat $out/kernel_stats_nsga2_100k.csv | cut -c1-150 | head -30
