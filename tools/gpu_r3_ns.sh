#!/bin/bash
# Plane driver off the best case: tests, then the ill-conditioned benchmark problems with their kernel tables.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_ops_gpu.py tests/test_single_objective_algorithms.py tests/test_state.py -m gpu -x -q --timeout 180 --timeout-method thread -k "newton or ns_ or cma or factor" > gpurun_out/ns_tests.log 2>&1
rc=$?; tail -4 gpurun_out/ns_tests.log; [ $rc -ne 0 ] && exit $rc
for p in ellipsoid rosenbrock; do
  timeout -k 10 300 python bench.py --steps 300 --warmup 50 --problem $p --no-secondary > gpurun_out/bench_$p.json || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_$p.json").read().strip().splitlines()[-1])
print("$p", round(d["ms_per_step"],4), "ms", d["factorisation"], d["best_fitness_end"])
PY
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ellipsoid -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 50 --problem ellipsoid --no-secondary > /dev/null 2>&1 || exit 1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_ellipsoid -name '*kernel_stats.csv' | head -1)
cut -d, -f1-4 "$f" | cut -c1-140 | head -24
find $GRAFT_REPO_ROOT/gpurun_out/prof_ellipsoid -name '*kernel_trace.csv' -delete
