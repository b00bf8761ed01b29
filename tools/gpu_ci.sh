#!/bin/bash
# Runs on the GPU box: GPU tests, then benches.  A step that ends by signal, timeout, abort or
# segfault stops the script (nothing else is started on the GPU after such an exit).
mkdir -p gpurun_out
ok_or_testfail() { [ "$1" -eq 0 ] || [ "$1" -eq 1 ]; }
echo "[ci] pytest -m gpu"
timeout -k 10 600 python -u -m pytest tests -m gpu -q --timeout 120 --timeout-method thread -x > gpurun_out/pytest_gpu.log 2>&1
rc=$?; echo "[ci] pytest rc=$rc"; tail -3 gpurun_out/pytest_gpu.log
ok_or_testfail $rc || exit $rc
for args in "$@"; do
  tag=$(echo "$args" | tr -c 'a-zA-Z0-9' '_' | cut -c1-60)
  echo "[ci] bench $args"
  timeout -k 10 300 python -u bench.py $args > "gpurun_out/bench_${tag}.log" 2>&1
  rc=$?; echo "[ci] bench rc=$rc"; tail -2 "gpurun_out/bench_${tag}.log"
  [ $rc -eq 0 ] || exit $rc
done
