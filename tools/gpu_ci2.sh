#!/bin/bash
# GPU tests, then the secondary benchmark configurations named on the command line.
mkdir -p gpurun_out
echo "[ci2] pytest -m gpu"
timeout -k 10 600 python -u -m pytest tests -m gpu -q --timeout 120 --timeout-method thread -x > gpurun_out/pytest_gpu.log 2>&1
rc=$?; echo "[ci2] pytest rc=$rc"; tail -5 gpurun_out/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
i=0
for args in "$@"; do
  i=$((i+1))
  echo "[ci2] $args"
  timeout -k 10 400 python -u benchmarks/run.py $args > "gpurun_out/b2_${i}.log" 2>&1
  rc=$?; echo "[ci2] rc=$rc"; tail -2 "gpurun_out/b2_${i}.log"
  [ $rc -eq 0 ] || exit $rc
done
