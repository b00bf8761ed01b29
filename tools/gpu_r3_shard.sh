#!/bin/bash
# Row-sharded folded tail: tests, then the sharded workflow's cost at one rank against the single-process pair.
set -o pipefail
mkdir -p gpurun_out
echo "== tests" && timeout -k 10 600 python -u -m pytest tests/test_ops_gpu.py -x -q --timeout 180 --timeout-method thread \
  -k "folded or sharded or auto_measures or native_generation or expm or small_svd" > gpurun_out/shard_tests.log 2>&1
rc=$?; tail -5 gpurun_out/shard_tests.log; [ $rc -ne 0 ] && exit $rc
echo "== bench single" && timeout -k 10 300 python bench.py --steps 300 --warmup 30 --no-secondary > gpurun_out/bench_single.json || exit 1
tail -1 gpurun_out/bench_single.json | cut -c1-200
echo "== bench sharded W=1 (folded tail)" && timeout -k 10 300 python bench.py --steps 300 --warmup 30 --no-secondary --force-sharded > gpurun_out/bench_sharded_w1.json || exit 1
tail -1 gpurun_out/bench_sharded_w1.json | cut -c1-200
echo "== bench sharded W=1 (launch chain)" && EVOX_AMD_CMA_FAST_SHARDED=0 timeout -k 10 300 python bench.py --steps 300 --warmup 30 --no-secondary --force-sharded > gpurun_out/bench_sharded_w1_chain.json || exit 1
tail -1 gpurun_out/bench_sharded_w1_chain.json | cut -c1-200
