#!/bin/bash
# Secondary benchmark configurations on the GPU box; stops at the first step that does not exit 0.
mkdir -p gpurun_out
run() {  # tag, timeout, args...
  tag=$1; lim=$2; shift 2
  echo "[b2] $tag: $*"
  timeout -k 10 "$lim" python -u benchmarks/run.py "$@" > "gpurun_out/b2_${tag}.log" 2>&1
  rc=$?; echo "[b2] rc=$rc"; tail -2 "gpurun_out/b2_${tag}.log"
  [ $rc -eq 0 ] || exit $rc
}
run cma_bf16 200 cmaes_sphere_bf16 --steps 50 --warmup 10
run nsga2_10k 200 nsga2_dtlz2 --pop 10000 --steps 5 --warmup 2
run openes 300 openes_mlp --steps 5 --warmup 2
run openes_bf16 300 openes_mlp --steps 5 --warmup 2 --param-dtype bf16
run shade 200 island_shade --steps 20 --warmup 3
run nsga2_100k 400 nsga2_dtlz2 --pop 100000 --steps 2 --warmup 1
