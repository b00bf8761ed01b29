#!/bin/bash
# Ill-conditioned objectives: the Newton–Schulz factorisation (default) against a full eigendecomposition every
# generation, same seeds, same generation counts.  Writes one JSON line per run to gpurun_out/robustness.jsonl.
#   usage (on the GPU box): bash tools/gpu_robustness.sh [steps]
set -e
steps=${1:-1500}
export PYTHONPATH="$(dirname "$0")/..:${PYTHONPATH:-}"
out=gpurun_out/robustness.jsonl
mkdir -p gpurun_out
: > "$out"
for prob in ellipsoid rosenbrock; do
  for prec in bf16 fp32; do
    echo "== $prob newton_schulz $prec" >&2
    timeout -k 10 300 python bench.py --problem $prob --precision $prec --steps "$steps" --warmup 0 --prime 20 | tail -1 >> "$out"
  done
  echo "== $prob eigh fp32" >&2
  timeout -k 10 900 python bench.py --problem $prob --precision fp32 --decomp eigh --steps "$steps" --warmup 0 --prime 2 | tail -1 >> "$out"
done
# a size at which the covariance really becomes ill-conditioned within the run (adaptation rates grow as 1/n²)
for prob in ellipsoid rosenbrock; do
  for prec in bf16 fp32; do
    echo "== small $prob newton_schulz $prec" >&2
    timeout -k 10 300 python bench.py --dim 128 --pop 512 --problem $prob --precision $prec --steps 4000 --warmup 0 --prime 20 | tail -1 >> "$out"
  done
  echo "== small $prob eigh fp32" >&2
  timeout -k 10 600 python bench.py --dim 128 --pop 512 --problem $prob --precision fp32 --decomp eigh --steps 4000 --warmup 0 --prime 2 | tail -1 >> "$out"
done
python - "$out" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    r = json.loads(line)
    c = r["config"]
    print(f'{c["model"]:34s} {c["decomposition"]:14s} {r["dtype"]:5s} ms/gen {r["ms_per_step"]:8.3f}  f(mean) {r["best_fitness_end"]:.4e}  '
          f'sigma {r["sigma_end"]:.3e}  {r.get("factorisation")}')
PY
