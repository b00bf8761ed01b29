#!/bin/bash
# kernel tables of the algorithms whose generation is bound by a small dense factorisation (dim 100, pop 1024)
out=$GRAFT_REPO_ROOT/gpurun_out/small
mkdir -p $out
cd $GRAFT_REPO_ROOT
for a in ${SMALL_NAMES:-AMaLGaM RMES ASEBO IPOPCMAES}; do
  echo "== $a" | tee -a $out/summary.txt
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$a -o p -- python3 tools/graph_survey.py --only $a --steps 50 --warmup 20 > $out/$a.log 2>&1 || { tail -20 $out/$a.log; exit 1; }
  grep '^{' $out/$a.log | cut -c1-200 | tee -a $out/summary.txt
  f=$(find $out/prof_$a -name 'p_kernel_stats.csv' | sort | tail -1)
  python - "$f" <<'PY' | tee -a $out/summary.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:12]:
    print(f'{float(r["TotalDurationNs"])/tot*100:5.1f}%  calls {r["Calls"]:>6}  avg {float(r["AverageNs"])/1e3:8.1f} us  {r["Name"][:110]}')
PY
done
