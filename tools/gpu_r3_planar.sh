#!/bin/bash
# kernel table of a graph-replayed OpenES generation on the planar hopper
cd $GRAFT_REPO_ROOT
out=gpurun_out/planar; mkdir -p $out
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o p -- python3 tools/planar_es_timing.py ${1:-hopper} 4096 100 > $out/run.log 2>&1 || { tail -20 $out/run.log; exit 1; }
grep '^{' $out/run.log
f=$(find $out/prof -name 'p_kernel_stats.csv' | sort | tail -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6)
for r in rows[:16]:
    print(f'{float(r["TotalDurationNs"])/tot*100:5.1f}%  calls {r["Calls"]:>7}  avg {float(r["AverageNs"])/1e3:8.1f} us  {r["Name"][:100]}')
PY
find $out/prof -name '*kernel_trace.csv' -delete
