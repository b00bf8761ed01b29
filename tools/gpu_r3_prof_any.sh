#!/bin/bash
# kernel table of an arbitrary python tool: gpu_r3_prof_any.sh <name> <script> [args...]
cd $GRAFT_REPO_ROOT
name=$1; shift
out=gpurun_out/prof_$name; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 "$@" > $out/run.log 2>&1 || { tail -20 $out/run.log; exit 1; }
grep '^{' $out/run.log | cut -c1-300
f=$(find $out -name 'p_kernel_stats.csv' | sort | tail -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", round(tot / 1e6, 2))
for r in rows[:18]:
    print(f'{float(r["TotalDurationNs"])/tot*100:5.1f}%  calls {r["Calls"]:>7}  avg {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:110]}')
PY
find $out -name '*kernel_trace.csv' -delete
