set -e
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "ibea or IBEA or removal_loop or sequential_selection" 2>&1 | tail -5
for a in IBEA; do
  bash tools/gpu_prof2.sh mo_$a tools/cpu_profile_mo.py $a 2000
  grep pop= gpurun_out/prof_mo_$a/stdout.log
  python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_mo_IBEA/kernel_stats_top.csv')))
for r in rows[:4]: print(r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage'])
PY
done
