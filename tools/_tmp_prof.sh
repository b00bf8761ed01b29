set -e
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "spea2 or SPEA2 or moselect or truncat" 2>&1 | tail -3
for a in SPEA2; do
  bash tools/gpu_prof2.sh mo_$a tools/cpu_profile_mo.py $a 2000
  grep pop= gpurun_out/prof_mo_$a/stdout.log
  head -8 gpurun_out/prof_mo_$a/kernel_stats_top.csv | cut -c1-130
done
