set -e
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "ibea or IBEA or stochastic_ranking or SRA or sequential_selection or removal_loop" 2>&1 | tail -5
timeout -k 10 200 python tools/cpu_profile_mo.py SRA 2000 > gpurun_out/prof_SRA.txt 2>&1
head -3 gpurun_out/prof_SRA.txt
timeout -k 10 200 python tools/cpu_profile_mo.py IBEA 2000 > gpurun_out/prof_IBEA.txt 2>&1
head -3 gpurun_out/prof_IBEA.txt
timeout -k 10 200 python tools/cpu_profile_mo.py IBEA 10000 > gpurun_out/prof_IBEA.txt 2>&1
head -3 gpurun_out/prof_IBEA.txt
