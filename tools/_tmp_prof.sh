set -e
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "stochastic_ranking or SRA or sequential_selection" 2>&1 | tail -3
timeout -k 10 200 python tools/cpu_profile_mo.py SRA 2000 2>&1 | grep "pop="
timeout -k 10 200 python tools/cpu_profile_mo.py SRA 10000 2>&1 | grep "pop="
