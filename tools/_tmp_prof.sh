set -e
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "bf16 or cma or CMA" 2>&1 | tail -3
timeout -k 10 200 python bench.py | cut -c1-330
timeout -k 10 200 python bench.py --precision fp32 | cut -c1-330
