set -e
timeout -k 10 200 python tools/kernels_pmc.py 2>&1 | tail -3
bash tools/gpu_pmc.sh
python tools/pmc_summary.py gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 gpurun_out/pmc_tcc > gpurun_out/pmc_hand_written_kernels.json
find gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 gpurun_out/pmc_tcc -name '*.csv' -size +2M -delete
python3 -c "
import json
d=json.load(open('gpurun_out/pmc_hand_written_kernels.json')); print(len(d), 'kernels'); print(sorted(d)[:80])
"
