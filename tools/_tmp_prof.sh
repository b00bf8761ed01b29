set -e
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
timeout -k 10 400 python tools/mo_survey.py 2000 > gpurun_out/mo_survey_2000.jsonl 2> gpurun_out/mo_survey_2000.err
python3 -c "
import json
for l in open('gpurun_out/mo_survey_2000.jsonl'):
    d=json.loads(l); print(d['algorithm'], round(d.get('ms_per_generation',-1),2), d.get('igd_after_65_generations'), d.get('error'))
"
