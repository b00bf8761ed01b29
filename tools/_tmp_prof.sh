bash tools/gpu_ci.sh "" || exit $?
timeout -k 10 400 python tools/mo_survey.py 2000 > gpurun_out/mo_survey_2000.jsonl 2> gpurun_out/mo_survey_2000.err || exit $?
python3 -c "
import json
for l in open('gpurun_out/mo_survey_2000.jsonl'):
    d=json.loads(l); print(d['algorithm'], round(d['ms_per_generation'],2), round(d['igd_after_65_generations'],4))
"
