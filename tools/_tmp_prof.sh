timeout -k 10 400 python tools/mo_survey.py 2000 > gpurun_out/mo_survey_2000.jsonl 2> gpurun_out/mo_survey_2000.err || exit $?
timeout -k 10 900 python tools/mo_survey.py 10000 > gpurun_out/mo_survey_10000.jsonl 2> gpurun_out/mo_survey_10000.err || exit $?
python3 -c "
import json
a={json.loads(l)['algorithm']:json.loads(l) for l in open('gpurun_out/mo_survey_2000.jsonl')}
b={json.loads(l)['algorithm']:json.loads(l) for l in open('gpurun_out/mo_survey_10000.jsonl')}
for k in a: print(k, round(a[k]['ms_per_generation'],2), round(a[k]['igd_after_65_generations'],4), '|', round(b[k]['ms_per_generation'],2), round(b[k]['igd_after_65_generations'],4))
"
