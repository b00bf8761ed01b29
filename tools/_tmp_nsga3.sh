set -e
timeout -k 10 300 python -u -m pytest tests/test_ops_gpu.py -x -q --timeout 120 --timeout-method thread -k "small_solve or niche_fill" 2>&1 | tail -5
timeout -k 10 400 python tools/mo_survey.py 2000 > gpurun_out/mo_survey_2000.jsonl 2> gpurun_out/mo_survey_2000.err
cat gpurun_out/mo_survey_2000.jsonl
