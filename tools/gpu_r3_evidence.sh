#!/bin/bash
# Evidence for the documented numbers: the driver-style bench line (with the secondary configurations) and the graph
# survey of every single-objective algorithm.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python bench.py > gpurun_out/bench_headline_r3.json 2> gpurun_out/bench_headline_r3.err || exit 1
cut -c1-300 gpurun_out/bench_headline_r3.json
timeout -k 10 900 python tools/graph_survey.py > gpurun_out/graph_survey_r3.log 2>&1 || exit 1
grep '^{' gpurun_out/graph_survey_r3.log > gpurun_out/graph_survey_dim100_pop1024.jsonl
wc -l gpurun_out/graph_survey_dim100_pop1024.jsonl
