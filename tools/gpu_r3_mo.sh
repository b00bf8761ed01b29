#!/bin/bash
# Generation time of the slow multi-objective algorithms at pop 10^4, with a per-kernel profile of each.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python tools/mo_survey.py 10000 ${MO_NAMES:-SRA BCEIBEA IBEA KnEA BiGE} > gpurun_out/mo_stragglers.jsonl 2> gpurun_out/mo_stragglers.err || exit 1
cat gpurun_out/mo_stragglers.jsonl | cut -c1-200
cd /tmp && export TMPDIR=/tmp
for a in ${MO_PROFILE:-SRA BCEIBEA IBEA}; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_mo_$a -o p -- python3 $GRAFT_REPO_ROOT/tools/mo_survey.py 10000 $a > /dev/null 2>&1 || exit 1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_mo_$a -name '*kernel_stats.csv' | head -1)
  echo "== $a"; head -14 "$f" | cut -d, -f1-4 | cut -c1-150
  find $GRAFT_REPO_ROOT/gpurun_out/prof_mo_$a -name '*kernel_trace.csv' -delete
done
