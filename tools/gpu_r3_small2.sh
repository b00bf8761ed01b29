#!/bin/bash
# survey rows of the algorithms bound by a small factorisation, at two settings of the eager device threshold
cd $GRAFT_REPO_ROOT
out=gpurun_out/small2; mkdir -p $out
timeout -k 10 300 python -u -m pytest tests/test_ops_gpu.py -q --timeout 120 --timeout-method thread -k "small_eigh or small_cholesky" > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -1 $out/tests.log
timeout -k 10 200 python tools/eigh_spd_timing.py 100 128 2>&1 | grep "^{" | tee $out/timing.jsonl
for thr in 64 192; do
  echo "== EVOX_AMD_DEVICE_EIGH_MAX=$thr"
  EVOX_AMD_DEVICE_EIGH_MAX=$thr timeout -k 10 400 python tools/graph_survey.py --only AMaLGaM,RMES,IPOPCMAES,BIPOPCMAES,CMAES,MAES,ASEBO > $out/survey_$thr.log 2>&1 || { tail -20 $out/survey_$thr.log; exit 1; }
  grep '^{' $out/survey_$thr.log | cut -c1-220
done
