for a in CLPSO SLPSOGS SLPSOUS FIPS CR_FM_NES SwmmPSO FSPSO; do
timeout -k 10 100 python tools/_tmp_dbg.py $a 2>&1 | grep -v amdgpu.ids | tail -4
done
timeout -k 10 600 python tools/graph_survey.py --only CLPSO,SLPSOGS,SLPSOUS,FIPS,CR_FM_NES,SwmmPSO,FSPSO 2>&1 | grep "^{"
