for a in IMMOEA MOEADM2M KnEA EAGMOEAD BiGE TDEA; do
timeout -k 10 200 python tools/cpu_profile_mo.py $a 2000 > gpurun_out/prof_$a.txt 2>&1 || exit $?
head -34 gpurun_out/prof_$a.txt | grep -v "^$\|Ordered\|List reduced\|function calls" | cut -c1-150
done
