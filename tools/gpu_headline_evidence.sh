#!/bin/bash
# Evidence for the headline generation (bench.py defaults): kernel statistics, hardware counters of its kernels, the
# stand-alone GEMM harness, and the benchmark line itself.  Each GPU step has its own time limit; a failure ends the script.
# Output under gpurun_out/headline/.
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/headline
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python bench.py --steps 1000 --warmup 50 > $out/bench_1000.json
timeout -k 10 200 python bench.py > $out/bench_default.json
timeout -k 10 200 python bench.py --precision fp32 > $out/bench_fp32.json
timeout -k 10 200 python bench.py --force-sharded --steps 1000 --warmup 50 > $out/bench_sharded_world1.json
[ -x tools/bin/gemm_check ] && timeout -k 10 120 tools/bin/gemm_check > $out/gemm_check.txt 2>&1 || true
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 > $out/stats_stdout.log 2>&1
f=$(find $out/stats -name '*kernel_stats.csv' | head -1)
head -40 "$f" > $out/kernel_stats_bf16.csv
find $out/stats -name '*kernel_trace.csv' -delete
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_sharded -- python3 $GRAFT_REPO_ROOT/bench.py --force-sharded --steps 200 --warmup 20 > $out/stats_sharded_stdout.log 2>&1
f=$(find $out/stats_sharded -name '*kernel_stats.csv' | head -1)
head -40 "$f" > $out/kernel_stats_bf16_sharded_world1.csv
find $out/stats_sharded -name '*kernel_trace.csv' -delete
pmc() {
  name=$1; shift
  mkdir -p $out/pmc_$name
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_$name -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --prime 6 > $out/pmc_$name/stdout.log 2>&1
  echo "[pmc] $name done"
}
pmc sq1 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_WAVES
pmc sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $out/pmc_sq1 $out/pmc_sq2 $out/pmc_tcc > $out/pmc_headline_kernels.json
find $out -name '*counter_collection.csv' -delete
find $out -name '*agent_info.csv' -delete
echo done
