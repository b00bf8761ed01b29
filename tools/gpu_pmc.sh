#!/bin/bash
# Hardware-counter passes over the hand-written kernels (one rocprofv3 run per counter group;
# counters only, no tracing domains).  Output: gpurun_out/pmc_<group>/ ... counter_collection.csv
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$name
  mkdir -p $out
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/kernels_pmc.py > $out/stdout.log 2>&1
  rc=$?; echo "[pmc] $name rc=$rc"; [ $rc -eq 0 ] || exit $rc
}
run sq1 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_WAVES
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
# the raw per-dispatch tables are large (tens of MiB): summarise on the box and keep the summary only
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $GRAFT_REPO_ROOT/gpurun_out/pmc_sq1 $GRAFT_REPO_ROOT/gpurun_out/pmc_sq2 \
  $GRAFT_REPO_ROOT/gpurun_out/pmc_tcc > $GRAFT_REPO_ROOT/gpurun_out/pmc_hand_written_kernels.json \
  && rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_sq1 $GRAFT_REPO_ROOT/gpurun_out/pmc_sq2 $GRAFT_REPO_ROOT/gpurun_out/pmc_tcc
