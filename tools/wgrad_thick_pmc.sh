#!/bin/bash
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; RES=$OUT/thick_wgrad_pmc.txt; : > $RES
cd /tmp && export TMPDIR=/tmp
for ctr in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE" "TA_BUSY_avr"; do
  D=$OUT/pmc_tmp; rm -rf $D; mkdir -p $D
  rocprofv3 --pmc $ctr --kernel-trace -d $D -- python3 $ROOT/tools/wgrad_thick_probe.py pmc > /dev/null 2>&1 || true
  DB=$(find $D -name '*.db' | head -1)
  echo "## $ctr" >> $RES
  python3 $ROOT/tools/pmc_dump.py $DB conv_wgrad_kernel >> $RES 2>&1 || true
  rm -rf $D
done
