#!/bin/bash
# Hardware counters of one bf16 kernel (separate --pmc passes):  tools/bf16_pmc.sh <tag> <pass> <shape index> <kernel substring>
#   -> gpurun_out/<tag>_bf16_pmc.txt
set -e
TAG=${1:-pmc}; PASS=${2:-wgrad}; SHAPE=${3:-6}; KERN=${4:-bf16}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
RES=$OUT/${TAG}_bf16_pmc.txt
: > "$RES"
cd /tmp && export TMPDIR=/tmp
for ctr in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS" \
           "TA_TA_BUSY_sum TA_BUSY_avr" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  D=$OUT/${TAG}_pmc_tmp
  rm -rf "$D"; mkdir -p "$D"
  rocprofv3 --pmc $ctr --kernel-trace -d "$D" -- python3 "$ROOT/tools/bf16_bench.py" --only $PASS --shape $SHAPE --reps 4 > /dev/null 2>> "$OUT/${TAG}_pmc.err" || true
  DB=$(find "$D" -name '*.db' | head -1)
  echo "## --pmc $ctr" >> "$RES"
  if [ -n "$DB" ]; then python3 "$ROOT/tools/pmc_dump.py" "$DB" "$KERN" >> "$RES" 2>&1 || true; else echo "(no database: counter rejected?)" >> "$RES"; fi
  rm -rf "$D"
done
cat "$RES"
