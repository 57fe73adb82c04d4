#!/bin/bash
# MFMA-busy, wait and HBM-traffic counters of the thin stride-2 S / T kernels (separate --pmc passes; rolling-window
# kernels by default, GANLAB_S2_ROLL=0 for the tile kernels).   tools/s2_roll_pmc.sh <tag>  -> gpurun_out/<tag>_s2_roll_pmc.txt
set -e
TAG=${1:-s2pmc}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
RES=$OUT/${TAG}_s2_roll_pmc.txt
: > "$RES"
cd /tmp && export TMPDIR=/tmp
for ctr in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  D=$OUT/${TAG}_pmc_tmp
  rm -rf "$D"; mkdir -p "$D"
  rocprofv3 --pmc $ctr --kernel-trace -d "$D" -- python3 "$ROOT/tools/s2_roll_bench.py" 32 512 short > /dev/null 2>> "$OUT/${TAG}_pmc.err" || true
  DB=$(find "$D" -name '*.db' | head -1)
  echo "## --pmc $ctr   (dispatch, kernel, counter, value, ns)" >> "$RES"
  python3 "$ROOT/tools/pmc_dump.py" "$DB" conv_s2_ | awk -F, '{k=$2" "$3; n[k]++; v[k]+=$4; t[k]+=$5} END {for (k in n) printf "%s  launches %d  mean value %.0f  mean ns %.0f\n", k, n[k], v[k]/n[k], t[k]/n[k]}' | sort >> "$RES" 2>&1 || true
  rm -rf "$D"
done
cat "$RES"
