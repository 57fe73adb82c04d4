#!/bin/bash
# MFMA-busy and HBM-traffic counters of the rolling-window weight-gradient kernels (separate --pmc passes).
#   tools/wgrad_pmc.sh <tag>   -> gpurun_out/<tag>_wgrad_pmc.txt
set -e
TAG=${1:-pmc}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
RES=$OUT/${TAG}_wgrad_pmc.txt
: > "$RES"
cd /tmp && export TMPDIR=/tmp
for kind in plain pool up; do
  for ctr in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    D=$OUT/${TAG}_pmc_tmp
    rm -rf "$D"; mkdir -p "$D"
    rocprofv3 --pmc $ctr --kernel-trace -d "$D" -- python3 "$ROOT/tools/wgrad_pmc.py" $kind > /dev/null 2>> "$OUT/${TAG}_pmc.err" || true
    DB=$(find "$D" -name '*.db' | head -1)
    echo "## $kind : --pmc $ctr" >> "$RES"
    python3 "$ROOT/tools/pmc_dump.py" "$DB" wgrad_roll >> "$RES" 2>&1 || true
    rm -rf "$D"
  done
done
tail -60 "$RES"
