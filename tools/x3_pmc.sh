#!/bin/bash
# Counters of the split-product kernels (separate --pmc passes over tools/x3_pmc_run.py: 8 launches of each form on the
# 256 -> 256 @64^2 x32 layer, the weight gradient and the two stride-2 forms at their 256-channel layers).
#   tools/x3_pmc.sh <tag>  -> gpurun_out/<tag>_x3_pmc.txt
set -e
TAG=${1:-x3pmc}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
RES=$OUT/${TAG}_x3_pmc.txt
: > "$RES"
cd /tmp && export TMPDIR=/tmp
for ctr in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  D=$OUT/${TAG}_pmc_tmp
  rm -rf "$D"; mkdir -p "$D"
  rocprofv3 --pmc $ctr --kernel-trace -d "$D" -- python3 "$ROOT/tools/x3_pmc_run.py" > /dev/null 2>> "$OUT/${TAG}_pmc.err" || true
  DB=$(find "$D" -name '*.db' | head -1)
  echo "## --pmc $ctr" >> "$RES"
  python3 "$ROOT/tools/pmc_dump.py" "$DB" x3 > "$OUT/${TAG}_raw.txt" 2>&1 || true
  python3 - "$OUT/${TAG}_raw.txt" >> "$RES" <<'PY'
import collections, re, sys
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for line in open(sys.argv[1]):
    if line.startswith('#'): continue
    parts = line.rstrip('\n').split(',')
    if len(parts) < 5: continue
    name = re.sub(r'\(.*$', '', ','.join(parts[1:-3]).replace('void ', '').replace('(anonymous namespace)::', ''))
    a = agg[(name, parts[-3])]; a[0] += 1; a[1] += float(parts[-2]); a[2] += float(parts[-1])
for (k, c), (n, v, ns) in sorted(agg.items()):
    print(f'{k:40s} {c:26s} launches {n:3d}  mean {v / n:16.0f}  mean us {ns / n / 1e3:8.1f}')
PY
  rm -rf "$D"
done
cat "$RES"
