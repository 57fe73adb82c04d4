#!/bin/bash
# rocprofv3 kernel trace of `bench.py --config C` -> per-kernel summary CSV:  tools/profile_config.sh <tag> <C> <steps>
set -e
TAG=${1:-prof}; CFG=${2:-2}; STEPS=${3:-20}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT/$TAG"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/$TAG" -- python3 "$ROOT/bench.py" --config $CFG --steps $STEPS --warmup 1 --no-cpu-baseline \
    > "$OUT/${TAG}_bench_under_rocprof.json" 2> "$OUT/${TAG}_rocprof.err" || true
DB=$(find "$OUT/$TAG" -name '*.db' | head -1)
python3 "$ROOT/tools/rocpd_stats.py" "$DB" $((STEPS + 2)) "$OUT/${TAG}_kernel_stats.csv" > "$OUT/${TAG}_kernel_stats.txt"
rm -rf "$OUT/$TAG"
head -40 "$OUT/${TAG}_kernel_stats.txt"
