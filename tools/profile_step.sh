#!/bin/bash
# rocprofv3 kernel trace of the default bench command (2 timed steps + 1 warm-up + the FLOP-counting step: 4 steps) -> per-kernel summary CSV.
#   tools/profile_step.sh <tag>      writes gpurun_out/<tag>_step_kernel_stats.csv (+ .txt head) on the GPU box
set -e
TAG=${1:-prof}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT/$TAG"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/$TAG" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline \
    > "$OUT/${TAG}_bench_under_rocprof.json" 2> "$OUT/${TAG}_rocprof.err" || true
DB=$(find "$OUT/$TAG" -name '*.db' | head -1)
python3 "$ROOT/tools/rocpd_stats.py" "$DB" 4 "$OUT/${TAG}_step_kernel_stats.csv" > "$OUT/${TAG}_step_kernel_stats.txt"
rm -rf "$OUT/$TAG"
head -45 "$OUT/${TAG}_step_kernel_stats.txt"
