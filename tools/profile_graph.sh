#!/bin/bash
# rocprofv3 kernel trace of a launch-bound configuration replayed as HIP graphs:  tools/profile_graph.sh <tag> <config> <steps>
set -e
TAG=${1:-gprof}; CFG=${2:-2}; STEPS=${3:-40}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT/$TAG"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$OUT/$TAG" -- python3 "$ROOT/bench.py" --config $CFG --steps $STEPS --warmup 6 --no-cpu-baseline --step-graph 1 \
    > "$OUT/${TAG}_bench_under_rocprof.json" 2> "$OUT/${TAG}_rocprof.err" || true
DB=$(find "$OUT/$TAG" -name '*.db' | head -1)
python3 "$ROOT/tools/graph_gaps.py" "$DB" $STEPS "$OUT/${TAG}_graph_kernels.csv" > "$OUT/${TAG}_graph_kernels.txt"
rm -rf "$OUT/$TAG"
head -60 "$OUT/${TAG}_graph_kernels.txt"
