#!/bin/bash
# End-of-round records on ONE box (copied to profiles/ afterwards): both bench lines, per-layer step profile, kernel stats,
# the per-op error probe, the split-product kernels' benches.
#   tools/round_records.sh <tag>
TAG=${1:-rNN}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_x3.json 2> $OUT/bench_x3.err
GANLAB_X3=0 timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_exact_fp32.json 2> $OUT/bench_exact.err
timeout -k 10 300 python tools/step_layers.py > $OUT/step_layers.txt 2>&1
timeout -k 10 300 python tools/op_error_probe.py > $OUT/op_error_probe.txt 2>&1
timeout -k 10 300 python tools/x3_bench.py --forms > $OUT/x3_bench.txt 2>&1
timeout -k 10 300 python tools/x3_bench.py --wgrad > $OUT/x3_wgrad.txt 2>&1
tail -1 $OUT/bench_x3.json | cut -c1-200; tail -1 $OUT/bench_exact_fp32.json | cut -c1-200
