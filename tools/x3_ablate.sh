#!/bin/bash
# Ablation builds of the split-product kernels (csrc/conv_x3.hip X3_EXP; wrong results, timing only) on a few layers:
#   tools/x3_ablate.sh <outdir>     (build first: make -C gan_lab_amd/csrc VARIANT=exp1 DEFS=-DX3_EXP=1 ... exp2 exp4 exp7)
OUT=${1:-gpurun_out/x3_ablate}
mkdir -p "$OUT"
for v in "" exp1 exp2 exp4 exp7; do
  lib=libganlab_hip${v:+_$v}.so
  echo "== $lib" >> "$OUT/ablate.txt"
  GANLAB_HIP_LIB=$lib timeout -k 10 200 python tools/conv_bench.py --layers plain --kinds fwd 2>&1 | grep -E "TFLOP" | cut -c1-60 >> "$OUT/ablate.txt"
  GANLAB_HIP_LIB=$lib timeout -k 10 200 python tools/conv_bench.py --layers s2 --kinds fwd 2>&1 | grep -E "up @(32|64|128) " | cut -c1-60 >> "$OUT/ablate.txt"
done
cat "$OUT/ablate.txt"
