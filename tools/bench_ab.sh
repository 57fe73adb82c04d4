#!/bin/bash
# The headline step with every build of the library under csrc/ (make VARIANT=...), interleaved, on one box.
#   tools/bench_ab.sh <out-prefix> [steps] [reps]
out=$1; steps=${2:-6}; reps=${3:-2}
mkdir -p "$(dirname "$out")"
for rep in $(seq 1 $reps); do
  for lib in gan_lab_amd/csrc/libganlab_hip*.so; do
    b=$(basename $lib); tag=${b#libganlab_hip}; tag=${tag%.so}; tag=${tag:-_default}
    GANLAB_HIP_LIB=$b python bench.py --steps $steps --warmup 2 --no-cpu-baseline > ${out}${tag}_$rep.json 2> ${out}${tag}_$rep.err
  done
done
