#!/bin/bash
# tools/conv_bench.py once per build of the library, on one box.   tools/conv_ab.sh <out-prefix> <conv_bench args...>
out=$1; shift
mkdir -p "$(dirname "$out")"
for lib in gan_lab_amd/csrc/libganlab_hip*.so; do
  b=$(basename $lib); tag=${b#libganlab_hip}; tag=${tag%.so}; tag=${tag:-_default}
  GANLAB_HIP_LIB=$b python tools/conv_bench.py "$@" > ${out}${tag}.txt 2>&1
done
