#!/bin/bash
# A/B of the second accumulator set (csrc/common.h GL_ACC_DUMP): accuracy probes and the headline step with both builds
# of the library in one call on one box.  Build first:
#   make -C gan_lab_amd/csrc && make -C gan_lab_amd/csrc VARIANT=nodump DEFS=-DGL_ACC_DUMP=0
# usage: tools/acc_dump_ab.sh <out-prefix> [steps]
set -e
out=${1:-gpurun_out/acc_dump_ab}; steps=${2:-6}
mkdir -p "$(dirname "$out")"
for lib in libganlab_hip.so libganlab_hip_nodump.so; do
  tag=${lib#libganlab_hip}; tag=${tag%.so}; tag=${tag:-_dump}
  GANLAB_HIP_LIB=$lib python tools/op_error_probe.py > ${out}_op_error${tag}.txt 2>&1
  GANLAB_HIP_LIB=$lib python tools/dgrad_chain_probe.py 128 > ${out}_dgrad_chain${tag}.txt 2>&1
done
for rep in 1 2; do
  for lib in libganlab_hip.so libganlab_hip_nodump.so; do
    tag=${lib#libganlab_hip}; tag=${tag%.so}; tag=${tag:-_dump}
    GANLAB_HIP_LIB=$lib python bench.py --steps $steps --warmup 2 --no-cpu-baseline > ${out}_bench${tag}_$rep.json 2> ${out}_bench${tag}_$rep.err
  done
done
GANLAB_HIP_LIB=libganlab_hip.so python tools/step_layers.py > ${out}_step_layers_dump.txt 2>&1
GANLAB_HIP_LIB=libganlab_hip_nodump.so python tools/step_layers.py > ${out}_step_layers_nodump.txt 2>&1
