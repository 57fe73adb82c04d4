#!/usr/bin/env python3
"""Same-box A/B of two builds of libganlab_hip.so: the boxes of the pool differ by +-1.5 % in step time, so a kernel
change worth 0.5 % can only be judged by alternating both builds on ONE box inside one gpurun call.

    tools/ab_bench.py OLD.so NEW.so [rounds=3] [-- extra bench.py args]

Copies each library over gan_lab_amd/csrc/libganlab_hip.so in turn (the NEW one is left installed), runs
`bench.py --no-cpu-baseline`, prints ms/step and the two in-step kernel times per run and the means at the end.
Build the variants beforehand, e.g.
    hipcc --offload-arch=gfx950 -O3 -fPIC -DRW_LOAD_AT=12 -c conv.hip -o /tmp/c.o && \\
    hipcc --offload-arch=gfx950 -shared -fPIC -o tools/tmp/lib_new.so /tmp/c.o conv_s2.o conv_bf16.o pointwise.o norm.o data.o
(put them under the repo, not /tmp: only the repo travels to the GPU box)."""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'gan_lab_amd', 'csrc', 'libganlab_hip.so')


def main():
    args = sys.argv[1:]
    extra = []
    if '--' in args:
        i = args.index('--')
        args, extra = args[:i], args[i + 1:]
    old, new = args[0], args[1]
    rounds = int(args[2]) if len(args) > 2 else 3
    res = {'old': [], 'new': []}
    for _ in range(rounds):
        for tag, path in (('old', old), ('new', new)):
            shutil.copyfile(path, LIB)
            out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--no-cpu-baseline'] + extra,
                                 capture_output=True, text=True, cwd=ROOT).stdout.strip().splitlines()
            b = json.loads(out[-1])
            row = (b['ms_per_step'], (b.get('roofline') or {}).get('ms_per_launch'),
                   (b.get('roofline_top_kernel_by_time') or {}).get('ms_per_launch'))
            res[tag].append(row)
            print(tag, *row, flush=True)
    for tag in ('old', 'new'):
        cols = list(zip(*res[tag]))
        print(tag, 'mean', *[round(sum(c) / len(c), 4) if c[0] is not None else None for c in cols])


if __name__ == '__main__':
    main()
