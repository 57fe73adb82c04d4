#!/usr/bin/env python3
"""Forward / input-gradient / weight-gradient kernels of the headline network's layer shapes in isolated loops, batch 32:
executed TFLOP/s and the dispatched symbol.  For A/B builds of the library run it once per build
(GANLAB_HIP_LIB=libganlab_hip_<variant>.so) in the same gpurun call; rounds are interleaved inside one process only for
the layers of ONE build, so compare runs on the same box.
    python tools/conv_bench.py [--layers thick|thin|all] [--kinds fwd,dgrad,wgrad] [--batch 32]"""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_lab_amd import _lib, ops

# (name, cin, cout, h_in, up, pool)
THICK = [('32->32 @512', 32, 32, 512, 0, 0), ('64->64 @256', 64, 64, 256, 0, 0), ('128->128 @128', 128, 128, 128, 0, 0),
         ('256->256 @64', 256, 256, 64, 0, 0), ('512->512 @32', 512, 512, 32, 0, 0), ('512->512 @16', 512, 512, 16, 0, 0),
         ('32->64 pool @512', 32, 64, 512, 0, 1), ('64->128 pool @256', 64, 128, 256, 0, 1),
         ('128->256 pool @128', 128, 256, 128, 0, 1), ('256->512 pool @64', 256, 512, 64, 0, 1),
         ('512->512 pool @32', 512, 512, 32, 0, 1),
         ('64->32 up @256', 64, 32, 256, 1, 0), ('128->64 up @128', 128, 64, 128, 1, 0), ('256->128 up @64', 256, 128, 64, 1, 0),
         ('512->256 up @32', 512, 256, 32, 1, 0), ('512->512 up @16', 512, 512, 16, 1, 0)]
SMALL = [('512->512 @16', 512, 512, 16, 0, 0), ('512->512 @8', 512, 512, 8, 0, 0), ('512->512 @4', 512, 512, 4, 0, 0),
         ('16->16 @512 (w%64)', 16, 16, 512, 0, 0), ('16->16 @96', 16, 16, 96, 0, 0)]
THIN = [('16->16 @1024', 16, 16, 1024, 0, 0), ('16->32 pool @1024', 16, 32, 1024, 0, 1), ('32->16 up @512', 32, 16, 512, 1, 0)]


def timeit(fn, warm, reps):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--layers', default='thick')
    p.add_argument('--kinds', default='fwd,dgrad')
    p.add_argument('--batch', type=int, default=32)
    p.add_argument('--reps', type=int, default=10)
    a = p.parse_args()
    layers = {'thick': THICK, 'thin': THIN, 'all': THICK + THIN, 'plain': THICK[:5], 's2': THICK[6:], 'small': SMALL}[a.layers]
    print('library:', os.path.basename(_lib.SO_PATH))
    tot = {}
    for name, cin, cout, hin, up, pool in layers:
        x = torch.randn(a.batch, cin, hin, hin, device='cuda')
        w = torch.randn(cout, cin, 3, 3, device='cuda')
        g = ops.Geom(a.batch, cin, hin, hin, cout, 3, 1, up, pool)
        gy = torch.randn(*g.out_shape, device='cuda')
        fl = ops.conv_flops(g)
        fns = {'fwd': lambda: ops.k_conv_fwd(x, w, None, g, 0.05), 'dgrad': lambda: ops.k_conv_dgrad(gy, w, g, 0.05),
               'wgrad': lambda: ops.k_conv_wgrad(gy, x, g, 0.05)}
        for kind in a.kinds.split(','):
            ms = timeit(fns[kind], 6, a.reps)
            sym, grid = _lib.last_launch()
            tot[kind] = tot.get(kind, 0.) + ms
            print(f'{name:20s} {kind:6s} {ms:7.3f} ms  {fl / ms / 1e9:6.1f} TFLOP/s  {fl / ms / 1e9 / 157.3:5.3f}   '
                  f'{sym.split("(")[0].replace("(anonymous namespace)::", "").replace("void ", "")[:60]} grid {grid}', flush=True)
        del x, w, gy
    print('sum of ms per kind:', {k: round(v, 3) for k, v in tot.items()})


if __name__ == '__main__':
    main()
