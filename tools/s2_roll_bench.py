#!/usr/bin/env python3
"""Thin stride-2 fused layers (16 <-> 32 channels at the top resolution) - forward and input gradient, executed
TFLOP/s (16 low-resolution taps: 2*16*Cin*Cout*Hl*Wl*N).  Run twice, GANLAB_S2_ROLL=0 (tile kernels) and default
(rolling-window kernels, conv_s2_roll.hip): the switch is read once per process.
    python tools/s2_roll_bench.py [batch] [low_res]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_lab_amd import _lib, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
LO = int(sys.argv[2]) if len(sys.argv) > 2 else 512
SHORT = len(sys.argv) > 3          # under rocprofv3 --pmc: a handful of launches


def timeit(fn, warm=12, reps=20):
    if SHORT:
        warm, reps = 2, 4
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print('GANLAB_S2_ROLL =', os.environ.get('GANLAB_S2_ROLL', '(default: on)'))
for name, cin, cout, hin, up, pool in (('pooled conv 16->32', 16, 32, 2 * LO, 0, 1), ('up-conv 32->16', 32, 16, LO, 1, 0)):
    x = torch.randn(B, cin, hin, hin, device='cuda')
    w = torch.randn(cout, cin, 3, 3, device='cuda')
    g = ops.Geom(B, cin, hin, hin, cout, 3, 1, up, pool)
    gy = torch.randn(*g.out_shape, device='cuda')
    fl = 2.0 * 16 * cin * cout * LO * LO * B
    for kind, fn in (('fwd', lambda: ops.k_conv_fwd(x, w, None, g, 0.05)), ('dgrad', lambda: ops.k_conv_dgrad(gy, w, g, 0.05))):
        ms = timeit(fn)
        sym, grid = _lib.last_launch()
        print(f'{name:20s} {kind:6s} {ms:7.3f} ms  {fl / ms / 1e9:6.1f} TFLOP/s  {fl / ms / 1e9 / 157.3:5.3f} of peak   '
              f'{sym.split("(")[0].replace("(anonymous namespace)::", "")[:40]} grid {grid}')
    del x, w, gy
