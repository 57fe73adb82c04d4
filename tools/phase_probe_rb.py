#!/usr/bin/env python3
"""Where a step of the column-block rolling kernels (csrc/conv_roll_blur.hip: plain / + blur / fromRGB fold) goes, measured
with every workgroup's neighbours present: accumulated wall-clock time of the phases of wave 0, debug build of the library
    make -C gan_lab_amd/csrc VARIANT=phases DEFS=-DGL_PHASES
    GANLAB_HIP_LIB=libganlab_hip_phases.so python tools/phase_probe_rb.py"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gan_lab_amd import _lib, ops

L = _lib.lib()
L.ganlab_dbg_set_phase_buf_rb.argtypes = [ctypes.c_void_p]
n, c, res = 32, 16, 1024
x = torch.randn(n, c, res, res, device='cuda')
w = torch.randn(c, c, 3, 3, device='cuda')
b = torch.randn(c, device='cuda')
g = ops.Geom(n, c, res, res, c, 3, 1, 0)
buf = torch.zeros(12 << 16, dtype=torch.int64, device='cuda')
assert L.ganlab_dbg_set_phase_buf_rb(buf.data_ptr()) == 0
NAMES = ['prologue', 'MFMA phase', 'activation (+ 4 row stores: plain) + publish', 'wait at barrier 1',
         'plain: prefetched rows -> ring', 'wait at barrier 2', 'blur: prefetched rows -> ring', 'blur: exchange reads + horizontal',
         'blur: vertical + 4 row stores', 'blur: sign bits']
for name, fn in (('plain (RB_PLAIN)', lambda: ops.k_conv_fwd(x, w, b, g, 0.05, 1.0, ops.ACT_LRELU, 0.2)),
                 ('conv + LeakyReLU + blur (RB_BLUR)', lambda: ops.k_conv_fwd_blur_bits(x, w, b, g, 0.05, 1.0, 0.2))):
    for _ in range(4):
        buf.zero_()
        fn()
    torch.cuda.synchronize()
    sym, grid = _lib.last_launch()
    d = buf.cpu().numpy().reshape(-1, 12)[:grid]
    steps = d[:, 10].astype(np.float64)
    ph = d[:, :10].astype(np.float64) * 10.0      # ns (wall_clock64 ticks at 100 MHz)
    per_step = ph[:, 1:].sum(axis=1) / steps
    print(f'{name}: {sym.split("(")[0][-44:]}, {grid} workgroups x {steps.mean():.1f} steps; per step {per_step.mean() / 1e3:.2f} us')
    for i, nm in enumerate(NAMES):
        v = ph[:, i] / (steps if i else 1)
        print(f'   {nm:38s} {v.mean() / 1e3:7.3f} us {"per step" if i else "per workgroup"}   ({100 * v.mean() / per_step.mean():5.1f} % of a step)' if i else
              f'   {nm:38s} {v.mean() / 1e3:7.3f} us per workgroup')
