#!/usr/bin/env python3
"""Experiment: the thin rolling 3x3 conv with the wave = column block layout (csrc/conv_roll_blur.hip, BLUR = false) against
conv.hip's conv_fwd_roll_kernel (wave = row) on the north-star instance; checks equality first.
    python tools/roll_col_bench.py [batch] [res]"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_lab_amd import _lib, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
L = _lib.lib()
fn = L.ganlab_dbg_conv_fwd_roll_col_f32
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.POINTER(_lib.ConvGeom), ctypes.c_float, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
x = torch.randn(B, 16, R, R, device='cuda')
w = torch.randn(16, 16, 3, 3, device='cuda')
b = torch.randn(16, device='cuda')
g = ops.Geom(B, 16, R, R, 16, 3, 1, 0)
wp = ops._packed(w, ops.PACK_FWD, 0.05)
y2 = torch.empty(B, 16, R, R, device='cuda')


def col():
    rc = fn(x.data_ptr(), wp.data_ptr(), b.data_ptr(), y2.data_ptr(), g.ref(), 1.0, ops.ACT_LRELU, 0.2, ops._st())
    assert rc == 0, rc


def row():
    return ops.k_conv_fwd(x, w, b, g, 0.05, 1.0, ops.ACT_LRELU, 0.2)


y1 = row()
col()
torch.cuda.synchronize()
print('max |col - row| / max|row| =', ((y1 - y2).abs().max() / y1.abs().max()).item())
del y1
for name, f in (('wave = row   (conv_fwd_roll_kernel)', row), ('wave = column block', col)):
    for _ in range(8):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    fl = 2.0 * 9 * 16 * 16 * R * R * B
    print(f'{name:40s} {ms:7.3f} ms  {fl / ms / 1e9:6.1f} TFLOP/s  {fl / ms / 1e9 / 157.3:5.3f} of peak')
