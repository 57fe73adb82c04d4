#!/usr/bin/env python3
"""Does the relative placement of the input and output tensors matter for the thin-layer conv (HBM channel / bank
aliasing of the read and the write stream)?  Times the north-star conv with the output view shifted by various byte
offsets inside one big allocation."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gan_lab_amd import _lib, ops

B, C, R = 32, 16, 1024
n = B * C * R * R
x = torch.randn(B, C, R, R, device='cuda')
w = torch.randn(C, C, 3, 3, device='cuda')
g = ops.Geom(B, C, R, R, C, 3, 1, 0)
wp = ops._packed(w, _lib.PACK_FWD, 0.05)
pool = torch.empty(n + (64 << 20), device='cuda')
L = _lib.lib()
print('x at %#x, pool at %#x (delta %d MiB)' % (x.data_ptr(), pool.data_ptr(), (pool.data_ptr() - x.data_ptr()) >> 20))
for off in (4096, 0, 256, 0, 128, 64, 16, 1024, 0, 2 << 20, 4 << 20, 0):
    y = pool[off // 4: off // 4 + n].view(B, C, R, R)

    def run():
        ops.check(L.ganlab_conv_fwd_f32(ops._p(x), ops._p(wp), None, ops._p(y), g.ref(), 1.0, 0, 0.2, ops._st()), 'conv')
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(f'output offset {off:>10d} B: {e0.elapsed_time(e1) / 10:.3f} ms')
