#!/usr/bin/env python3
"""The 32-channel form of the transposed split-product kernel at the step's 64 <-> 32 layers: ms per launch against the exact kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_lab_amd import ops, _lib


def timeit(fn, rounds=5, reps=5):
    for _ in range(3):
        fn()
    out = []
    for _ in range(rounds):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / reps)
    return sorted(out)[len(out) // 2]


def sw(fn, on):
    def f():
        prev = ops.set_x3(on)
        try:
            return fn()
        finally:
            ops.set_x3(prev)
    return f


b = 32
x = torch.randn(b, 64, 256, 256, device='cuda'); w = torch.randn(32, 64, 3, 3, device='cuda')
g = ops.Geom(b, 64, 256, 256, 32, 3, 1, up=1)
s_, t_ = torch.rand(b, 64, device='cuda') + 0.5, torch.randn(b, 64, device='cuda')
fwd = lambda: ops.k_conv_fwd(x, w, None, g, 0.05)
aff = lambda: ops.k_conv_fwd_aff(x, s_, t_, w, g, 0.05)
for name, fn in (('up 64->32 @256->512 fwd', fwd), ('up 64->32 fwd, affine on load', aff)):
    t1, t3 = timeit(sw(fn, False)), timeit(sw(fn, True))
    print(f'{name}: exact {t1:.3f} ms  3xbf16 {t3:.3f} ms  {t1 / t3:.2f}x  [{_lib.last_launch()[0][:60]}]', flush=True)
del x
gy = torch.randn(b, 64, 256, 256, device='cuda'); wp = torch.randn(64, 32, 3, 3, device='cuda')
gp = ops.Geom(b, 32, 512, 512, 64, 3, 1, pool=1)
dg = lambda: ops.k_conv_dgrad(gy, wp, gp, 0.05)
t1, t3 = timeit(sw(dg, False)), timeit(sw(dg, True))
print(f'pool 32->64 @512->256 input gradient: exact {t1:.3f} ms  3xbf16 {t3:.3f} ms  {t1 / t3:.2f}x  [{_lib.last_launch()[0][:60]}]', flush=True)
