#!/usr/bin/env python3
"""Debug tool: forward / dgrad / wgrad of a list of conv geometries against torch on the CPU (max relative error)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from gan_lab_amd import ops

CASES = [  # N, Cin, H, W, Cout, ks, pad, up
    (8, 512, 8, 8, 512, 3, 1, 0), (8, 512, 8, 8, 512, 3, 1, 1), (8, 512, 16, 16, 512, 3, 1, 0),
    (8, 512, 8, 8, 512, 1, 0, 0), (8, 512, 8, 8, 512, 1, 0, 1), (8, 512, 4, 4, 512, 3, 1, 1),
    (8, 512, 16, 16, 256, 3, 1, 1), (8, 256, 32, 32, 128, 3, 1, 1), (8, 512, 4, 4, 512, 1, 0, 1),
    (8, 3, 64, 64, 64, 3, 1, 0), (8, 64, 64, 64, 3, 3, 1, 0),
]
if len(sys.argv) > 1:
    CASES = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]


def rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


for c in CASES:
    n, cin, h, w, cout, ks, pad, up = c
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, ks, ks, generator=g)
    s = 1.0 / (cin * ks * ks) ** 0.5
    xr = x.clone().requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    xi = F.interpolate(xr, scale_factor=2, mode='nearest') if up else xr
    yr = F.conv2d(xi * s, wr, padding=pad)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xg = x.clone().cuda().requires_grad_(True)
    wg = wt.clone().cuda().requires_grad_(True)
    y = ops.conv2d(xg, wg, None, scale=s, padding=pad, up=bool(up))
    y.backward(gy.cuda())
    print(c, 'fwd %.1e dgrad %.1e wgrad %.1e' % (rel(y.detach().cpu(), yr.detach()), rel(xg.grad.cpu(), xr.grad),
                                                 rel(wg.grad.cpu(), wr.grad)), flush=True)
