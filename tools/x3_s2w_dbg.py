#!/usr/bin/env python3
"""Which weights of the stride-2 split-product weight gradient differ from float64? (debug aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from gan_lab_amd import ops, _lib

n, ci, co, hl, wl, kind = [int(v) if v.isdigit() else v for v in (sys.argv[1:7] if len(sys.argv) > 6 else '2 32 192 32 96 pool'.split())]
up = kind == 'up'
g = torch.Generator().manual_seed(71 + ci + n + hl)
hi, wi = (hl, wl) if up else (2 * hl, 2 * wl)
x = torch.randn(n, ci, hi, wi, generator=g)
gy = torch.randn(n, co, 2 * hl, 2 * wl, generator=g) if up else torch.randn(n, co, hl, wl, generator=g)
geom = ops.Geom(n, ci, hi, wi, co, 3, 1, up=1) if up else ops.Geom(n, ci, hi, wi, co, 3, 1, pool=1)
gw3 = ops.k_conv_wgrad(gy.cuda(), x.cuda(), geom, 1.0).cpu().double()
wd = torch.zeros(co, ci, 3, 3, dtype=torch.float64, requires_grad=True)
yd = F.conv2d(F.interpolate(x.double(), scale_factor=2, mode='nearest'), wd, padding=1) if up else F.avg_pool2d(F.conv2d(x.double(), wd, padding=1), 2)
gwd, = torch.autograd.grad(yd, wd, gy.double())
bad = ((gw3 - gwd).abs() > 1e-3 * gwd.abs().max()).nonzero()
print('mismatches', bad.shape[0], 'of', gwd.numel())
for row in bad[:40].tolist():
    c, i, ky, kx = row
    print(row, float(gw3[c, i, ky, kx]), float(gwd[c, i, ky, kx]), 'diff', float(gw3[c, i, ky, kx] - gwd[c, i, ky, kx]))
