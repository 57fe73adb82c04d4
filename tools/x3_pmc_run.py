#!/usr/bin/env python3
"""Workload of tools/x3_pmc.sh: a few launches of every split-product kernel form at its 256-channel layer, batch 32."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_lab_amd import ops

b = 32
x = torch.randn(b, 256, 64, 64, device='cuda'); w = torch.randn(256, 256, 3, 3, device='cuda'); gy = torch.randn(b, 256, 64, 64, device='cuda')
g = ops.Geom(b, 256, 64, 64, 256, 3, 1)
xl = torch.randn(b, 256, 32, 32, device='cuda'); wu = torch.randn(128, 256, 3, 3, device='cuda'); gu = ops.Geom(b, 256, 32, 32, 128, 3, 1, up=1)
xp = torch.randn(b, 128, 64, 64, device='cuda'); wp = torch.randn(256, 128, 3, 3, device='cuda'); gp = ops.Geom(b, 128, 64, 64, 256, 3, 1, pool=1)
gyp = torch.randn(b, 256, 32, 32, device='cuda')
for _ in range(8):
    ops.k_conv_wgrad(gyp, xp, gp, 0.05)
    ops.k_conv_fwd(x, w, None, g, 0.05)
    ops.k_conv_dgrad_mask(gy, w, x, g, 0.05, 0.2)
    ops.k_conv_wgrad(gy, x, g, 0.05)
    ops.k_conv_fwd(xl, wu, None, gu, 0.05)
    ops.k_conv_fwd(xp, wp, None, gp, 0.05)
torch.cuda.synchronize()
