#!/usr/bin/env python3
"""Launch weight-gradient instances a few times - run under rocprofv3 --pmc (tools/wgrad_pmc.sh).
    tools/wgrad_pmc.py plain|pool|up"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_lab_amd import ops
kind = sys.argv[1] if len(sys.argv) > 1 else 'plain'
n, ci, co, r = {'plain': (32, 16, 16, 1024), 'pool': (32, 16, 32, 1024), 'up': (32, 32, 16, 512)}[kind]
x = torch.randn(n, ci, r, r, device='cuda')
g = ops.Geom(n, ci, r, r, co, 3, 1, int(kind == 'up'), int(kind == 'pool'))
gy = torch.randn(*g.out_shape, device='cuda')
for _ in range(8):
    gw = ops.k_conv_wgrad(gy, x, g, 1.0)
torch.cuda.synchronize()
print('done', kind, float(gw.flatten()[0]))
