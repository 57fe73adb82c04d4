#!/usr/bin/env python3
"""Launch the north-star conv instance (3x3, 16->16, 1024^2, batch 32) a few times - used under
rocprofv3 (--kernel-trace --stats, and separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_lab_amd import ops
B, C, R = 32, 16, 1024
if len(sys.argv) > 2:      # tools/roofline_kernel.py 256 64 -> the thick-layer instance
    C, R = int(sys.argv[1]), int(sys.argv[2])
x = torch.randn(B, C, R, R, device='cuda'); w = torch.randn(C, C, 3, 3, device='cuda')
g = ops.Geom(B, C, R, R, C, 3, 1, 0)
for _ in range(6):
    y = ops.k_conv_fwd(x, w, None, g, 0.05)
torch.cuda.synchronize()
print('done', float(y[0, 0, 0, 0]))
