import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_lab_amd import ops, _lib
for ci, co, hw in [(64, 64, 256), (128, 128, 128), (256, 256, 64)]:
    g = ops.Geom(32, ci, hw, hw, co, 3, 1)
    print(ci, co, hw, 'x3_ok fwd', ops.x3_ok(g), 'dgrad', ops.x3_ok(g, True), 'bf', g.bf, 's2', g.s2)
    gy = torch.randn(*g.out_shape, device='cuda'); w = torch.randn(co, ci, 3, 3, device='cuda'); x = torch.randn(*g.in_shape, device='cuda')
    ops.k_conv_dgrad(gy, w, g, 0.1); print('  dgrad ->', _lib.last_launch()[0][:80])
    ops.k_conv_dgrad_mask(gy, w, x, g, 0.1, 0.2); print('  dgrad_mask ->', _lib.last_launch()[0][:80])
    print('  mask ok', ops.conv_dgrad_mask_ok(g))
