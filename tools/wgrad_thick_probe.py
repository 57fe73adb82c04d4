import os, sys
sys.path.insert(0, '/root/repo')
import torch
from gan_lab_amd import ops
n, c, r = 32, 256, 64
x = torch.randn(n, c, r, r, device='cuda'); gy = torch.randn(n, c, r, r, device='cuda')
g = ops.Geom(n, c, r, r, c, 3, 1)
for _ in range(6):
    gw = ops.k_conv_wgrad(gy, x, g, 1.0)
torch.cuda.synchronize()
print(float(gw.flatten()[0]))
