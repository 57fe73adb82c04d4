import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_lab_amd import ops, _lib
L = _lib.lib()
g = torch.Generator().manual_seed(3)
wt = torch.randn(128, 64, 3, 3, generator=g).cuda()
n = L.ganlab_conv_x3_pack(None, None, 128, 64, 0, 0.37, None)
one = torch.zeros(n, dtype=torch.bfloat16, device='cuda')
L.ganlab_conv_x3_pack(wt.data_ptr(), one.data_ptr(), 128, 64, 0, 0.37, None)
pk = one.float().view(2, 18, 3, 4, 64, 8)
total = pk.double().sum(dim=2)
ref0 = (wt[:64, :16, 0, 0] * 0.37).t().reshape(2, 8, 64).permute(0, 2, 1).double()
d = (total[0, 0, 0:2] - ref0).abs()
print('max diff', d.max().item(), 'mismatches', (d > 0).sum().item(), 'of', d.numel())
i = d.flatten().argmax().item()
print('at', i, total[0, 0, 0:2].flatten()[i].item(), ref0.flatten()[i].item(), 'planes', pk[0, 0, :, 0:2].reshape(3, -1)[:, i].tolist())
ref_f = (wt[:64, :16, 0, 0].double() * 0.37).t().reshape(2, 8, 64).permute(0, 2, 1)
print('vs float64 product with python 0.37: max', (total[0, 0, 0:2] - ref_f).abs().max().item())
import numpy as np
s32 = float(np.float32(0.37))
ref_g = (wt[:64, :16, 0, 0].double() * s32).float().double().t().reshape(2, 8, 64).permute(0, 2, 1)
print('vs fl(w * fl32(0.37)) computed in double then rounded: mismatches', ((total[0, 0, 0:2] - ref_g).abs() > 0).sum().item())
