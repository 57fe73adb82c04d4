#!/usr/bin/env python3
"""Where the split-product kernel's output differs from float64: per 16-channel block, per tile row, per 4-pixel group."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from gan_lab_amd import ops
from x3_bench import pack_x3, fwd_x3

torch.manual_seed(0)
ci, co, hw, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
x = torch.randn(n, ci, hw, hw)
w = torch.randn(co, ci, 3, 3) / (3 * ci ** 0.5)
g = ops.Geom(n, ci, hw, hw, co, 3, 1)
exact = F.conv2d(x.double(), w.double(), None, padding=1)
y = fwd_x3(x.cuda(), pack_x3(w.cuda(), 0, 1.0), None, g).double().cpu()
e = (y - exact).abs()
print('rms rel', ((y - exact).pow(2).mean().sqrt() / exact.pow(2).mean().sqrt()).item(), 'max', e.max().item())
print('per image', [f'{v:.1e}' for v in e.amax(dim=(1, 2, 3)).tolist()])
print('per 16-channel block', [f'{v:.1e}' for v in e.view(n, co // 16, 16, hw, hw).amax(dim=(0, 2, 3, 4)).tolist()])
print('per channel%16', [f'{v:.1e}' for v in e.view(n, co // 16, 16, hw, hw).amax(dim=(0, 1, 3, 4)).tolist()])
print('per row%16', [f'{v:.1e}' for v in e.view(n, co, hw // 16, 16, hw).amax(dim=(0, 1, 2, 4)).tolist()])
print('per col%16', [f'{v:.1e}' for v in e.view(n, co, hw, hw // 16, 16).amax(dim=(0, 1, 2, 3)).tolist()])
# which input channels matter: zero out all but one 16-channel half and look at the error
for h in range(ci // 16):
    xz = torch.zeros_like(x)
    xz[:, 16 * h:16 * h + 16] = x[:, 16 * h:16 * h + 16]
    ex = F.conv2d(xz.double(), w.double(), None, padding=1)
    yy = fwd_x3(xz.cuda(), pack_x3(w.cuda(), 0, 1.0), None, g).double().cpu()
    print(f'only half {h}: rms rel {((yy - ex).pow(2).mean().sqrt() / ex.pow(2).mean().sqrt()).item():.2e}')
# which taps: zero all weights but one tap
for t in range(9):
    wz = torch.zeros_like(w)
    wz.view(co, ci, 9)[:, :, t] = w.view(co, ci, 9)[:, :, t]
    ex = F.conv2d(x.double(), wz.double(), None, padding=1)
    yy = fwd_x3(x.cuda(), pack_x3(wz.cuda(), 0, 1.0), None, g).double().cpu()
    print(f'only tap {t}: rms rel {((yy - ex).pow(2).mean().sqrt() / ex.pow(2).mean().sqrt()).item():.2e}')
