#!/usr/bin/env python3
"""Where does a multi-tile strip differ from the CPU conv?  Prints the max error per 32-pixel tile column and per
tile row for a thin layer big enough to engage strips of several tiles (debug tool)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from gan_lab_amd import ops

n, c, r = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (12, 16, 512)
g = torch.Generator().manual_seed(0)
x = torch.randn(n, c, r, r, generator=g)
w = torch.randn(c, c, 3, 3, generator=g)
y = ops.conv2d(x.cuda(), w.cuda(), None, scale=0.1, padding=1).cpu()
ref = F.conv2d(x * 0.1, w, padding=1)
err = (y - ref).abs()
print('max err', err.max().item(), 'ref max', ref.abs().max().item())
e_col = err.amax(dim=(0, 1, 2)).view(-1, 32).amax(dim=1)
print('per 32-px tile column:', [f'{v:.1e}' for v in e_col.tolist()])
e_row = err.amax(dim=(0, 1, 3)).view(-1, 8).amax(dim=1)
print('per 8-px tile row (first 16):', [f'{v:.1e}' for v in e_row.tolist()[:16]])
e_n = err.amax(dim=(1, 2, 3))
print('per image:', [f'{v:.1e}' for v in e_n.tolist()])
e_c = err.amax(dim=(0, 2, 3))
print('per channel:', [f'{v:.1e}' for v in e_c.tolist()])
bad = (err > 1e-3).nonzero()
print('first bad:', bad[:5].tolist(), 'count', len(bad))
if len(bad):
    i = bad[0]
    print('col within tile of bad px:', sorted(set((bad[:2000, 3] % 32).tolist())))
    print('row within tile of bad px:', sorted(set((bad[:2000, 2] % 8).tolist())))
