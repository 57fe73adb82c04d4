#!/usr/bin/env python3
"""Debug experiment: time the north-star conv with parts of the strip kernel compiled out (-DGL_EXP_*)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, 'gan_lab_amd', 'csrc')
import torch
from gan_lab_amd import _lib, ops
srcs = [os.path.join(CSRC, f) for f in ('conv.hip', 'conv_s2.hip', 'pointwise.hip', 'data.hip')]
objs = {}
def build(flags, tag):
    out = os.path.join(ROOT, 'gpurun_out', f'libexp_{tag}.so')
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-shared'] + flags + ['-o', out, srcs[0]] +
                          [s for s in srcs[1:]])
    return out
x = torch.randn(32, 16, 1024, 1024, device='cuda'); w = torch.randn(16, 16, 3, 3, device='cuda')
for tag, flags in [('base', []), ('nomfma', ['-DGL_EXP_NOMFMA']), ('nostore', ['-DGL_EXP_NOSTORE']), ('noload', ['-DGL_EXP_NOLOAD']),
                   ('nomem', ['-DGL_EXP_NOLOAD', '-DGL_EXP_NOSTORE'])]:
    so = build(flags, tag)
    _lib._LIB = None; _lib.SO_PATH = so
    ops._PACK_CACHE.clear()
    g = ops.Geom(32, 16, 1024, 1024, 16, 3, 1, 0)
    for _ in range(2): ops.k_conv_fwd(x, w, None, g, 0.05)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.k_conv_fwd(x, w, None, g, 0.05)
    e1.record(); torch.cuda.synchronize()
    print(f'{tag:8s} {e0.elapsed_time(e1)/5:.3f} ms')
