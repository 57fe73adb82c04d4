#!/usr/bin/env python3
"""Which resource bounds the thin-layer strip kernel?  Builds debug copies of the kernel library with one part of
the per-tile work removed (-DGL_ABL_NOLOAD: no HBM reads of the next patch, -DGL_ABL_NOSTORE: no output stores,
-DGL_ABL_NOMFMA: no LDS operand reads / MFMAs) and times the north-star 16->16 3x3 conv at 1024^2 x 32 with each.
Debug tool - results are wrong by construction, nothing here is part of the product build."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'gan_lab_amd', 'csrc')
SRCS = [os.path.join(CSRC, f) for f in ('conv.hip', 'conv_s2.hip', 'conv_bf16.hip', 'pointwise.hip', 'norm.hip', 'data.hip')]
VARIANTS = {'full': [], 'old horizontal strip kernel': ['-DGL_ABL_OLDSTRIP'], 'noload': ['-DGL_ABL_NOLOAD'], 'nostore': ['-DGL_ABL_NOSTORE'], 'nomfma': ['-DGL_ABL_NOMFMA'],
            'noload+nostore': ['-DGL_ABL_NOLOAD', '-DGL_ABL_NOSTORE'],
            }


def main():
    if len(sys.argv) > 1:           # child: time one variant
        sys.path.insert(0, ROOT)
        import torch
        from gan_lab_amd import _lib, ops
        _lib.SO_PATH = sys.argv[1]
        _lib.lib()
        x = torch.randn(32, 16, 1024, 1024, device='cuda')
        w = torch.randn(16, 16, 3, 3, device='cuda')
        g = ops.Geom(32, 16, 1024, 1024, 16, 3, 1, 0)
        for _ in range(3):
            ops.k_conv_fwd(x, w, None, g, 0.05)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.k_conv_fwd(x, w, None, g, 0.05)
        e1.record()
        torch.cuda.synchronize()
        print(f'{sys.argv[2]:>16}: {e0.elapsed_time(e1) / 10:.3f} ms')
        return
    out = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    for name, flags in VARIANTS.items():
        so = os.path.join(out, f'libganlab_abl_{name.replace("+", "_").replace(" ", "_")}.so')
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-shared', '-o', so] +
                              flags + SRCS, stderr=subprocess.DEVNULL)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), so, name])


if __name__ == '__main__':
    main()
