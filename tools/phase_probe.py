#!/usr/bin/env python3
"""Where does a conv_fwd workgroup spend its life?  Builds a debug copy of the kernel library with
-DGL_PHASES (per-workgroup wall-clock stamps: start / tile staged in LDS / MFMA loop done / stores issued),
runs the north-star 16->16 3x3 conv at 1024^2 x 32 and prints the mean phase durations and the number of
workgroups alive at a time.  Debug tool - not part of the product build."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, 'gan_lab_amd', 'csrc')
OUT = os.path.join(ROOT, 'gpurun_out', 'libganlab_phases.so')


def build():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    srcs = [os.path.join(CSRC, f) for f in ('conv.hip', 'conv_s2.hip', 'conv_bf16.hip', 'pointwise.hip', 'norm.hip', 'data.hip')]
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-shared', '-DGL_PHASES',
                           '-o', OUT] + srcs)


def main():
    if not os.path.exists(OUT) or '--build' in sys.argv:
        build()
    if '--build-only' in sys.argv:
        return
    import torch
    from gan_lab_amd import _lib, ops
    _lib._LIB = None
    _lib.SO_PATH = OUT
    L = _lib.lib()
    L.ganlab_dbg_set_phase_buf.argtypes = [ctypes.c_void_p]
    c, res, n = 16, 1024, 32
    if len(sys.argv) > 1 and sys.argv[1].isdigit():
        c = int(sys.argv[1])
        res = {16: 1024, 32: 512, 64: 256, 128: 128, 256: 64, 512: 32}[c]
    x = torch.randn(n, c, res, res, device='cuda')
    w = torch.randn(c, c, 3, 3, device='cuda')
    g = ops.Geom(n, c, res, res, c, 3, 1, 0)
    buf = torch.zeros(1 << 22, dtype=torch.int64, device='cuda')
    assert L.ganlab_dbg_set_phase_buf(buf.data_ptr()) == 0
    for _ in range(3):
        ops.k_conv_fwd(x, w, None, g, 0.05)
    torch.cuda.synchronize()
    b = buf.cpu().numpy().reshape(-1, 8)
    b = b[b[:, 0] > 0]
    t0 = b[:, 0].min()
    ns = 10.0   # wall_clock64 ticks at 100 MHz
    import numpy as np
    d = (b[:, :4] - t0) * ns / 1e3
    print(f'{len(b)} workgroups; kernel span {d[:, 3].max():.1f} us')
    print('mean us: load+stage %.2f | mfma loop %.2f | epilogue %.2f | total %.2f' %
          ((d[:, 1] - d[:, 0]).mean(), (d[:, 2] - d[:, 1]).mean(), (d[:, 3] - d[:, 2]).mean(), (d[:, 3] - d[:, 0]).mean()))
    for q in (10, 50, 90):
        print(f'p{q}: load+stage %.2f  mfma %.2f  epi %.2f' % tuple(np.percentile(d[:, i + 1] - d[:, i], q) for i in range(3)))
    acc = b[:, 4:8] * ns / 1e3
    if acc.sum() > 0:
        tot = acc.sum(axis=1).mean()
        print('strip totals per workgroup (us): wait-at-barrier %.1f | regs->LDS + barrier %.1f | prefetch issue + '
              'MFMA loop %.1f | epilogue %.1f | sum %.1f' % (*acc.mean(axis=0), tot))
    alive = (d[:, 3] - d[:, 0]).sum() / d[:, 3].max()
    print(f'mean workgroups alive: {alive:.0f} (= {alive / 256:.2f} per CU)')


if __name__ == '__main__':
    main()
