#!/usr/bin/env python3
"""The thick fp32 weight-gradient kernel (conv_wgrad_kernel, 64 co x 32 ci tiles) on the layer shapes of the three fp32
configurations: ms per launch and TFLOP/s; GANLAB_WGRAD_XCD=0/1 selects the block order (A/B).  With an argument `pmc` it
only launches one shape a few times (tools/wgrad_thick_pmc.sh).
    python tools/wgrad_thick_probe.py [pmc]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_lab_amd import _lib, ops

SHAPES = [(32, 512, 32, 512), (32, 256, 64, 256), (32, 128, 128, 128), (32, 64, 256, 64), (32, 512, 16, 512),
          (64, 512, 8, 512), (64, 256, 16, 512), (64, 128, 32, 128)]


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'pmc':
        n, c, r = 32, 256, 64
        x = torch.randn(n, c, r, r, device='cuda'); gy = torch.randn(n, c, r, r, device='cuda')
        g = ops.Geom(n, c, r, r, c, 3, 1)
        for _ in range(6):
            gw = ops.k_conv_wgrad(gy, x, g, 1.0)
        torch.cuda.synchronize()
        print(float(gw.flatten()[0]))
        return
    for n, c, r, co in SHAPES:
        x = torch.randn(n, c, r, r, device='cuda'); gy = torch.randn(n, co, r, r, device='cuda')
        g = ops.Geom(n, c, r, r, co, 3, 1)
        for _ in range(3):
            ops.k_conv_wgrad(gy, x, g, 1.0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.k_conv_wgrad(gy, x, g, 1.0)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f'{c:3d}->{co:3d} @{r:3d}^2 x{n}: {ms:.3f} ms  {2.0 * 9 * c * co * r * r * n / ms / 1e9:6.1f} TFLOP/s', flush=True)


if __name__ == '__main__':
    main()
