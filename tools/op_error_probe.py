#!/usr/bin/env python3
"""Per-op rounding error of the conv kernels against float64, next to the ATen CPU fp32 conv's."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from gan_lab_amd import ops, _lib


def err(a, ref):
    a, ref = a.double().cpu(), ref.double()
    return ((a - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item(), ((a - ref).abs().max() / ref.abs().max()).item()


def main():
    torch.manual_seed(0)
    torch.set_num_threads(16)
    ops._X3_MIN_TILES = 1          # the split-product kernels on these small batches too (the step uses them at batch 32)
    cases = [('fwd', 2, 64, 64, 128, 128, 0, 0), ('fwd', 2, 256, 256, 32, 32, 0, 0), ('dgrad', 2, 128, 128, 64, 64, 0, 0),
             ('dgrad', 2, 512, 512, 16, 16, 0, 0), ('fwd', 2, 128, 64, 32, 32, 1, 0), ('dgrad', 2, 64, 128, 64, 64, 0, 1),
             ('fwd', 2, 64, 128, 64, 64, 0, 1), ('dgrad', 2, 256, 128, 16, 16, 1, 0),
             ('fwd', 2, 16, 16, 256, 256, 0, 0), ('fwd', 2, 16, 16, 256, 224, 0, 0), ('fwd', 2, 32, 32, 128, 128, 0, 0),
             ('fwd', 2, 64, 64, 64, 64, 0, 0), ('fwd', 2, 512, 512, 16, 16, 0, 0),
             ('fwd', 2, 32, 16, 128, 128, 1, 0), ('fwd', 2, 16, 32, 256, 256, 0, 1), ('fwd', 2, 64, 32, 64, 64, 1, 0),
             ('dgrad', 2, 16, 16, 256, 256, 0, 0), ('dgrad', 2, 32, 16, 128, 128, 1, 0), ('dgrad', 2, 16, 32, 256, 256, 0, 1),
             ('wgrad', 4, 64, 64, 128, 128, 0, 0), ('wgrad', 4, 256, 256, 32, 32, 0, 0), ('wgrad', 2, 64, 128, 128, 128, 0, 1),
             ('wgrad', 2, 128, 64, 64, 64, 1, 0), ('wgrad', 2, 256, 512, 64, 64, 0, 1), ('wgrad', 2, 16, 32, 256, 256, 0, 1),
             ('wgrad', 2, 32, 32, 128, 128, 0, 0)]
    for kind, n, ci, co, h, w, up, pool in cases:
        x = torch.randn(n, ci, h, w)
        wt = torch.randn(co, ci, 3, 3) / (3 * ci ** 0.5)
        g = ops.Geom(n, ci, h, w, co, 3, 1, up, pool)

        def ref_fwd(xx, ww):
            xi = F.interpolate(xx, scale_factor=2, mode='nearest') if up else xx
            y = F.conv2d(xi, ww, padding=1)
            return F.avg_pool2d(y, 2) if pool else y
        prev = ops.set_x3(False)          # first the exact-fp32 MFMA kernels ...
        try:
            hx = ops.k_conv_fwd(x.cuda(), wt.cuda(), None, g, 1.0) if kind == 'fwd' else None
        finally:
            ops.set_x3(prev)
        if kind == 'wgrad':
            gy = torch.randn(*g.out_shape)
            prev = ops.set_x3(False)
            try:
                hx = ops.k_conv_wgrad(gy.cuda(), x.cuda(), g, 1.0)
            finally:
                ops.set_x3(prev)
            c0 = _lib.launch_count()
            hip = ops.k_conv_wgrad(gy.cuda(), x.cuda(), g, 1.0)
            names = [n_ for n_, _ in _lib.launches_since(c0)]
            name = next((n_ for n_ in names if 'conv_x3' in n_), names[0] if names else '')
            wd = wt.double().requires_grad_(True)
            exact, = torch.autograd.grad(ref_fwd(x.double(), wd), wd, gy.double())
            wf = wt.clone().requires_grad_(True)
            cpu, = torch.autograd.grad(ref_fwd(x, wf), wf, gy)
        elif kind == 'fwd':
            hip = ops.k_conv_fwd(x.cuda(), wt.cuda(), None, g, 1.0)
            name, _ = _lib.last_launch()
            exact = ref_fwd(x.double(), wt.double())
            cpu = ref_fwd(x, wt)
        else:
            gy = torch.randn(*g.out_shape)
            prev = ops.set_x3(False)
            try:
                hx = ops.k_conv_dgrad(gy.cuda(), wt.cuda(), g, 1.0)
            finally:
                ops.set_x3(prev)
            hip = ops.k_conv_dgrad(gy.cuda(), wt.cuda(), g, 1.0)
            name, _ = _lib.last_launch()
            xd = x.double().requires_grad_(True)
            exact, = torch.autograd.grad(ref_fwd(xd, wt.double()), xd, gy.double())
            xf = x.clone().requires_grad_(True)
            cpu, = torch.autograd.grad(ref_fwd(xf, wt), xf, gy)
        eh, ec, ex = err(hip, exact), err(cpu, exact), err(hx, exact)
        x3 = 'conv_x3' in name or 'x3w_reduce' in name or 'x3sw_reduce' in name
        print(f'{kind:5s} {ci:3d}->{co:3d} {h}x{w} up{up} pool{pool}: HIP rms {eh[0]:.2e} max {eh[1]:.2e} | CPU fp32 rms {ec[0]:.2e} max {ec[1]:.2e} '
              f'| ratio rms {eh[0] / ec[0]:.2f}' + (f' (3xbf16; exact-fp32 kernel {ex[0] / ec[0]:.2f})' if x3 else '') +
              f'  [{name.split("(")[0][-48:]}]', flush=True)


if __name__ == '__main__':
    main()
