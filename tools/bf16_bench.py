#!/usr/bin/env python3
"""bf16-compute 3x3 convolutions of BASELINE config #2 (StyleGAN 128^2, batch 8): forward, input gradient and weight
gradient per layer shape - time per launch and TFLOP/s against the 2.5 PFLOP/s dense bf16 peak, plus an exactness check
against float64 on the same bf16-rounded operands at a small size.
    python tools/bf16_bench.py [--reps 20] [--check]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from gan_lab_amd import _lib, ops  # noqa: E402

PEAK = 2500.0
# (N, Cin, H, W, Cout): the 3x3 layers of the 128^2 StyleGAN at batch 8 that run in bf16 (inputs already upsampled)
SHAPES = [(8, 512, 16, 16, 512), (8, 512, 32, 32, 512), (8, 512, 64, 64, 256), (8, 256, 64, 64, 256),
          (8, 256, 64, 64, 512), (8, 256, 128, 128, 128), (8, 128, 128, 128, 128), (8, 128, 128, 128, 256)]


def bf(x):
    return x.to(torch.bfloat16).to(torch.float64)


def timeit(fn, reps):
    fn()
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def check():
    worst = 0.0
    for n, ci, h, w, co in [(2, 64, 8, 32, 64), (2, 128, 16, 64, 64), (1, 64, 32, 32, 192), (3, 192, 24, 96, 128),
                            (2, 64, 16, 16, 128), (3, 128, 8, 8, 64), (2, 64, 4, 4, 64)]:
        gen = torch.Generator().manual_seed(n * 1000 + ci + h)
        x = torch.randn(n, ci, h, w, generator=gen)
        wt = torch.randn(co, ci, 3, 3, generator=gen)
        gy = torch.randn(n, co, h, w, generator=gen)
        with ops.compute_dtype('bf16'):
            g = ops.Geom(n, ci, h, w, co, 3, 1)
        if g.bf is None:
            print(f'N{n} {ci}->{co} {h}x{w}: not a bf16 shape', flush=True)
            continue
        y = ops.k_conv_fwd(x.cuda(), wt.cuda(), None, g, 0.05).double().cpu()
        gx = ops.k_conv_dgrad(gy.cuda(), wt.cuda(), g, 0.05).double().cpu()
        gw = ops.k_conv_wgrad(gy.cuda(), x.cuda(), g, 0.05).double().cpu()
        y_ref = F.conv2d(bf(x), bf(wt * 0.05), padding=1)
        gx_ref = F.conv_transpose2d(bf(gy), bf(wt * 0.05), padding=1)
        if w % 32 == 0:
            gw_ref = torch.nn.grad.conv2d_weight(bf(x), wt.shape, bf(gy), padding=1) * 0.05
        else:       # 16-wide maps: exact fp32 weight gradient
            gw_ref = torch.nn.grad.conv2d_weight(x.double(), wt.shape, gy.double(), padding=1) * 0.05
        errs = [((a - b).abs().max() / b.abs().max()).item() for a, b in ((y, y_ref), (gx, gx_ref), (gw, gw_ref))]
        print(f'N{n} {ci}->{co} {h}x{w}: fwd {errs[0]:.1e} dgrad {errs[1]:.1e} wgrad {errs[2]:.1e}', flush=True)
        worst = max(worst, *errs)
    print(f'worst relative error vs float64 on bf16-rounded operands: {worst:.2e}', flush=True)
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--check', action='store_true')
    ap.add_argument('--only', default='')
    ap.add_argument('--shape', type=int, default=-1, help='index into SHAPES (default: all)')
    a = ap.parse_args()
    torch.manual_seed(0)
    if a.check:
        if check() > 5e-5:
            sys.exit('bf16 kernels disagree with the float64 reference')
    print('shape                        pass    kernel                               grid     us   TFLOP/s  of 2.5PF',
          flush=True)
    for n, ci, h, w, co in (SHAPES if a.shape < 0 else SHAPES[a.shape:a.shape + 1]):
        with ops.compute_dtype('bf16'):
            g = ops.Geom(n, ci, h, w, co, 3, 1)
        if g.bf is None:
            print(f'N{n} {ci}->{co} {h}x{w}: not a bf16 shape (fp32 kernels)', flush=True)
        x = torch.randn(n, ci, h, w, device='cuda')
        wt = torch.randn(co, ci, 3, 3, device='cuda')
        gy = torch.randn(n, co, h, w, device='cuda')
        flop = 2.0 * 9 * ci * co * h * w * n
        for name, fn in (('fwd', lambda: ops.k_conv_fwd(x, wt, None, g, 0.05)),
                         ('dgrad', lambda: ops.k_conv_dgrad(gy, wt, g, 0.05)),
                         ('wgrad', lambda: ops.k_conv_wgrad(gy, x, g, 0.05))):
            if a.only and name not in a.only.split(','):
                continue
            ms = timeit(fn, a.reps)
            kern, grid = _lib.last_launch()
            kern = (kern or '?').replace('(anonymous namespace)::', '')
            tf = flop / ms / 1e9
            print(f'N{n} {ci:3d}->{co:3d} {h:3d}x{w:<3d}         {name:6s}  {kern[:36]:36s} {grid:5d} {ms * 1e3:7.1f} '
                  f'{tf:8.1f}  {tf / PEAK:6.3f}', flush=True)


if __name__ == '__main__':
    main()
