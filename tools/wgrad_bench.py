#!/usr/bin/env python3
"""Weight-gradient kernels: correctness against a float64 reference and TFLOP/s, rolling-window kernel (wgrad_roll.hip)
vs the tile kernel (GANLAB_WGRAD_ROLL=0), same process, interleaved rounds.
    python tools/wgrad_bench.py [--big]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from gan_lab_amd import _lib, ops  # noqa: E402


def ref_wgrad(gy, x, pad=1, up=False, pool=False):
    """float64 on the CPU (small cases only)."""
    gy, x = gy.double().cpu(), x.double().cpu()
    w = torch.zeros(gy.shape[1], x.shape[1], 3, 3, dtype=torch.float64, requires_grad=True)
    xin = F.interpolate(x, scale_factor=2, mode='nearest') if up else x
    y = F.conv2d(xin, w, padding=pad)
    if pool:
        y = F.avg_pool2d(y, 2)
    y.backward(gy)
    return w.grad


def run(gy, x, g, roll):
    os.environ['GANLAB_WGRAD_ROLL'] = '1' if roll else '0'
    out = ops.k_conv_wgrad(gy, x, g, 1.0)
    name, grid = _lib.last_launch()
    return out


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--big', action='store_true')
    ap.add_argument('--spu', type=int, nargs='*', default=[0])
    a = ap.parse_args()
    torch.manual_seed(0)
    small = [(2, 16, 16, 8, 64), (3, 12, 10, 40, 64), (2, 16, 16, 64, 128), (2, 32, 24, 16, 64), (1, 5, 32, 72, 192),
             (2, 16, 16, 68, 64)]
    worst = 0.0
    for n, ci, co, h, w in small:
        x = torch.randn(n, ci, h, w, device='cuda')
        gy = torch.randn(n, co, h, w, device='cuda')
        g = ops.Geom(n, ci, h, w, co, 3, 1, 0)
        ref = ref_wgrad(gy, x)
        for roll in (True, False):
            out = run(gy, x, g, roll).double().cpu()
            err = ((out - ref).abs().max() / ref.abs().max()).item()
            print(f'{"roll" if roll else "tile"} N{n} {ci}->{co} {h}x{w}: rel err {err:.2e}', flush=True)
            worst = max(worst, err)
    # stride-2 fused layers: (N, Cin, Cout, H, W of the INPUT, up / pool)
    s2 = [(2, 16, 32, 16, 64, 'pool'), (2, 32, 16, 8, 32, 'up'), (3, 24, 40, 24, 128, 'pool'), (1, 40, 12, 6, 64, 'up'),
          (2, 64, 64, 36, 64, 'pool'), (2, 16, 16, 68, 32, 'up')]
    for n, ci, co, h, w, kind in s2:
        up, pool = kind == 'up', kind == 'pool'
        x = torch.randn(n, ci, h, w, device='cuda')
        g = ops.Geom(n, ci, h, w, co, 3, 1, int(up), int(pool))
        assert g.s2, (n, ci, co, h, w, kind)
        gy = torch.randn(*g.out_shape, device='cuda')
        ref = ref_wgrad(gy, x, 1, up, pool)
        for roll in (True, False):
            out = run(gy, x, g, roll).double().cpu()
            err = ((out - ref).abs().max() / ref.abs().max()).item()
            print(f'{"roll" if roll else "tile"} s2-{kind} N{n} {ci}->{co} in {h}x{w}: rel err {err:.2e}', flush=True)
            worst = max(worst, err)
    assert worst < 1e-5, worst
    if a.big:
        cases = [(32, 16, 16, 1024, ''), (32, 32, 32, 512, ''),
                 (32, 16, 32, 1024, 'pool'), (32, 32, 64, 512, 'pool'), (32, 64, 128, 256, 'pool'),
                 (32, 128, 256, 128, 'pool'), (32, 256, 512, 64, 'pool'), (32, 32, 16, 512, 'up'),
                 (32, 64, 32, 256, 'up'), (32, 512, 256, 32, 'up')]
        for n, ci, co, r, kind in cases:
            up, pool = kind == 'up', kind == 'pool'
            x = torch.randn(n, ci, r, r, device='cuda')
            g = ops.Geom(n, ci, r, r, co, 3, 1, int(up), int(pool))
            gy = torch.randn(*g.out_shape, device='cuda')
            flops = ops.conv_flops(g)
            a_, b_ = run(gy, x, g, True), run(gy, x, g, False)
            err = ((a_ - b_).abs().max() / b_.abs().max()).item()
            res = {}
            variants = [('tile', False, None, '1')] + [(f'roll spu={s_} xcd={xc}', True, s_, xc) for s_ in a.spu
                                                       for xc in ('0', '1')]
            for rnd in range(3):
                for name, roll, spu, xc in variants:
                    os.environ.pop('GANLAB_WR_SPU', None)
                    if spu:
                        os.environ['GANLAB_WR_SPU'] = str(spu)
                    os.environ['GANLAB_WR_XCD'] = xc
                    ms = timeit(lambda: run(gy, x, g, roll), 10)
                    res.setdefault(name, []).append(ms)
            for name, roll, spu, xc in variants:
                ms = min(res[name])
                print(f'{name:20s} N{n} {ci}->{co} @{r}^2 {kind}: {ms:.3f} ms (min of 3x10)  '
                      f'{flops / ms / 1e9:.1f} TFLOP/s = {flops / ms / 1e9 / 157.3:.3f} of peak; roll-vs-tile diff '
                      f'{err:.1e}', flush=True)
    os.environ.pop('GANLAB_WGRAD_ROLL', None)


if __name__ == '__main__':
    main()
