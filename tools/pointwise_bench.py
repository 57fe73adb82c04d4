#!/usr/bin/env python3
"""Achieved HBM bandwidth of the pointwise kernels at StyleGAN-1024 layer shapes (algorithmic bytes / time)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gan_lab_amd import ops  # noqa: E402


def timeit(fn, reps=10):
    for _ in range(2):
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
    for (n, c, r) in [(32, 16, 1024), (32, 64, 256), (32, 512, 32)]:
        x = torch.randn(n, c, r, r, device='cuda')
        y = torch.randn(n, c, r, r, device='cuda')
        nz = torch.randn(n, 1, r, r, device='cuda')
        b = torch.randn(c, device='cuda')
        nw = torch.randn(c, device='cuda')
        sz = x.numel() * 4 / 1e9
        rows = [
            ('blur (1R1W)', 2, lambda: ops.k_blur(x)),
            ('bias_act+noise (1R1W)', 2, lambda: ops.k_bias_act(x, b, nz, nw, 1.0, 1, 0.2)),
            ('act_bwd (2R1W)', 3, lambda: ops.k_act_bwd(x, y, 0.2)),
            ('act_bwd_bias (2R1W)', 3, lambda: ops.k_act_bwd_bias(x, y, 0.2, 1.0)),
            ('blur_bias_act (1R1W)', 2, lambda: ops.k_blur_bias_act(x, b, nz, nw, 1.0, 1, 0.2)),
            ('blur_act_bwd A (2R1W)', 3, lambda: ops.k_blur_act_bwd(x, y, 0.2, 1.0, True)),
            ('act_bwd_blur AT (2R1W)', 3, lambda: ops.k_act_bwd_blur(x, y, nz, 0.2, 1.0, True, True)),
            ('axpby (2R1W)', 3, lambda: ops.k_axpby(x, y, 0.5, 0.5)),
            ('pool2 (1R .25W)', 1.25, lambda: ops.k_pool2(x)),
            ('instnorm stats+apply (2R1W)', 3, lambda: ops._InstNormStyle.apply(x, None, 1e-8)),
        ]
        print(f'--- {n}x{c}x{r}x{r}: {sz:.2f} GB per tensor')
        for name, units, fn in rows:
            ms = timeit(fn)
            print(f'{name:28s} {ms:8.3f} ms  {units * sz / ms * 1e3:7.0f} GB/s')


if __name__ == '__main__':
    main()
