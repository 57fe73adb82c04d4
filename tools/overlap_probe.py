#!/usr/bin/env python3
"""Would running a layer's input-gradient and weight-gradient kernels on two streams pay?  Times the pair back to back on
one stream against the pair issued on two streams (joined by events), per layer shape of the 1024^2 network.
    python tools/overlap_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_lab_amd import ops

SHAPES = [(32, 512, 32, 512), (32, 256, 64, 256), (32, 128, 128, 128), (32, 64, 256, 64), (32, 32, 512, 32),
          (32, 16, 1024, 16), (32, 512, 16, 512), (32, 512, 8, 512)]


def bench(fn, reps=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    side = torch.cuda.Stream()
    for n, c, r, co in SHAPES:
        x = torch.randn(n, c, r, r, device='cuda')
        w = torch.randn(co, c, 3, 3, device='cuda')
        gy = torch.randn(n, co, r, r, device='cuda')
        g = ops.Geom(n, c, r, r, co, 3, 1)

        def seq():
            ops.k_conv_dgrad(gy, w, g, 0.05)
            ops.k_conv_wgrad(gy, x, g, 0.05)

        def par():
            main_s = torch.cuda.current_stream()
            side.wait_stream(main_s)
            with torch.cuda.stream(side):
                ops.k_conv_wgrad(gy, x, g, 0.05)
            ops.k_conv_dgrad(gy, w, g, 0.05)
            main_s.wait_stream(side)

        t_d = bench(lambda: ops.k_conv_dgrad(gy, w, g, 0.05))
        t_w = bench(lambda: ops.k_conv_wgrad(gy, x, g, 0.05))
        t_s, t_p = bench(seq), bench(par)
        print(f'{c:3d}->{co:3d} @{r:4d}^2 x{n}: dgrad {t_d:.3f} wgrad {t_w:.3f} sum {t_d + t_w:.3f}  one stream {t_s:.3f}  '
              f'two streams {t_p:.3f} ms ({100 * (t_s - t_p) / t_s:+.1f} %)', flush=True)
        del x, w, gy


if __name__ == '__main__':
    main()
