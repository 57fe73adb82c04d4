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


def mixed():
    """An MFMA-bound thick conv next to an HBM-bound pointwise pass (two independent critic passes half a network apart
    would put such pairs side by side): back to back on one stream vs on two streams."""
    side = torch.cuda.Stream()
    convs = [(32, 256, 64, 256), (32, 512, 32, 512), (32, 128, 128, 128)]
    passes = [('blur 16ch 1024^2', lambda t: ops.k_blur(t), (32, 16, 1024, 1024)),
              ('blur 32ch 512^2', lambda t: ops.k_blur(t), (32, 32, 512, 512)),
              ('channel sum 16ch 1024^2', lambda t: ops.k_channel_sum(t), (32, 16, 1024, 1024))]
    for n, c, r, co in convs:
        x = torch.randn(n, c, r, r, device='cuda')
        w = torch.randn(co, c, 3, 3, device='cuda')
        g = ops.Geom(n, c, r, r, co, 3, 1)
        conv = lambda: ops.k_conv_fwd(x, w, None, g, 0.05)      # noqa: E731
        for name, fn, shape in passes:
            t = torch.randn(*shape, device='cuda')
            pw = lambda: fn(t)                                   # noqa: E731
            k = max(1, round(bench(conv) / bench(pw)))          # as many pointwise passes as fill one conv

            def seq():
                conv()
                for _ in range(k):
                    pw()

            def par():
                main_s = torch.cuda.current_stream()
                side.wait_stream(main_s)
                with torch.cuda.stream(side):
                    for _ in range(k):
                        pw()
                conv()
                main_s.wait_stream(side)

            t_c, t_p1 = bench(conv), bench(pw)
            t_s, t_p = bench(seq), bench(par)
            print(f'conv {c}->{co} @{r}^2 {t_c:.3f} ms + {k} x {name} {t_p1:.3f} ms: one stream {t_s:.3f}, two streams '
                  f'{t_p:.3f} ms ({100 * (t_s - t_p) / t_s:+.1f} %)', flush=True)
            del t
        del x, w


if __name__ == '__main__':
    mixed() if 'mixed' in sys.argv[1:] else main()
