#!/usr/bin/env python3
"""Per-layer microbenchmark of the MFMA conv kernels on the layer shapes of StyleGAN-1024 at batch 32:
forward, input gradient and weight gradient TFLOP/s (algorithmic FLOPs 2*k^2*Cin*Cout*Ho*Wo*N)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_lab_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
if len(sys.argv) > 2:          # tools/conv_bench.py 8 bf16 -> config #2 shapes on the bf16-compute kernels
    ops.set_compute_dtype(sys.argv[2])
SHAPES = [  # (Cin, Cout, Hin, up)
    (16, 16, 1024, 0), (32, 16, 512, 1), (16, 32, 1024, 0), (32, 32, 512, 0), (64, 32, 256, 1), (32, 64, 512, 0),
    (64, 64, 256, 0), (128, 64, 128, 1), (64, 128, 256, 0), (128, 128, 128, 0), (256, 128, 64, 1),
    (128, 256, 128, 0), (256, 256, 64, 0), (512, 256, 32, 1), (256, 512, 64, 0), (512, 512, 32, 0),
    (512, 512, 16, 0), (512, 512, 8, 0), (512, 512, 4, 0),
]
if ops.get_compute_dtype() == 'bf16':      # StyleGAN-128 layer shapes
    SHAPES = [(128, 128, 128, 0), (256, 128, 64, 1), (128, 256, 128, 0), (256, 256, 64, 0), (512, 256, 32, 1),
              (256, 512, 64, 0), (512, 512, 32, 0)]


def timeit(fn, reps=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print(f'{"shape":>28} {"GFLOP":>8} | {"fwd ms":>8} {"TF/s":>6} | {"dgrad ms":>8} {"TF/s":>6} | {"wgrad ms":>8} {"TF/s":>6}')
tot = [0, 0, 0, 0]
for cin, cout, h, up in SHAPES:
    x = torch.randn(B, cin, h, h, device='cuda')
    w = torch.randn(cout, cin, 3, 3, device='cuda')
    g = ops.Geom(B, cin, h, h, cout, 3, 1, up)
    gy = torch.randn(*g.out_shape, device='cuda')
    fl = 2.0 * 9 * cin * cout * g.Ho * g.Wo * B
    tf = timeit(lambda: ops.k_conv_fwd(x, w, None, g, 0.05))
    td = timeit(lambda: ops.k_conv_dgrad(gy, w, g, 0.05))
    tw = timeit(lambda: ops.k_conv_wgrad(gy, x, g, 0.05))
    tot[0] += fl; tot[1] += tf; tot[2] += td; tot[3] += tw
    print(f'{cin:4d}->{cout:4d} @{h:4d}{"^" if up else " "} {fl/1e9:8.1f} | {tf:8.3f} {fl/tf/1e9:6.1f} | {td:8.3f} {fl/td/1e9:6.1f} | '
          f'{tw:8.3f} {fl/tw/1e9:6.1f}')
    del x, w, gy
print(f'{"total":>28} {tot[0]/1e9:8.1f} | {tot[1]:8.3f} {tot[0]/tot[1]/1e9:6.1f} | {tot[2]:8.3f} {tot[0]/tot[2]/1e9:6.1f} | '
      f'{tot[3]:8.3f} {tot[0]/tot[3]/1e9:6.1f}')
