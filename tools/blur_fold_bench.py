#!/usr/bin/env python3
"""The three rolling-window kernels that end in the blur (csrc/conv_roll_blur.hip, csrc/conv_s2_roll_blur.hip) at the
benchmark's size, each next to the two launches it replaces (conv kernel + blur pass):
    python tools/blur_fold_bench.py [batch] [high_res] [short]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_lab_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
HI = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
SHORT = len(sys.argv) > 3
LO = HI // 2


def timeit(fn, warm=8, reps=16):
    if SHORT:
        warm, reps = 2, 4
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def report(name, flop, fused, parts):
    f = timeit(fused)
    ps = [timeit(p) for p in parts]
    print(f'{name:44s} fused {f:6.3f} ms ({flop / f / 1e9:5.1f} TFLOP/s, {flop / f / 1e9 / 157.3:4.2f})   separate '
          f'{" + ".join(f"{p:5.3f}" for p in ps)} = {sum(ps):6.3f} ms')


# D forward: conv 16 -> 16 + bias + LeakyReLU + blur (+ bits)
x = torch.randn(B, 16, HI, HI, device='cuda')
w = torch.randn(16, 16, 3, 3, device='cuda')
b = torch.randn(16, device='cuda')
g = ops.Geom(B, 16, HI, HI, 16, 3, 1, 0)
keep = {}


def conv_plain():
    keep['y'] = ops.k_conv_fwd(x, w, b, g, 0.05, 1.0, ops.ACT_LRELU, 0.2)


conv_plain()
report('conv 16->16 + LeakyReLU + blur (D forward)', 2.0 * 9 * 16 * 16 * HI * HI * B,
       lambda: ops.k_conv_fwd_blur_bits(x, w, b, g, 0.05, 1.0, 0.2), [conv_plain, lambda: ops.k_blur_bits(keep['y'])])
bits = ops.k_blur_bits(keep['y'])[1]
del x, keep['y']
# D backward: pooled conv's input gradient + blur^T + LeakyReLU' + bias gradient
wb = torch.randn(32, 16, 3, 3, device='cuda')
gp = ops.Geom(B, 16, HI, HI, 32, 3, 1, 0, 1)
gy = torch.randn(*gp.out_shape, device='cuda')


def dgrad_plain():
    keep['g'] = ops.k_conv_dgrad(gy, wb, gp, 0.05)


dgrad_plain()
report('pooled-conv dgrad + blur^T + act\' (D backward)', 2.0 * 16 * 16 * 32 * LO * LO * B,
       lambda: ops.k_conv_s2_dgrad_blur_act(gy, wb, bits, gp, 0.05, 0.2, 1.0, True),
       [dgrad_plain, lambda: ops.k_blur_act_bwd(keep['g'], bits, 0.2, 1.0, True)])
del gy, keep['g'], bits
# G forward: up-conv 32 -> 16 (deferred input) + blur + noise + bias + LeakyReLU + statistics
a = torch.randn(B, 32, LO, LO, device='cuda')
s_, t_ = torch.rand(B, 32, device='cuda') + 0.5, torch.randn(B, 32, device='cuda')
wu = torch.randn(16, 32, 3, 3, device='cuda')
gu = ops.Geom(B, 32, LO, LO, 16, 3, 1, 1, 0)
nz = torch.randn(B, 1, HI, HI, device='cuda')
bb, nw = torch.randn(1, 16, 1, 1, device='cuda'), torch.randn(1, 16, 1, 1, device='cuda')


def up_plain():
    keep['c'] = ops.k_conv_fwd_aff(a, s_, t_, wu, gu, 0.05)


up_plain()
report('up-conv 32->16 (AFF) + blur + tail (G forward)', 2.0 * 16 * 32 * 16 * LO * LO * B,
       lambda: ops.k_conv_s2_fwd_blur_tail(a, s_, t_, wu, bb, nz, nw, gu, 0.05, 1.0, ops.ACT_LRELU, 0.2, 1e-8),
       [up_plain, lambda: ops.k_blur_bias_act_stats(keep['c'], bb, nz, nw, 1.0, ops.ACT_LRELU, 0.2, 1e-8)])
