#!/usr/bin/env python3
"""Per-layer conv timing of one G+D step at the bench configuration: wraps the conv launchers of
gan_lab_amd.ops with device events and prints, per (kind, geometry): calls, total ms, executed TFLOP/s
(stride-2 fused layers priced with their 16 low-res taps) and the share of the step.
    python tools/step_layers.py [--res 1024] [--batch 32] [--model stylegan|progan|resnetgan]"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from gan_lab_amd import ops  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--res', type=int, default=1024)
    p.add_argument('--batch', type=int, default=32)
    p.add_argument('--model', default='stylegan', choices=('stylegan', 'progan', 'resnetgan'))
    a = p.parse_args()
    torch.cuda.set_device(0)
    if a.model == 'resnetgan':      # BASELINE config #5: one main iteration = 1 G + 5 critic iterations
        import types
        wl = bench.Workload(types.SimpleNamespace(model='resnetgan', res=a.res, batch=a.batch, dtype='f32', world=1,
                                                  nimg_transition=0), torch)
        one = wl.step
    else:
        L = bench.build_learner(a.res, a.batch, 'cuda', 'f32', a.model)
        real = torch.rand(a.batch, 3, a.res, a.res, device='cuda') * 2 - 1
        one = lambda: bench.one_step(L, real)
    one()
    one()
    rec = []

    def wrap(name, fn, gi):
        def f(*args, **kw):
            g = args[gi]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*args, **kw)
            e1.record()
            rec.append((name, g, e0, e1))
            return out
        return f

    ops.k_conv_fwd = wrap('fwd', ops.k_conv_fwd, 3)
    ops.k_conv_dgrad = wrap('dgrad', ops.k_conv_dgrad, 2)
    ops.k_conv_wgrad = wrap('wgrad', ops.k_conv_wgrad, 2)
    # the variants with a LeakyReLU derivative folded in (masked dgrad epilogue; fromRGB's streaming kernels)
    ops.k_conv_dgrad_mask = wrap('dgrad', ops.k_conv_dgrad_mask, 3)
    ops.k_conv_dgrad_act = wrap('dgrad', ops.k_conv_dgrad_act, 3)
    ops.k_conv_wgrad_act = wrap('wgrad', ops.k_conv_wgrad_act, 3)
    ops.k_conv_fwd_mask = wrap('fwd', ops.k_conv_fwd_mask, 3)
    # deferred-InstanceNorm consumers (affine on load): k_conv_fwd_aff(a, s, t, w, g, scale), k_conv_wgrad_aff(gy, a, s, t, g, ..)
    ops.k_conv_fwd_aff = wrap('fwd', ops.k_conv_fwd_aff, 4)
    ops.k_conv_wgrad_aff = wrap('wgrad', ops.k_conv_wgrad_aff, 4)
    # the rolling kernels with the blur folded in (the blur / tail pass they absorb is part of their time, not of their FLOPs):
    # k_conv_fwd_blur_bits(x, w, bias, g, ..), k_conv_s2_fwd_blur_tail(a, s, t, w, bias, noise, nw, g, ..),
    # k_conv_s2_dgrad_blur_act(gy, w, bits, g, ..)
    def wrap_opt(name, fn, gi):          # these return None where they do not take the geometry: not a launch
        def f(*args, **kw):
            g = args[gi]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*args, **kw)
            e1.record()
            if out is not None:
                rec.append((name, g, e0, e1))
            return out
        return f
    ops.k_conv_fwd_blur_bits = wrap_opt('fwd+blur', ops.k_conv_fwd_blur_bits, 3)
    ops.k_conv_s2_fwd_blur_tail = wrap('fwd+blur+tail', ops.k_conv_s2_fwd_blur_tail, 7)
    ops.k_conv_s2_dgrad_blur_act = wrap("dgrad+blur+act'", ops.k_conv_s2_dgrad_blur_act, 3)
    ops.k_conv_dgrad_rgb_sums = wrap('dgrad+fromRGB bwd', ops.k_conv_dgrad_rgb_sums, 3)    # (gz, w, handoff, g, scale)
    torch.cuda.synchronize()
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record()
    one()
    s1.record()
    torch.cuda.synchronize()
    step_ms = s0.elapsed_time(s1)
    agg = collections.OrderedDict()
    for name, g, e0, e1 in rec:
        if g.s2:
            lo_h, lo_w = (g.Hin, g.Win) if g.up else (g.Ho, g.Wo)
            flops = 2.0 * 16 * g.Cin * g.Cout * lo_h * lo_w * g.N
            tag = 'up' if g.up else 'pool'
        else:
            hv, wv = (2 * g.Hin, 2 * g.Win) if g.up else (g.Hin, g.Win)
            ho, wo = hv + 2 * g.pad - g.ks + 1, wv + 2 * g.pad - g.ks + 1
            flops = 2.0 * g.ks * g.ks * g.Cin * g.Cout * ho * wo * g.N
            tag = 'up(unfused)' if g.up else ''
        key = (name, g.ks, g.Cin, g.Cout, g.Hin, tag)
        v = agg.setdefault(key, [0, 0.0, 0.0])
        v[0] += 1
        v[1] += e0.elapsed_time(e1)
        v[2] += flops
    tot = sum(v[1] for v in agg.values())
    print(f'step {step_ms:.1f} ms; conv launchers {tot:.1f} ms ({100 * tot / step_ms:.0f}%), '
          f'executed conv TFLOP {sum(v[2] for v in agg.values()) / 1e12:.2f}')
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f'{v[1]:8.2f} ms {v[0]:3d}x  {v[2] / v[1] / 1e9:6.1f} TF/s  {k}')


if __name__ == '__main__':
    main()
