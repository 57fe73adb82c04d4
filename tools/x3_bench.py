#!/usr/bin/env python3
"""Split-product (3 x bf16) conv kernels against the exact-fp32 kernels: rms error against float64 next to ATen's CPU fp32
conv, and ms per launch in interleaved rounds of one process (VERDICT r04 item 1 gate: >= 1.5x and error <= 1.1x ATen's).
    python tools/x3_bench.py [--batch 32] [--err-only] [--time-only]"""
import argparse
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from gan_lab_amd import _lib, ops
from gan_lab_amd._lib import check

LAYERS = [(64, 64, 256), (128, 128, 128), (256, 256, 64), (512, 512, 32), (512, 512, 16)]


def pack_x3(w, mode, scale):
    L = _lib.lib()
    n = L.ganlab_conv_x3_pack(None, None, w.shape[0], w.shape[1], mode, scale, None)
    assert n > 0, n
    out = torch.empty((n,), dtype=torch.bfloat16, device=w.device)
    rc = L.ganlab_conv_x3_pack(w.data_ptr(), out.data_ptr(), w.shape[0], w.shape[1], mode, scale, None)
    assert rc == n, rc
    return out


def fwd_x3(x, wp, bias, g, act=0):
    y = torch.empty(g.out_shape, device=x.device)
    check(_lib.lib().ganlab_conv_fwd_x3(x.data_ptr(), wp.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(),
                                        g.ref(), 1.0, act, 0.2, None), 'conv_fwd_x3')
    return y


def dgrad_x3(gy, wp, g):
    gx = torch.empty(g.in_shape, device=gy.device)
    check(_lib.lib().ganlab_conv_dgrad_x3(gy.data_ptr(), wp.data_ptr(), gx.data_ptr(), g.ref(), None), 'conv_dgrad_x3')
    return gx


def err(a, ref):
    a, ref = a.double().cpu(), ref.double()
    return ((a - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()


def errors():
    torch.manual_seed(0)
    torch.set_num_threads(16)
    for ci, co, hw in [(64, 64, 64), (128, 128, 32), (256, 256, 32), (512, 512, 16), (256, 128, 16), (64, 192, 16)]:
        n = 2
        x = torch.randn(n, ci, hw, hw)
        w = torch.randn(co, ci, 3, 3) / (3 * ci ** 0.5)
        b = torch.randn(co)
        g = ops.Geom(n, ci, hw, hw, co, 3, 1)
        exact = F.conv2d(x.double(), w.double(), b.double(), padding=1)
        cpu = F.conv2d(x, w, b, padding=1)
        hip = ops.k_conv_fwd(x.cuda(), w.cuda(), b.cuda(), g, 1.0)
        x3 = fwd_x3(x.cuda(), pack_x3(w.cuda(), 0, 1.0), b.cuda(), g)
        ec, eh, e3 = err(cpu, exact), err(hip, exact), err(x3, exact)
        print(f'fwd   {ci:3d}->{co:3d} {hw}x{hw}: ATen {ec:.3e} | fp32 MFMA {eh:.3e} ({eh / ec:.2f}) | 3xbf16 {e3:.3e} ({e3 / ec:.2f})', flush=True)
        gy = torch.randn(n, co, hw, hw)
        xd = x.double().requires_grad_(True)
        exact, = torch.autograd.grad(F.conv2d(xd, w.double(), padding=1), xd, gy.double())
        xf = x.clone().requires_grad_(True)
        cpu, = torch.autograd.grad(F.conv2d(xf, w, padding=1), xf, gy)
        hip = ops.k_conv_dgrad(gy.cuda(), w.cuda(), g, 1.0)
        x3 = dgrad_x3(gy.cuda(), pack_x3(w.cuda(), 1, 1.0), g)
        ec, eh, e3 = err(cpu, exact), err(hip, exact), err(x3, exact)
        print(f'dgrad {ci:3d}->{co:3d} {hw}x{hw}: ATen {ec:.3e} | fp32 MFMA {eh:.3e} ({eh / ec:.2f}) | 3xbf16 {e3:.3e} ({e3 / ec:.2f})', flush=True)


def wgrad_report(batch, rounds, reps):
    """Weight gradient: error against float64 next to ATen's CPU fp32 (batch 4) and ms per launch at the bench batch."""
    torch.manual_seed(1)
    for ci, co, hw in [(64, 64, 128), (128, 128, 64), (256, 256, 32)]:
        n = 4
        x, gy = torch.randn(n, ci, hw, hw), torch.randn(n, co, hw, hw)
        g = ops.Geom(n, ci, hw, hw, co, 3, 1)
        wd = torch.zeros(co, ci, 3, 3, dtype=torch.float64, requires_grad=True)
        exact, = torch.autograd.grad(F.conv2d(x.double(), wd, padding=1), wd, gy.double())
        wf = torch.zeros(co, ci, 3, 3, requires_grad=True)
        cpu, = torch.autograd.grad(F.conv2d(x, wf, padding=1), wf, gy)
        x3 = ops.k_conv_wgrad(gy.cuda(), x.cuda(), g, 1.0)
        prev = ops.set_x3(False)
        hip = ops.k_conv_wgrad(gy.cuda(), x.cuda(), g, 1.0)
        ops.set_x3(prev)
        ec, eh, e3 = err(cpu, exact), err(hip, exact), err(x3, exact)
        print(f'wgrad {ci:3d}->{co:3d} {hw}x{hw} x{n}: ATen {ec:.3e} | fp32 MFMA {eh:.3e} ({eh / ec:.2f}) | 3xbf16 {e3:.3e} ({e3 / ec:.2f})', flush=True)
    for ci, co, hw in LAYERS[:4]:
        x = torch.randn(batch, ci, hw, hw, device='cuda')
        gy = torch.randn(batch, co, hw, hw, device='cuda')
        g = ops.Geom(batch, ci, hw, hw, co, 3, 1)
        fl = ops.conv_flops(g)
        ms = {'fp32': [], 'x3': []}
        def run(on):
            prev = ops.set_x3(on)
            try:
                return ops.k_conv_wgrad(gy, x, g, 0.05)
            finally:
                ops.set_x3(prev)
        for on in (False, True):
            for _ in range(4):
                run(on)
        for _ in range(rounds):
            for k, on in (('fp32', False), ('x3', True)):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    run(on)
                e1.record()
                torch.cuda.synchronize()
                ms[k].append(e0.elapsed_time(e1) / reps)
        med = {k: sorted(v)[len(v) // 2] for k, v in ms.items()}
        print(f'wgrad {ci:3d}->{co:3d} @{hw:3d} x{batch}: fp32 MFMA {med["fp32"]:.3f} ms ({fl / med["fp32"] / 1e9:6.1f} TF/s)  3xbf16 {med["x3"]:.3f} ms '
              f'({fl / med["x3"] / 1e9:6.1f} TF/s fp32-equivalent)  speed-up {med["fp32"] / med["x3"]:.2f}x', flush=True)
        del x, gy


def s2_wgrad_report(batch, rounds, reps):
    """Stride-2 weight gradient (box form): error against float64 next to ATen's CPU fp32 (batch 2), ms per launch at the bench batch."""
    torch.manual_seed(2)
    for ci, co, hl, kind in [(64, 128, 64, 'pool'), (128, 64, 64, 'up'), (256, 512, 32, 'pool')]:
        n, up = 2, kind == 'up'
        hi = hl if up else 2 * hl
        x = torch.randn(n, ci, hi, hi)
        gy = torch.randn(n, co, 2 * hl, 2 * hl) if up else torch.randn(n, co, hl, hl)
        g = ops.Geom(n, ci, hi, hi, co, 3, 1, up=1) if up else ops.Geom(n, ci, hi, hi, co, 3, 1, pool=1)

        def ref(xx, ww):
            return F.conv2d(F.interpolate(xx, scale_factor=2, mode='nearest'), ww, padding=1) if up else F.avg_pool2d(F.conv2d(xx, ww, padding=1), 2)
        wd = torch.zeros(co, ci, 3, 3, dtype=torch.float64, requires_grad=True)
        exact, = torch.autograd.grad(ref(x.double(), wd), wd, gy.double())
        wf = torch.zeros(co, ci, 3, 3, requires_grad=True)
        cpu, = torch.autograd.grad(ref(x, wf), wf, gy)
        x3 = ops.k_conv_wgrad(gy.cuda(), x.cuda(), g, 1.0)
        name = _lib.last_launch()[0]
        prev = ops.set_x3(False)
        hip = ops.k_conv_wgrad(gy.cuda(), x.cuda(), g, 1.0)
        ops.set_x3(prev)
        ec, eh, e3 = err(cpu, exact), err(hip, exact), err(x3, exact)
        print(f's2 wgrad {kind:4s} {ci:3d}->{co:3d} low {hl}x{hl} x{n}: ATen {ec:.3e} | fp32 MFMA {eh:.3e} ({eh / ec:.2f}) | 3xbf16 box {e3:.3e} ({e3 / ec:.2f})'
              f'  [{name.split("(")[0][-40:]}]', flush=True)
    layers = [(32, 64, 256, 'pool'), (64, 128, 128, 'pool'), (128, 256, 64, 'pool'), (256, 512, 32, 'pool'),
              (64, 32, 256, 'up'), (128, 64, 128, 'up'), (256, 128, 64, 'up'), (512, 256, 32, 'up')]
    for ci, co, hl, kind in layers:
        up = kind == 'up'
        hi = hl if up else 2 * hl
        x = torch.randn(batch, ci, hi, hi, device='cuda')
        gy = torch.randn(batch, co, 2 * hl, 2 * hl, device='cuda') if up else torch.randn(batch, co, hl, hl, device='cuda')
        g = ops.Geom(batch, ci, hi, hi, co, 3, 1, up=1) if up else ops.Geom(batch, ci, hi, hi, co, 3, 1, pool=1)
        fl = 2.0 * 16 * ci * co * batch * hl * hl          # the 16-tap low-resolution form the exact kernel is priced on
        ms = {'fp32': [], 'x3': []}

        def run(on):
            prev = ops.set_x3(on)
            try:
                return ops.k_conv_wgrad(gy, x, g, 0.05)
            finally:
                ops.set_x3(prev)
        for on in (False, True):
            for _ in range(4):
                run(on)
        for _ in range(rounds):
            for k, on in (('fp32', False), ('x3', True)):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    run(on)
                e1.record()
                torch.cuda.synchronize()
                ms[k].append(e0.elapsed_time(e1) / reps)
        med = {k: sorted(v)[len(v) // 2] for k, v in ms.items()}
        gb = (x.numel() + gy.numel()) * 4 / 1e9
        print(f's2 wgrad {kind:4s} {ci:3d}->{co:3d} low {hl:3d} x{batch}: fp32 MFMA {med["fp32"]:.3f} ms ({fl / med["fp32"] / 1e9:6.1f} TF/s 16-tap)  '
              f'3xbf16 box {med["x3"]:.3f} ms ({gb / med["x3"] * 1e3:5.0f} GB/s of operands)  speed-up {med["fp32"] / med["x3"]:.2f}x', flush=True)
        del x, gy


def times(batch, rounds, reps, forms=False):
    for ci, co, hw in LAYERS:
        x = torch.randn(batch, ci, hw, hw, device='cuda')
        w = torch.randn(co, ci, 3, 3, device='cuda')
        g = ops.Geom(batch, ci, hw, hw, co, 3, 1)
        wp3 = pack_x3(w, 0, 0.05)
        fl = ops.conv_flops(g)
        fns = {'fp32': lambda: ops.k_conv_fwd(x, w, None, g, 0.05), 'x3': lambda: fwd_x3(x, wp3, None, g)}
        if forms and not ops.conv_tail_shape_ok(x.shape, w):
            continue
        if forms:
            def sw(fn, on):
                def f():
                    prev = ops.set_x3(on)
                    try:
                        return fn()
                    finally:
                        ops.set_x3(prev)
                return f
            gy = torch.randn(*g.out_shape, device='cuda')
            s_, t_ = torch.rand(batch, ci, device='cuda') + 0.5, torch.randn(batch, ci, device='cuda')
            mask = lambda: ops.k_conv_dgrad_mask(gy, w, x, g, 0.05, 0.2)
            aff = lambda: ops.k_conv_fwd_aff(x, s_, t_, w, g, 0.05)
            bias, nz, nw = torch.randn(co, device='cuda'), torch.randn(batch, 1, hw, hw, device='cuda'), torch.randn(co, device='cuda')
            tail = lambda: ops._ConvModTail.apply(x, s_, t_, w, bias, nz, nw, None, 0.05, 1.0, ops.ACT_LRELU, 0.2, 1e-8)
            fns = {'fp32': sw(mask, False), 'x3': sw(mask, True), 'aff fp32': sw(aff, False), 'aff x3': sw(aff, True),
                   'tail fp32': sw(tail, False), 'tail x3': sw(tail, True)}
        d = 0.0 if forms else (fns['fp32']() - fns['x3']()).abs().max().item()
        ms = {k: [] for k in fns}
        for k in fns:
            for _ in range(5):
                fns[k]()
        for _ in range(rounds):
            for k, fn in fns.items():
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ms[k].append(e0.elapsed_time(e1) / reps)
        med = {k: sorted(v)[len(v) // 2] for k, v in ms.items()}
        if forms:
            print(f'{ci:3d}->{co:3d} @{hw:3d} x{batch}: masked dgrad {med["fp32"]:.3f} -> {med["x3"]:.3f} ms ({med["fp32"] / med["x3"]:.2f}x) | affine fwd '
                  f'{med["aff fp32"]:.3f} -> {med["aff x3"]:.3f} ({med["aff fp32"] / med["aff x3"]:.2f}x) | layer tail {med["tail fp32"]:.3f} -> '
                  f'{med["tail x3"]:.3f} ({med["tail fp32"] / med["tail x3"]:.2f}x)', flush=True)
            continue
        print(f'{ci:3d}->{co:3d} @{hw:3d} x{batch}: fp32 MFMA {med["fp32"]:.3f} ms ({fl / med["fp32"] / 1e9:6.1f} TF/s)  3xbf16 {med["x3"]:.3f} ms '
              f'({fl / med["x3"] / 1e9:6.1f} TF/s fp32-equivalent, {6 * fl / med["x3"] / 1e9:7.1f} bf16)  speed-up {med["fp32"] / med["x3"]:.2f}x  '
              f'max |diff| {d:.2e}', flush=True)
        del x, w


if __name__ == '__main__':
    p = argparse.ArgumentParser()
    p.add_argument('--batch', type=int, default=32)
    p.add_argument('--rounds', type=int, default=5)
    p.add_argument('--reps', type=int, default=10)
    p.add_argument('--err-only', action='store_true')
    p.add_argument('--time-only', action='store_true')
    p.add_argument('--wgrad', action='store_true', help='weight gradient: error and time')
    p.add_argument('--s2wgrad', action='store_true', help='stride-2 weight gradient (box form): error and time')
    p.add_argument('--forms', action='store_true', help='time the masked / affine / layer-tail forms through ops')
    a = p.parse_args()
    if a.wgrad:
        wgrad_report(a.batch, a.rounds, a.reps)
        sys.exit(0)
    if a.s2wgrad:
        s2_wgrad_report(a.batch, a.rounds, a.reps)
        sys.exit(0)
    if not a.time_only:
        errors()
    if not a.err_only:
        times(a.batch, a.rounds, a.reps)
        if a.forms:
            times(a.batch, a.rounds, a.reps, True)
