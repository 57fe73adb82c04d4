#!/usr/bin/env python3
"""Probe: pointwise-kernel variants (env toggles, one child process each) vs a plain device copy."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    from gan_lab_amd import ops, _lib
    from tools.pointwise_bench import timeit
    tag = ' '.join(f'{k}={os.environ.get(k)}' for k in ('GANLAB_PW_CHUNK', 'GANLAB_PW_CONTIG'))
    for (n, c, r) in [(32, 16, 1024), (32, 32, 512), (32, 64, 256)]:
        x = torch.randn(n, c, r, r, device='cuda'); y = torch.empty_like(x)
        nz = torch.randn(n, 1, r, r, device='cuda'); b = torch.randn(c, device='cuda'); nw = torch.randn(c, device='cuda')
        mean = torch.zeros(n, c, device='cuda'); rstd = torch.ones(n, c, device='cuda'); st = torch.randn(n, 2 * c, device='cuda')
        sz = x.numel() * 4 / 1e9
        L = _lib.lib()
        f = lambda: ops.check(L.ganlab_instnorm_style_fwd_f32(ops._p(x), ops._p(mean), ops._p(rstd), ops._p(st), ops._p(y), n, c, r * r, ops._st()), 'x')
        ms = timeit(f, 20); ms2 = timeit(lambda: y.copy_(x), 20)
        ms3 = timeit(lambda: ops.k_bias_act_stats(x, b, nz, nw, 1.0, 1, 0.2, 1e-8), 20)
        xin = x.clone().requires_grad_(True)
        out = ops.layer_tail(xin, b.view(1, -1, 1, 1).clone().requires_grad_(True), nz, nw.view(1, -1, 1, 1).clone().requires_grad_(True), st, act='lrelu', slope=0.2, blur=False, eps=1e-8)
        g = torch.randn_like(out)
        ms4 = timeit(lambda: torch.autograd.grad(out, xin, g, retain_graph=True), 10)
        print(f'{tag} {n}x{c}x{r}^2: apply {ms:.3f} ms {2*sz/ms*1e3:.0f} GB/s | copy {ms2:.3f} {2*sz/ms2*1e3:.0f} | '
              f'bias_act_stats {ms3:.3f} {2*sz/ms3*1e3:.0f} | tail bwd (reduce 2R + apply 2R1W) {ms4:.3f} {5*sz/ms4*1e3:.0f}', flush=True)
else:
    for contig in ('0', '1'):
        subprocess.run([sys.executable, __file__, 'child'], env=dict(os.environ, GANLAB_PW_CHUNK='2', GANLAB_PW_CONTIG=contig))
