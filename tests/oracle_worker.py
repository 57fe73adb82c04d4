"""Test infrastructure: one D+G step of ``oracle/step.py FunctionalGAN`` in a process of its own.

tests/test_gpu_fullsize.py runs the learner's step at StyleGAN-1024 full width; the CPU oracle needs minutes per
evaluation there, in fp32 AND in float64, so the two evaluations run side by side in two processes (each with half of the
host's cores; ``torch.set_default_dtype`` is process-global, which rules threads out) while the GPU does its part.

    python tests/oracle_worker.py <inputs.pt> <outputs.pt> <float32|float64> <threads>
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(inp, dt):
    from oracle import nets, step
    c = lambda v: v.to(dt)
    gan = step.FunctionalGAN({k: c(v) for k, v in inp['sd_g'].items()}, {k: c(v) for k, v in inp['sd_d'].items()},
                             nets.make_cfg(), model='stylegan', loss=inp['loss'], gp=inp['gp'], lda=inp['lda'],
                             eps_drift=inp['eps_drift'], lr=inp['lr'])
    torch.set_default_dtype(dt)
    o_ld, parts = gan.d_step(c(inp['zd']), c(inp['real']), [c(n) for n in inp['nd']], lr_factor=inp['lr_factor'],
                             cutoff_idx=inp['cut_d'], z_mix=c(inp['zmix_d']))
    o_gd = {k: v.grad.detach().clone() for k, v in gan.d.items() if v.grad is not None}
    o_lg = gan.g_step(c(inp['zg']), [c(n) for n in inp['ng']], lr_factor=inp['lr_factor'], beta=inp['beta'],
                      cutoff_idx=inp['cut_g'], z_mix=c(inp['zmix_g']))
    o_gg = {k: v.grad.detach().clone() for k, v in gan.g.items() if v.grad is not None}
    return dict(ld=o_ld, lg=o_lg, gd=o_gd, gg=o_gg, gp=parts['gp'].detach(), d_real=parts['d_real'].detach(),
                d_fake=parts['d_fake'].detach(), g={k: v.detach() for k, v in gan.g.items()},
                d={k: v.detach() for k, v in gan.d.items()}, lag=gan.lagged)


if __name__ == '__main__':
    src, dst, dtname, threads = sys.argv[1:5]
    torch.set_num_threads(max(1, int(threads)))
    t0 = time.time()
    out = run(torch.load(src), getattr(torch, dtname))
    out['seconds'] = time.time() - t0
    torch.save(out, dst + '.tmp')
    os.replace(dst + '.tmp', dst)
