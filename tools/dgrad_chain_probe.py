#!/usr/bin/env python3
"""Accuracy of the critic's INPUT gradient (what the G step back-propagates into the generator) against the oracle in
float64, with the CPU fp32 oracle beside it: full-width StyleGAN discriminator at several resolutions / batch sizes.
    python tools/dgrad_chain_probe.py [res ...]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from gan_lab_amd import progressive as P
from gan_lab_amd.progan.architectures import StyleDiscriminator
from gan_lab_amd.utils import backprop_utils as bp
from oracle import nets, ops as O


def build(res, seed=3):
    torch.manual_seed(seed)
    P.StyleGAN.reset_state()
    d = StyleDiscriminator(final_res=res, blur_type='binomial')
    for _ in range(int(np.log2(res)) - 2):
        d.increase_scale()
        d.scale_inc_metadata_updated = False      # (the generator normally takes the other half of the lock-step)
    d.fade_in_phase = False
    d.alpha = 1
    with torch.no_grad():
        for k, p in d.named_parameters():
            if k.endswith('bias'):
                p.normal_(0, 0.3)
    return d


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max()).item(), ((a - b).norm() / b.norm()).item()


# draws: (resolution, batch, weight seed) - seed 3 is round 3's pair of draws per resolution, 3 + batch a second network
ratios = []
for res, b, seed in [(r, b, s) for r in ([int(a) for a in sys.argv[1:]] or [32, 128]) for b in (4, 8) for s in (3, 3 + b)]:
    if True:
        d = build(res, seed=seed)
        sd = {k: v.detach().clone() for k, v in d.state_dict().items()}
        gen = torch.Generator().manual_seed(17)
        x = torch.randn(b, 3, res, res, generator=gen) * 0.7
        outs = {}
        for name, dt in (('cpu32', torch.float32), ('cpu64', torch.float64)):
            old = torch.get_default_dtype()
            torch.set_default_dtype(dt)
            xs = x.to(dt).clone().requires_grad_(True)
            o = nets.disc_forward({k: v.to(dt) for k, v in sd.items()}, xs, nets.make_cfg())
            O.loss_gen('nonsaturating', o).backward()
            torch.set_default_dtype(old)
            outs[name] = (o.detach(), xs.grad.detach())
        d.cuda().train()
        for p in d.parameters():
            p.requires_grad_(False)
        xg = x.cuda().requires_grad_(True)
        og = d(xg)
        bp.loss_gen('nonsaturating', og).backward()
        torch.cuda.synchronize()
        ratios.append(rel(xg.grad, outs['cpu64'][1])[1] / rel(outs['cpu32'][1], outs['cpu64'][1])[1])
        print(f'res {res} batch {b} seed {seed}: logits hip/cpu32 vs f64 (max, l2): {rel(og, outs["cpu64"][0])} {rel(outs["cpu32"][0], outs["cpu64"][0])}'
              f' | input grad hip: {rel(xg.grad, outs["cpu64"][1])}  cpu32: {rel(outs["cpu32"][1], outs["cpu64"][1])}', flush=True)
        d.cpu()
ratios.sort()
print('input-gradient L2 error, HIP / CPU fp32 (both against float64), per draw:', [round(r, 2) for r in ratios],
      'median', round(ratios[len(ratios) // 2], 2))
