#!/usr/bin/env python3
"""Where the critic's input gradient loses accuracy: LeakyReLU mask bits that differ from the float64 evaluation, layer
by layer, for the HIP path and for the CPU fp32 oracle (full-width StyleGAN critic).  One flipped bit changes gz by
0.8 * gy at one element - the error of d loss / d image is made of these (tools/dgrad_chain_probe.py).
    python tools/flip_probe.py [res] [batch]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.nn.functional as F
from gan_lab_amd import ops, progressive as P
from gan_lab_amd.progan.architectures import StyleDiscriminator
from oracle import nets

res = int(sys.argv[1]) if len(sys.argv) > 1 else 128
b = int(sys.argv[2]) if len(sys.argv) > 2 else 4
torch.manual_seed(3)
P.StyleGAN.reset_state()
d = StyleDiscriminator(final_res=res, blur_type='binomial')
for _ in range(int(np.log2(res)) - 2):
    d.increase_scale()
    d.scale_inc_metadata_updated = False
d.fade_in_phase = False
d.alpha = 1
with torch.no_grad():
    for k, p in d.named_parameters():
        if k.endswith('bias'):
            p.normal_(0, 0.3)
sd = {k: v.detach().clone() for k, v in d.state_dict().items()}
x = torch.randn(b, 3, res, res, generator=torch.Generator().manual_seed(17)) * 0.7

logs = {}
orig_lrelu = F.leaky_relu


def run_oracle(name, dt):
    rec = []
    F.leaky_relu = lambda t, s=0.01, *a, **k: (rec.append(orig_lrelu(t, s, *a, **k)), rec[-1])[1]
    old = torch.get_default_dtype()
    torch.set_default_dtype(dt)
    try:
        with torch.no_grad():
            nets.disc_forward({k: v.to(dt) for k, v in sd.items()}, x.to(dt), nets.make_cfg())
    finally:
        torch.set_default_dtype(old)
        F.leaky_relu = orig_lrelu
    logs[name] = [t.detach() for t in rec]


run_oracle('cpu64', torch.float64)
run_oracle('cpu32', torch.float32)

rec = []
for fn_name in ('conv2d', 'bias_act', 'linear'):
    fn = getattr(ops, fn_name)

    def wrapped(*a, _fn=fn, **k):
        y = _fn(*a, **k)
        if k.get('act') == 'lrelu' and torch.is_tensor(y) and not k.get('blur', False):
            rec.append(y.detach().cpu())
        return y
    setattr(ops, fn_name, wrapped)
from gan_lab_amd.utils import custom_layers as CL      # noqa: E402  (binds ops.* at call time)
d.cuda().train()
with torch.no_grad():
    d(x.cuda())
torch.cuda.synchronize()
logs['hip'] = rec
print('LeakyReLU outputs logged:', {k: len(v) for k, v in logs.items()})
ref = logs['cpu64']
for name in ('hip', 'cpu32'):
    got = logs[name]
    print(f'--- {name}')
    used = set()
    for i, r in enumerate(ref):
        # match by shape and order
        j = next((j for j, g in enumerate(got) if j not in used and tuple(g.shape) == tuple(r.shape)), None)
        if j is None:
            print(f'  layer {i} {tuple(r.shape)}: not materialised on this path')
            continue
        used.add(j)
        g = got[j].double()
        flips = int(((g > 0) != (r > 0)).sum())
        err = ((g - r).norm() / r.norm()).item()
        print(f'  layer {i} {tuple(r.shape)}: flipped mask bits {flips:6d} of {r.numel():9d}   rel L2 error of the activation {err:.2e}')
