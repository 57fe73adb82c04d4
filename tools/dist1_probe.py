#!/usr/bin/env python3
"""Does the data-parallel code path on ONE rank (GANLAB_DIST_WORLD1=1: hooks, bucket launches from inside the backward, RCCL
all-reduce over a 1-rank group) leave the same parameters as the plain path?  Prints loss and parameter checksums per step.
    python tools/dist1_probe.py [steps] [res] [batch]        (run twice: with and without GANLAB_DIST_WORLD1=1)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
res = int(sys.argv[2]) if len(sys.argv) > 2 else 256
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 8
if os.environ.get('GANLAB_DIST_WORLD1') == '1':
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1)
import bench
torch.manual_seed(3)
import numpy as np
np.random.seed(3)
L = bench.build_learner(res, batch, 'cuda', 'f32', 'stylegan')
gen = torch.Generator().manual_seed(5)
for i in range(steps):
    real = (torch.rand(batch, 3, res, res, generator=gen) * 2 - 1).cuda()
    ld, lg = bench.one_step(L, real)
    torch.cuda.synchronize()
    print(f'step {i}: loss_d {float(ld)!r} loss_g {float(lg)!r} sum|D| {L.arena_d.flat.double().abs().sum().item()!r} '
          f'sum|G| {L.arena_g.flat.double().abs().sum().item()!r} sum|gD| {L.arena_d.gflat.double().abs().sum().item()!r}')
