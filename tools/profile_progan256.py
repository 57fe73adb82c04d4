#!/usr/bin/env python3
"""ProGAN-256 stabilised phase (BASELINE config #4's network, WGAN + WGAN-GP, batch 32) without the schedule in front:
    rocprofv3 --kernel-trace --stats -- python3 tools/profile_progan256.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
torch.cuda.set_device(0)
L = bench.build_learner(256, 32, 'cuda', 'f32', 'progan')
real = torch.rand(32, 3, 256, 256, device='cuda') * 2 - 1
for _ in range(6):
    bench.one_step(L, real)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(5):
    bench.one_step(L, real)
torch.cuda.synchronize()
print('ms/step', (time.perf_counter() - t0) / 5 * 1e3)
