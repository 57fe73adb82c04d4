#!/usr/bin/env python3
"""Stabilised phase of a progressive network at a given resolution (batch 32): host issue time vs wall time per step; run
under rocprofv3 --kernel-trace for the per-kernel picture of the low-resolution phases of a schedule (BASELINE config #4).
    python tools/profile_phase.py 32 [steps] [progan|stylegan]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
res = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
model = sys.argv[3] if len(sys.argv) > 3 else 'progan'
L = bench.build_learner(res, 32, 'cuda', 'f32', model)
real = torch.rand(32, 3, res, res, device='cuda') * 2 - 1
for _ in range(5):
    bench.one_step(L, real)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    bench.one_step(L, real)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'{model} res {res}: host issue {(t1 - t0) / steps * 1e3:.2f} ms/step, wall {(t2 - t0) / steps * 1e3:.2f} ms/step')
