#!/usr/bin/env python3
"""Host-side cost of a training step (the launch-bound configurations): cProfile over a few steps, GPU left to run behind.
    python tools/cpu_profile_step.py [--res 128 --batch 8 --dtype bf16] [--steps 10]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--res', type=int, default=128)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--model', default='stylegan')
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--top', type=int, default=45)
    a = ap.parse_args()
    learner = bench.build_learner(a.res, a.batch, 'cuda', a.dtype, a.model)
    sched = torch.rand(a.batch, 3, a.res, a.res, device='cuda') * 2 - 1
    for _ in range(3):
        bench.one_step(learner, sched)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        bench.one_step(learner, sched)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f'host issue time {t_issue / a.steps * 1e3:.2f} ms/step, with the GPU drained {t_all / a.steps * 1e3:.2f} ms/step')
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(a.steps):
        bench.one_step(learner, sched)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats('tottime').print_stats(a.top)


if __name__ == '__main__':
    main()
