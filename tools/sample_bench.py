#!/usr/bin/env python3
"""Sampling latency of the eval-mode StyleGAN-1024 generator (EWMA weights would be the same network): eager launches
through ctypes vs replay of the captured hipGraph (gan_lab_amd/graphs.py), batch 1 / 4 / 16."""
import contextlib
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from gan_lab_amd.graphs import GraphedGenerator  # noqa: E402


def per_call(fn, n=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    torch.cuda.set_device(0)
    res = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    with contextlib.redirect_stdout(io.StringIO()):
        L = bench.build_learner(res, 4, 'cuda')
    g = L.gen_model
    with torch.no_grad():
        g.train()
        g(torch.randn(4, 512, device='cuda'))     # one training-mode forward: initialises w_ewma (truncation trick)
    g.eval()
    for b in (1, 4, 16):
        gg = GraphedGenerator(g, b)
        gf = GraphedGenerator(g, b, follow_weight_updates=False)
        z = torch.randn(b, 512, device='cuda')
        with torch.no_grad():
            eager = per_call(lambda: g(z, noise=gg.noise))
        graph = per_call(lambda: gg(z, redraw_noise=False))
        frozen = per_call(lambda: gf(z, redraw_noise=False))
        frozen_noise = per_call(lambda: gf(z))
        print(json.dumps({'what': f'StyleGAN-{res} eval-mode generator forward', 'batch': b,
                          'eager_ms': round(eager, 3), 'hipgraph_replay_ms': round(graph, 3),
                          'hipgraph_replay_frozen_weights_ms': round(frozen, 3),
                          'hipgraph_frozen_with_fresh_noise_ms': round(frozen_noise, 3),
                          'images_per_sec_graph_frozen': round(b / frozen * 1e3, 1)}), flush=True)
        del gg, gf


if __name__ == '__main__':
    main()
