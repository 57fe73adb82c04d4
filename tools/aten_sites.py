#!/usr/bin/env python3
"""Which ATen operators still launch inside a step, and from where: counts every dispatched aten op that touches a GPU
tensor during one step of a BASELINE configuration, keyed by the innermost gan_lab_amd frame ("autograd engine" when the
engine itself adds gradients).  Usage: python tools/aten_sites.py [--config 5] [--top 40]"""
import argparse
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GANLAB_STEP_GRAPH', '0')      # count the eager step (a replayed graph dispatches nothing)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from torch.utils._pytree import tree_flatten

import bench

SKIP = ('aten.view', 'aten._unsafe_view', 'aten.detach', 'aten.alias', 'aten.empty', 'aten.as_strided', 'aten.slice',
        'aten.select', 'aten.t.', 'aten.transpose', 'aten.expand', 'aten.reshape', 'aten.unsqueeze', 'aten.squeeze',
        'aten.permute', 'aten.empty_like', 'aten.new_empty', 'aten.lift_fresh', 'aten._local_scalar_dense', 'aten.is_',
        'aten.unbind', 'aten.split', 'aten.narrow', 'aten.empty_strided', 'aten.set_', 'aten.record_stream')


class Count(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.sites = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if not name.startswith(SKIP):
            flat, _ = tree_flatten((args, out))
            if any(isinstance(v, torch.Tensor) and v.is_cuda for v in flat):
                site = 'autograd engine'
                for fr in reversed(traceback.extract_stack(limit=40)):
                    if 'gan_lab_amd' in fr.filename:
                        site = f'{os.path.relpath(fr.filename)}:{fr.lineno} {fr.name}'
                        break
                self.sites[(name, site)] += 1
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', type=int, default=5)
    ap.add_argument('--top', type=int, default=50)
    o = ap.parse_args()
    sys.argv = [sys.argv[0], '--config', str(o.config), '--no-cpu-baseline']
    a = bench.parse()
    a.world = 1
    w = bench.Workload(a, torch)
    for _ in range(2):
        w.step()
    torch.cuda.synchronize()
    with Count() as c:
        w.step()
    torch.cuda.synchronize()
    tot = sum(c.sites.values())
    print(f'config {o.config}: {tot} ATen operator calls on GPU tensors in one step')
    for (name, site), n in c.sites.most_common(o.top):
        print(f'{n:6d}  {name:34s} {site}')


if __name__ == '__main__':
    main()
