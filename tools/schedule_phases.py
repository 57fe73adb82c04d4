#!/usr/bin/env python3
"""Per-phase wall time of config #4's growth schedule (ProGAN 4 -> res, bs 32, nimg_transition images per phase):
    python tools/schedule_phases.py [res] [nimg_transition] [step_graph: auto|0|1]
prints one line per phase: resolution, fade-in or stabilised, iterations, seconds, ms per iteration."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    res = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    nimg = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    sg = sys.argv[3] if len(sys.argv) > 3 else 'auto'
    if sg != 'auto':
        os.environ['GANLAB_STEP_GRAPH'] = sg
    import torch
    import bench
    from gan_lab_amd.utils.data_utils import SyntheticImageLoader
    b = 32
    L = bench.build_learner(res, b, 'cuda', 'f32', 'progan', init_res=4, nimg_transition=nimg)
    dl = SyntheticImageLoader(1 << 22, b, 4, device='cuda')
    n_phases = 1 + 2 * (len(bin(res)) - len(bin(4)))
    iters = n_phases * ((nimg + b - 1) // b) + 8
    marks = []
    orig = L._apply_phase_events
    state = {'it': 0}

    def hooked(sched, *a, **k):
        before = (L.gen_model.curr_res, bool(L.gen_model.fade_in_phase))
        orig(sched, *a, **k)
        after = (L.gen_model.curr_res, bool(L.gen_model.fade_in_phase))
        if after != before or not marks:
            torch.cuda.synchronize()
            marks.append((time.perf_counter(), state['it'], after))
        state['it'] += 1
    L._apply_phase_events = hooked
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bench._quiet(L.train, dl, num_main_iters=iters)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    marks.append((t1, state['it'], None))
    print(f'schedule 4 -> {res}, nimg_transition {nimg}, step graph {sg}: {t1 - t0:.2f} s, {iters} iterations')
    for (ta, ia, ph), (tb, ib, _) in zip(marks[:-1], marks[1:]):
        print(f'  {ph[0]:4d}^2 {"fade-in   " if ph[1] else "stabilised"} {ib - ia:5d} iterations {tb - ta:7.2f} s '
              f'{(tb - ta) / max(ib - ia, 1) * 1e3:8.2f} ms / iteration')


if __name__ == '__main__':
    main()
