#!/usr/bin/env python3
"""Throughput of the OTHER BASELINE.json configurations through the learners' own ``train()`` (bench.py measures the
headline config #3).  One JSON line per configuration; these are reported lines, not the judged metric:

  #2  StyleGAN res_samples=128, bs 8, bf16 compute / fp32 master, nonsaturating + R1, stabilised phase
  #4  ProGAN res_samples=256, the FULL 4 -> 256 fade-in schedule (nimg_transition shortened, stated), WGAN + WGAN-GP
  #5  ResNet GAN 64x64, bs 64, WGAN-GP, num_disc_iters=5

    python tools/bench_configs.py [2] [4] [5]
"""
import contextlib
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from gan_lab_amd.config import make_config  # noqa: E402
from gan_lab_amd.utils.data_utils import SyntheticImageLoader  # noqa: E402


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def timed_train(L, dl, iters):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    quiet(L.train, dl, num_main_iters=iters)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def config2():
    from gan_lab_amd.stylegan.learner import StyleGANLearner
    bs = {r: 8 for r in (4, 8, 16, 32, 64, 128, 256, 512, 1024)}
    out = {}
    for dt in ('bf16', 'f32'):
        cfg = make_config('stylegan', dev='cuda', pin_memory=False, loss='nonsaturating', gradient_penalty='r1',
                          res_samples=128, res_dataset=128, init_res=128, batch_size=8, bs_dict=bs,
                          num_iters_save_model=10 ** 9, log_every=0, compute_dtype=dt)
        L = quiet(StyleGANLearner, cfg)
        dl = SyntheticImageLoader(1 << 20, 8, 128, device='cuda')
        timed_train(L, dl, 5)
        n = 40
        dt_s = timed_train(L, dl, n)
        out[dt] = round(8 * n / dt_s, 2)
        del L
        torch.cuda.empty_cache()
    return {'config': '#2 StyleGAN res_samples=128 bs=8, stabilised phase, nonsaturating + R1, learner.train()',
            'images_per_sec': out['bf16'], 'dtype': 'bf16 compute / fp32 storage+master',
            'images_per_sec_f32': out['f32']}


def config4():
    from gan_lab_amd.progan.learner import ProGANLearner
    bsz, nimg = 32, 4096
    cfg = make_config('progan', dev='cuda', pin_memory=False, res_samples=256, res_dataset=256, init_res=4,
                      batch_size=bsz, nimg_transition=nimg, num_iters_save_model=10 ** 9, log_every=0)
    L = quiet(ProGANLearner, cfg)
    dl = SyntheticImageLoader(1 << 20, bsz, 4, device='cuda')
    # 13 phases (4 stab, then fade + stab for 8..256), nimg/bsz iterations each, + a tail in the final phase
    iters = 13 * (nimg // bsz) + 16
    dt_s = timed_train(L, dl, iters)
    assert L.gen_model.curr_res == 256 and not L.gen_model.fade_in_phase, (L.gen_model.curr_res, L.gen_model.alpha)
    tail = 24
    dt_tail = timed_train(L, dl, tail)
    return {'config': f'#4 ProGAN res_samples=256, full 4->256 schedule (nimg_transition={nimg}, bs={bsz} at every '
                      f'resolution), WGAN + WGAN-GP + drift, learner.train()',
            'iterations': iters, 'schedule_seconds': round(dt_s, 2),
            'images_per_sec_whole_schedule': round(bsz * iters / dt_s, 2),
            'images_per_sec_at_256_stabilised': round(bsz * tail / dt_tail, 2), 'dtype': 'f32',
            'final_res': L.gen_model.curr_res, 'loss_d': L.last_losses.get('loss_d'), 'loss_g': L.last_losses.get('loss_g')}


def config5():
    from gan_lab_amd.resnetgan.learner import GANLearner
    cfg = make_config('resnetgan', dev='cuda', pin_memory=False, batch_size=64, res_samples=64, res_dataset=64,
                      num_iters_save_model=10 ** 9, log_every=0)
    L = quiet(GANLearner, cfg)
    dl = SyntheticImageLoader(1 << 20, 64, 64, device='cuda')
    timed_train(L, dl, 3)
    n = 20
    dt_s = timed_train(L, dl, n)
    nd = cfg.num_disc_iters
    return {'config': f'#5 ResNet GAN 64x64 bs=64, WGAN + WGAN-GP, num_disc_iters={nd}, learner.train()',
            'main_iters_per_sec': round(n / dt_s, 3), 'real_images_per_sec': round(64 * nd * n / dt_s, 1),
            'ms_per_main_iter': round(dt_s / n * 1e3, 2), 'dtype': 'f32'}


def main():
    torch.cuda.set_device(0)
    which = [a for a in sys.argv[1:] if a in ('2', '4', '5')] or ['2', '4', '5']
    for w in which:
        r = {'2': config2, '4': config4, '5': config5}[w]()
        r['n_gpus'] = 1
        r['data'] = 'synthetic, resident on the device'
        print(json.dumps(r), flush=True)


if __name__ == '__main__':
    main()
