#!/usr/bin/env python3
"""SURVEY.md §8f item 4 (class-conditional / auxiliary-classifier variants): does ANY of them run in the reference?

Runs the REAL reference learners (imported from /root/reference through tests/golden/_refstub.py, like
make_golden.py) with ``num_classes=3`` and each of {class_condition, use_auxiliary_classifier, both}, lets each one
construct and take up to three main iterations of its own ``train()`` on labelled random images, and records what
happened - the exception type, message and the innermost reference frame - in ``tests/golden/conditional_probe.json``
(data only).  ``tests/test_host_logic.py::test_conditional_variants_row_is_closed_by_the_probe`` reads that file.

    python tests/golden/probe_conditional.py
"""
import json
import os
import sys
import traceback
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (imports the reference through the stubs)

import numpy as np  # noqa: E402
import torch  # noqa: E402

ns = MG.ns
NUM_CLASSES = 3


class LabelledDL(MG._FakeDL):
    """The fake loader of make_golden.py with labels in [0, NUM_CLASSES)."""

    def __iter__(self):
        for xb, _ in super().__iter__():
            yield xb, torch.randint(0, NUM_CLASSES, (xb.shape[0],), generator=self.gen)


def innermost_reference_frame(tb):
    frames = [f for f in traceback.extract_tb(tb) if '/root/reference/' in f.filename]
    if not frames:
        return None
    f = frames[-1]
    return f'{os.path.relpath(f.filename, "/root/reference/gan_lab")}:{f.lineno} ({f.name})'


def attempt(name, build, run):
    rec = {'variant': name}
    try:
        learner = build()
        rec['constructed'] = True
    except Exception as e:  # noqa: BLE001 - the probe records whatever the reference raises
        rec.update(constructed=False, stage='constructor', exception=type(e).__name__, message=str(e)[:300],
                   where=innermost_reference_frame(e.__traceback__))
        return rec
    try:
        run(learner)
        rec.update(ran=True, stage='train', exception=None)
    except Exception as e:  # noqa: BLE001
        rec.update(ran=False, stage='train', exception=type(e).__name__, message=str(e)[:300],
                   where=innermost_reference_frame(e.__traceback__))
    return rec


def progressive_variants(model, sweep=((True, False, None, None), (False, True, None, None), (True, True, None, None))):
    """``sweep`` rows: (class_condition, use_auxiliary_classifier, loss override, gradient-penalty override)."""
    import torchvision.transforms as tvt
    out = []
    for cc, ac, loss, gp in sweep:
        def build(cc=cc, ac=ac, loss=loss, gp=gp):
            cfg, bs = MG._ref_progan_setup(num_main_iters=3)
            cfg.num_classes, cfg.class_condition, cfg.use_auxiliary_classifier = NUM_CLASSES, cc, ac
            if loss is not None:
                cfg.loss, cfg.gradient_penalty = loss, gp
            if model == 'StyleGAN':
                cfg.model = 'StyleGAN'
                ns.sb.FMAP_BASE, ns.sb.FMAP_MAX = MG.FMAP_BASE, MG.FMAP_MAX
                for k, v in dict(init_res=4, len_dlatent=MG.LEN_LATENT, mapping_num_fcs=MG.NUM_FCS, mapping_lrmul=.01,
                                 use_noise=True, use_pixelnorm=False, use_instancenorm=True, pct_mixing_reg=.9,
                                 beta_trunc_trick=.995, psi_trunc_trick=.7, cutoff_trunc_trick=None,
                                 loss='nonsaturating', gradient_penalty='r1').items():
                    setattr(cfg, k, v)
            import pickle
            with open(os.path.join(os.environ['HOME'], '.config.p'), 'wb') as f:
                pickle.dump(cfg, f)
            torch.manual_seed(0)
            np.random.seed(0)
            L = (ns.sl.StyleGANLearner if model == 'StyleGAN' else ns.pl.ProGANLearner)(cfg)
            L._probe_bs = bs
            return L

        def run(L):
            gen = torch.Generator().manual_seed(7)
            L.train(LabelledDL(64, L._probe_bs, 4, tvt.Resize, gen), num_main_iters=3)
        extra = '' if loss is None else f', loss={loss}, gradient_penalty={gp}'
        out.append(attempt(f'{model}: class_condition={cc}, use_auxiliary_classifier={ac}{extra}', build, run))
    return out


def resnet_variants():
    import argparse as ap
    import pickle
    import tempfile
    from pathlib import Path
    from PIL import Image
    out = []
    for cc, ac in ((True, False), (False, True), (True, True)):
        def build(cc=cc, ac=ac):
            tmp = tempfile.mkdtemp(prefix='ganlab_probe_')
            os.environ['HOME'] = tmp
            cfg = ap.Namespace(
                model='ResNet GAN', dev=torch.device('cpu'), n_gpu=1, enable_cudnn_autotuner=False, random_seed=0,
                gen_bs_mult=1, num_gen_iters=1, num_disc_iters=1, loss='wgan', gradient_penalty='wgan-gp', lda=10.,
                gamma=1., eps_drift=0., optimizer='adam', lr_base=1e-4, beta1=0., beta2=.9, eps=1e-8, wd=0.,
                lr_sched=None, lr_sched_custom=None, batch_size=4, num_main_iters=2, res_samples=32, res_dataset=32,
                model_upsample_type='nearest', model_downsample_type='average', align_corners=False, blur_type=None,
                nonlinearity='relu', leakiness=.01, use_equalized_lr=False, len_latent=16,
                latent_distribution='normal', num_classes=NUM_CLASSES, class_condition=cc,
                use_auxiliary_classifier=ac, ac_disc_scale=1., ac_gen_scale=.1, num_iters_valid=1000,
                metrics_dev=torch.device('cpu'), gen_metrics=[], disc_metrics=[], img_grid_sz=4,
                img_grid_show_labels=True, save_samples_dir=Path(tmp) / 'samples', save_model_dir=Path(tmp) / 'models',
                num_iters_save_model=10 ** 9, num_workers=0, pin_memory=False)
            dcfg = ap.Namespace(dataset='custom', dataset_dir=tmp, ds_mean=[.5, .5, .5], ds_std=[.5, .5, .5],
                                dataset_downsample_type=Image.BOX, include_valid_set=False)
            for fn, obj in (('.configs_dir.txt', None), ('.config.p', cfg), ('.data_config.p', dcfg)):
                with open(os.path.join(tmp, fn), 'wb') as f:
                    f.write(tmp.encode()) if obj is None else pickle.dump(obj, f)
            torch.manual_seed(0)
            return ns.rl.GANLearner(cfg)

        def run(L):
            gen = torch.Generator().manual_seed(7)
            import torchvision.transforms as tvt
            L.train(LabelledDL(64, 4, 32, tvt.Resize, gen), num_main_iters=2)
        out.append(attempt(f'ResNet GAN: class_condition={cc}, use_auxiliary_classifier={ac}', build, run))
    return out


def main():
    import contextlib
    import io
    recs = []
    # class conditioning alone fails inside the WGAN-GP penalty (the default): try it without / with another penalty too
    sweep_cc = ((True, False, 'wgan', None), (True, False, 'nonsaturating', None), (True, False, 'nonsaturating', 'r1'))
    for fn in (lambda: progressive_variants('ProGAN'), lambda: progressive_variants('ProGAN', sweep_cc),
               lambda: progressive_variants('StyleGAN'), resnet_variants):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            recs += fn()
    out = {'reference': 'sidward14/gan-lab v0.4.2 (/root/reference), torch ' + torch.__version__ + ' CPU',
           'num_classes': NUM_CLASSES, 'variants': recs,
           'any_variant_runs': any(r.get('ran') for r in recs)}
    path = os.path.join(HERE, 'conditional_probe.json')
    with open(path, 'w') as f:
        json.dump(out, f, indent=1)
    for r in recs:
        print(r)
    print('wrote', path)


if __name__ == '__main__':
    main()
