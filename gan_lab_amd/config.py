#!/usr/bin/env python3
"""Model configuration CLI / factory (drop-in surface of gan_lab/config.py).

``python -m gan_lab_amd.config stylegan --loss=nonsaturating --gradient_penalty=r1 ...`` parses the
same arguments with the same defaults as the reference (config.py:81-325), post-processes them the
same way (:337-376: torch.device, seeding, ``bs_dict`` rebuilt from ``--batch_size`` with 512 ->
bs//2 and 1024 -> bs//4) and pickles the Namespace to ``<this dir>/.config.p`` with the pointer file
``~/.configs_dir.txt`` (:408-412), which ``get_current_configuration('config')`` reads back.
``make_config(model, **overrides)`` builds the same Namespace in-process (tests, bench.py).
New, optional fields (ignored by the reference): ``log_every``, ``compute_dtype`` ('f32' = the reference's
arithmetic, 'bf16' = BASELINE config #2: bf16-compute 3x3 convolutions with fp32 storage / masters).
"""
import argparse
import os
import pickle
from pathlib import Path

import numpy as np
import torch

from ._int import str2bool

BS = 64
NIMG_TRANSITION = 600000
_HERE = os.path.abspath(os.path.dirname(__file__))
_MODELS = {'resnetgan': 'ResNet GAN', 'resnet gan': 'ResNet GAN', 'progan': 'ProGAN', 'stylegan': 'StyleGAN'}


def _spec(model_type):
    """(name, type, default[, choices]) rows; booleans use str2bool like the reference."""
    dev = 'cuda' if torch.cuda.is_available() else 'cpu'
    rows = [
        ('dev', str.casefold, dev), ('n_gpu', int, 1), ('enable_cudnn_autotuner', bool, False),
        ('random_seed', int, -1), ('gen_bs_mult', int, 1), ('num_gen_iters', int, 1),
        ('loss', str.casefold, 'wgan'), ('gradient_penalty', str.casefold, 'wgan-gp'), ('lda', float, 10.),
        ('gamma', float, 1.), ('lr_sched_custom', str.casefold, None), ('optimizer', str.casefold, 'adam'),
        ('beta1', float, 0.), ('eps', float, 1.e-8), ('wd', float, 0.), ('align_corners', bool, False),
        ('model_upsample_type', str.casefold, 'nearest'), ('model_downsample_type', str.casefold, 'average'),
        ('latent_distribution', str.casefold, 'normal'), ('num_classes', int, 0), ('class_condition', bool, False),
        ('use_auxiliary_classifier', bool, False), ('ac_disc_scale', float, 1.), ('ac_gen_scale', float, .1),
        ('num_iters_valid', int, 1000), ('metrics_dev', str.casefold, 'cpu'),
        ('gen_metrics', list, ['generator loss', 'fake realness', 'image grid']),
        ('disc_metrics', list, ['discriminator loss', 'fake realness', 'real realness']),
        ('img_grid_sz', int, 4), ('img_grid_show_labels', bool, True),
        ('save_samples_dir', Path, Path(_HERE + '/samples/')), ('num_iters_save_model', int, 1000),
        ('save_model_dir', Path, Path(_HERE + '/models/')), ('num_workers', int, 0),
        ('pin_memory', bool, dev == 'cuda'), ('log_every', int, 50), ('compute_dtype', str.casefold, 'f32'),
    ]
    if model_type == 'ResNet GAN':
        rows += [('batch_size', int, BS), ('num_main_iters', int, 300000), ('num_disc_iters', int, 5),
                 ('lr_base', float, .0001), ('lr_sched', str.casefold, None), ('beta2', float, .9),
                 ('res_samples', int, 64), ('res_dataset', int, 64), ('blur_type', str.casefold, None),
                 ('eps_drift', float, 0.), ('len_latent', int, 128), ('nonlinearity', str.casefold, 'relu'),
                 ('leakiness', float, .01), ('use_equalized_lr', bool, False)]
    else:
        rows += [('batch_size', int, BS),
                 ('bs_dict', dict, {4: BS, 8: BS, 16: BS, 32: BS, 64: BS, 128: BS, 256: BS, 512: BS // 2,
                                    1024: BS // 4}),
                 ('num_disc_iters', int, 1), ('nimg_transition', int, NIMG_TRANSITION), ('lr_base', float, .001),
                 ('lr_sched', str.casefold, 'resolution dependent'), ('beta2', float, .99),
                 ('res_samples', int, 1024), ('res_dataset', int, 1024), ('blur_type', str.casefold, 'binomial'),
                 ('bit_exact_resampling', bool, False), ('eps_drift', float, .001), ('len_latent', int, 512),
                 ('nonlinearity', str.casefold, 'leaky relu'), ('leakiness', float, .2),
                 ('use_equalized_lr', bool, True), ('normalize_z', bool, True), ('mbstd_group_size', int, 4),
                 ('use_ewma_gen', bool, True)]
        if model_type == 'ProGAN':
            rows += [('num_main_iters', int, (NIMG_TRANSITION // BS) * 20),
                     ('lr_fctr_dict', dict, {4: 1., 8: 1., 16: 1., 32: 1., 64: 1., 128: 1., 256: 1., 512: 1.,
                                             1024: 1.5}),
                     ('init_res', int, 4), ('use_pixelnorm', bool, True)]
        else:
            rows += [('num_main_iters', int, (NIMG_TRANSITION // BS) * 18),
                     ('lr_fctr_dict', dict, {4: 1., 8: 1., 16: 1., 32: 1., 64: 1., 128: 1.5, 256: 2., 512: 3.,
                                             1024: 3.}),
                     ('init_res', int, 8), ('len_dlatent', int, 512), ('mapping_num_fcs', int, 8),
                     ('mapping_lrmul', float, .01), ('use_noise', bool, True), ('use_pixelnorm', bool, False),
                     ('use_instancenorm', bool, True), ('pct_mixing_reg', float, .9),
                     ('beta_trunc_trick', float, .995), ('psi_trunc_trick', float, .7),
                     ('cutoff_trunc_trick', int, 4)]
    return rows


def _model_type(name):
    key = name.casefold().replace('-', '').replace('_', '')
    if key not in _MODELS:
        raise ValueError("Invalid model inputted. Currently supported models: resnetgan, progan, stylegan")
    return _MODELS[key]


def _postprocess(config, model_type, seed=True):
    """config.py:337-376."""
    config.dev = torch.device(config.dev)
    if config.pin_memory and config.dev != torch.device('cuda'):
        raise ValueError('--pin_memory should be set to `False` if not using CUDA.')
    if seed:
        if config.random_seed == -1:
            np.random.seed(None)
            torch.seed()
        elif 0 <= config.random_seed < 2 ** 32:
            np.random.seed(config.random_seed)
            torch.manual_seed(config.random_seed)
        else:
            raise ValueError("--random_seed must either be -1 for random seeding or be in the range [0,2**32) to "
                             "accommodate numpy's and torch's seeding specifications.")
    if model_type in ('ProGAN', 'StyleGAN',):
        _bs = config.batch_size
        config.bs_dict = {4: _bs, 8: _bs, 16: _bs, 32: _bs, 64: _bs, 128: _bs, 256: _bs, 512: _bs // 2,
                          1024: _bs // 4}
        if config.mbstd_group_size < -1 or not config.mbstd_group_size:
            raise ValueError("--mbstd_group_size must either be -1 to indicate not applying minibatch standard "
                             "deviation or a positive integer indicating the group size for the minibatch standard "
                             "deviation layer.")
    config.model = model_type
    return config


def make_config(model, seed=False, **overrides):
    """Namespace with the reference defaults for `model` ('stylegan' | 'progan'), then `overrides`,
    then the reference post-processing.  ``bs_dict`` may be overridden explicitly AFTER the rebuild
    (the reference only allows that through ``learner.config.bs_dict``)."""
    mt = _model_type(model)
    ns = argparse.Namespace(**{name: default for name, _, default in _spec(mt)})
    bs_dict = overrides.pop('bs_dict', None)
    for k, v in overrides.items():
        if not hasattr(ns, k):
            raise AttributeError(f'unknown config field {k!r} for {mt}')
        setattr(ns, k, v)
    ns = _postprocess(ns, mt, seed=seed)
    if bs_dict is not None:
        ns.bs_dict = dict(bs_dict)
    return ns


def main(argv=None):
    top = argparse.ArgumentParser(description='Configure a GAN model (gan-lab compatible).')
    top.add_argument('model', type=str)
    first, rest = top.parse_known_args(argv)
    mt = _model_type(first.model)
    parser = argparse.ArgumentParser()
    parser.add_argument('model', type=str)
    for name, typ, default in _spec(mt):
        if typ is bool:
            parser.add_argument('--' + name, type=str2bool, nargs='?', const=True, default=default)
        elif typ in (list, dict):
            parser.add_argument('--' + name, type=typ, default=default)
        else:
            parser.add_argument('--' + name, type=typ, default=default)
    config = parser.parse_args(argv)
    config = _postprocess(config, mt, seed=True)
    config.save_samples_dir.mkdir(parents=True, exist_ok=True)
    config.save_model_dir.mkdir(parents=True, exist_ok=True)
    with open(str(Path.home() / '.configs_dir.txt'), 'wb') as f:
        f.write(_HERE.encode())
    with open(_HERE + '/.config.p', 'wb') as f:
        pickle.dump(config, f, protocol=3)
    return config


if __name__ == '__main__':
    main()
