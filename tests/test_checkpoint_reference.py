"""Checkpoint wire-format compatibility, reverse direction: a file written by
``save_model(..., reference_format=True)`` of THIS package is read by the REFERENCE's own ``load_model``
(gan_lab/progan/learner.py:1305-1448, gan_lab/stylegan/learner.py:503-640), imported here from /root/reference.
Host logic only (``GANLAB_HOST_LOGIC_ONLY``: CPU tensors, no kernels).  Skipped where the reference is absent (the
GPU box): the forward direction (reference-written files -> this package) is covered there by committed fixtures."""
import contextlib
import functools
import io
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
pytestmark = pytest.mark.skipif(not os.path.isdir('/root/reference/gan_lab'),
                                reason='needs the reference checkout (build container only)')


@pytest.fixture(scope='module')
def ref():
    """tests/golden/make_golden.py with the reference imported through its stubs - and everything it put on
    ``sys.path`` / into ``sys.modules`` (the reference's top-level ``_int``, ``utils``, ``progan`` ... packages, the
    ``torchvision`` / ``indexed`` stand-ins) taken out again afterwards, so later tests see a clean interpreter."""
    path0, mods0 = list(sys.path), set(sys.modules)
    sys.path.insert(0, GOLDEN)
    with contextlib.redirect_stdout(io.StringIO()):
        import make_golden as MG
    yield MG
    sys.path[:] = path0
    for name in set(sys.modules) - mods0:
        f = getattr(sys.modules[name], '__file__', None) or ''
        if '/root/reference' in f or f.startswith(GOLDEN) or name.split('.')[0] in ('torchvision', 'indexed'):
            del sys.modules[name]


def _product_learner(kind, monkeypatch, tmp_path):
    monkeypatch.setenv('GANLAB_HOST_LOGIC_ONLY', '1')
    from gan_lab_amd import progressive as P
    from gan_lab_amd.config import make_config
    from gan_lab_amd.progan.learner import ProGANLearner
    from gan_lab_amd.stylegan.learner import StyleGANLearner
    P.FMAP_BASE, P.FMAP_MAX = 64, 16
    bs = 4
    common = dict(dev='cpu', pin_memory=False, res_samples=16, res_dataset=16, init_res=4, batch_size=bs, len_latent=16,
                  nimg_transition=22, log_every=0, save_model_dir=tmp_path, save_samples_dir=tmp_path,
                  bs_dict={4: bs, 8: bs, 16: bs // 2, 32: bs, 64: bs, 128: bs, 256: bs, 512: bs // 2, 1024: bs // 4})
    torch.manual_seed(3)
    with contextlib.redirect_stdout(io.StringIO()):
        if kind == 'stylegan':
            L = StyleGANLearner(make_config('stylegan', len_dlatent=16, mapping_num_fcs=2, cutoff_trunc_trick=1,
                                            beta_trunc_trick=.9, loss='nonsaturating', gradient_penalty='r1', **common))
        else:
            L = ProGANLearner(make_config('progan', **common))
        # one growth event: 8x8, mid fade-in, prev_torgb / prev_fromrgb in the optimiser sets
        L._grow()
    L.gen_model.alpha = 0.375
    gen = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for arena in (L.arena_g, L.arena_d):
            arena.flat.copy_(torch.randn(arena.flat.shape, generator=gen) * 0.3)
        L.ewma.flat.copy_(L.arena_g.flat * 0.9 + 0.01)
    # Adam moments as after 5 steps
    for opt, net in ((L.opt_gen, L.gen_model), (L.opt_disc, L.disc_model)):
        named = list(net.named_parameters())
        opt.import_moments(named, {'step': 5,
                                   'exp_avg': {k: torch.randn(p.shape, generator=gen) * 1e-2 for k, p in named},
                                   'exp_avg_sq': {k: torch.rand(p.shape, generator=gen) * 1e-3 for k, p in named}})
    L.not_trained_yet = False
    L.curr_img_num, L.curr_phase_num, L.curr_dataset_batch_num = 44, 1, 11
    from gan_lab_amd.schedule import PhaseSchedule
    L.sched = PhaseSchedule(4, 16, L.config.bs_dict, 22, 1).restore(8, 44, 1, [24, 24], 0.375)
    L.ds_mean = torch.tensor([.4, .5, .6]).view(3, 1, 1)
    L.ds_std = torch.tensor([.2, .25, .3]).view(3, 1, 1)
    L.valid_z = torch.randn(16, 16, generator=gen)
    if kind == 'stylegan':
        L.gen_model.w_ewma = torch.randn(16, generator=gen)
    return L


@pytest.mark.parametrize('kind', ['progan', 'stylegan'])
def test_reference_loads_a_checkpoint_written_in_reference_format(kind, ref, monkeypatch, tmp_path):
    MG = ref
    ns = MG.ns
    L = _product_learner(kind, monkeypatch, tmp_path)
    path = str(tmp_path / f'{kind}_model.tar')
    L.save_model(path, reference_format=True)
    from gan_lab_amd import checkpoint as ckpt
    assert ckpt.is_reference_format(ckpt.load_checkpoint(path))
    # ---- the reference's own learner reads it ----
    with contextlib.redirect_stdout(io.StringIO()):
        if kind == 'stylegan':
            cfg, _ = MG._ref_stylegan_setup(num_main_iters=3)
            R = ns.sl.StyleGANLearner(cfg)
        else:
            cfg, _ = MG._ref_progan_setup(num_main_iters=3)
            R = ns.pl.ProGANLearner(cfg)
        orig = torch.load
        monkeypatch.setattr(torch, 'load', functools.partial(orig, weights_only=False))   # torch >= 2.6 default
        R.load_model(path)
        monkeypatch.setattr(torch, 'load', orig)
    assert R.gen_model.curr_res == 8 and R.gen_model.fade_in_phase and abs(R.gen_model.alpha - 0.375) < 1e-12
    for mine, theirs in ((L.gen_model, R.gen_model), (L.disc_model, R.disc_model)):
        sd_a, sd_b = mine.state_dict(), theirs.state_dict()
        assert list(sd_a.keys()) == list(sd_b.keys())
        for k in sd_a:
            assert torch.equal(sd_a[k].cpu(), sd_b[k].cpu()), k
    # EWMA generator: module weights and the lagged_params dict (IndexedOrderedDict on the reference side)
    assert list(R.lagged_params.keys()) == list(L.lagged_params.keys())
    for k, v in L.lagged_params.items():
        assert torch.equal(R.lagged_params[k], v.cpu()), k
    lag_sd = R.gen_model_lagged.state_dict()
    for k, v in L.lagged_params.items():
        assert torch.equal(lag_sd[k].cpu(), v.cpu()), k
    # torch Adam state: step / exp_avg / exp_avg_sq per parameter, in most_parameters order
    for mine, theirs, net in ((L.opt_gen, R.opt_gen, L.gen_model), (L.opt_disc, R.opt_disc, L.disc_model)):
        mom = mine.export_moments(net.named_parameters())
        st = theirs.state_dict()
        names = [k for k, _ in net.named_parameters()]
        assert len(st['param_groups'][0]['params']) == len(names) == len(st['state'])
        for i, k in enumerate(names):
            assert float(st['state'][i]['step']) == 5.0
            assert torch.equal(st['state'][i]['exp_avg'], mom['exp_avg'][k]), k
            assert torch.equal(st['state'][i]['exp_avg_sq'], mom['exp_avg_sq'][k]), k
        assert st['param_groups'][0]['betas'] == (0.0, 0.99) and st['param_groups'][0]['eps'] == 1e-8
    assert (R.curr_img_num, R.curr_phase_num, R.batch_size) == (44, 1, L.batch_size)
    assert R.nimg_transition_lst == [24, 24] and R.progressively_grow and not R.not_trained_yet
    assert torch.equal(R.ds_mean, L.ds_mean) and torch.equal(R.valid_z, L.valid_z)
    assert R.config.res_samples == 16 and R.config.model == L.config.model
    assert isinstance(R.nl, torch.nn.LeakyReLU) and R.nl.negative_slope == 0.2
    if kind == 'stylegan':
        g = R.gen_model
        assert g.use_truncation_trick and g.trunc_cutoff_stage == 1 and g.w_ewma_beta == .9 and g.w_eval_psi == .7
        # the reference's quirk: after load the GENERATOR holds checkpoint['w_ewma_lagged'], the EWMA copy 'w_ewma'
        assert torch.equal(g.w_ewma, L.gen_model.w_ewma) and torch.equal(R.gen_model_lagged.w_ewma, L.gen_model.w_ewma)
    # and the reference can run its own generator on the loaded weights (CPU)
    R.gen_model.eval()
    with torch.no_grad():
        img = R.gen_model(torch.randn(2, 16))
    assert img.shape == (2, 3, 8, 8) and torch.isfinite(img).all()
    # ---- round trip back into this package ----
    L2 = _product_learner(kind, monkeypatch, tmp_path)
    with contextlib.redirect_stdout(io.StringIO()):
        L2.load_model(path)
    for a, b in ((L.arena_g, L2.arena_g), (L.arena_d, L2.arena_d)):      # (padding floats between parameters excluded)
        for (k, va), (_, vb) in zip(a.views_of(a.flat).items(), b.views_of(b.flat).items()):
            assert torch.equal(va, vb), k
    for k, v in L.lagged_params.items():
        assert torch.equal(L2.lagged_params[k], v), k
    assert torch.equal(L2.ds_std, L.ds_std)
    assert L2.sched.curr_img_num == 44 and L2.gen_model.alpha == 0.375
    from gan_lab_amd import progressive as P
    P.FMAP_BASE, P.FMAP_MAX = 8192, 512
