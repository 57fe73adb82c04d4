"""GPU tests of the learner: a D-iteration + G-iteration (fused Adam, EWMA) against training-step
vectors produced by the reference modules + torch.optim.Adam, and the full train() loop over a
growth schedule."""
import numpy as np
import pytest
import torch

from util import assert_close, load_golden, sub, t

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _widths():
    from gan_lab_amd import progressive as P
    P.FMAP_BASE, P.FMAP_MAX = 64, 16
    yield
    P.FMAP_BASE, P.FMAP_MAX = 8192, 512


def make_learner(kind, res, init_res=None, batch=4, **kw):
    from gan_lab_amd.config import make_config
    from gan_lab_amd.progan.learner import ProGANLearner
    from gan_lab_amd.stylegan.learner import StyleGANLearner
    common = dict(dev='cuda', pin_memory=False, res_samples=res, res_dataset=res, init_res=init_res or res,
                  batch_size=batch, len_latent=16, nimg_transition=24, num_iters_save_model=10 ** 9, log_every=0)
    common.update(kw)
    if kind == 'stylegan':
        cfg = make_config('stylegan', len_dlatent=16, mapping_num_fcs=2, cutoff_trunc_trick=None if res < 64 else 4,
                          **common)
        return StyleGANLearner(cfg)
    return ProGANLearner(make_config('progan', **common))


@pytest.mark.parametrize('name', ['step_stylegan16', 'step_stylegan8_fade', 'step_progan8'])
def test_training_step_matches_reference(name):
    from gan_lab_amd.stylegan.architectures import StyleAddNoise
    G = load_golden(name + '.npz')
    kind, loss, gp = [str(s) for s in G['meta']]
    res, alpha, fade = int(G['res']), float(G['alpha']), bool(G['fade_in'])
    L = make_learner(kind, res, loss=loss, gradient_penalty=gp, lr_base=float(G['lr']))
    L.gen_model.load_state_dict(sub(G, 'g0.'))
    L.disc_model.load_state_dict(sub(G, 'd0.'))
    L.ewma.flat.copy_(L.arena_g.flat)
    L.gen_model.fade_in_phase = fade
    L.gen_model.alpha = alpha if fade else 1
    L._set_optimizer()                       # parameter set depends on the phase (prev_torgb / prev_fromrgb)
    L.gen_model.train()
    L.disc_model.train()
    if kind == 'stylegan':
        L.gen_model.pct_mixing_reg = 0
        L.gen_model._use_mixing_reg = False
    L.beta = float(G['beta'])
    ok = {}

    def note(tag, model):
        for k, p in model.named_parameters():
            if p.grad is not None and p.grad.abs().max() > 0:
                m = (p.grad.abs() > 1e-5 * p.grad.abs().max()).cpu()
                ok[tag + k] = m if tag + k not in ok else (ok[tag + k] & m)

    StyleAddNoise.honour_noise_in_training = True
    try:
        for s in range(int(G['n_steps'])):
            kd = kg = {}
            if kind == 'stylegan':
                n = len(L.gen_model.gen_layers)
                kd = dict(noise=[t(G[f's{s}.nd{i}']).cuda() for i in range(n)])
                kg = dict(noise=[t(G[f's{s}.ng{i}']).cuda() for i in range(n)])
            L.set_requires_grad_disc(True)
            ld = L.d_step(t(G[f's{s}.real']).cuda(), zb=t(G[f's{s}.zd']).cuda(), gen_kwargs=kd,
                          eps_interp=t(G[f's{s}.eps_interp']).cuda())
            note('d.', L.disc_model)
            assert_close(ld, G[f's{s}.loss_d'], 1e-3, f'loss_d step {s}')
            L.set_requires_grad_disc(False)
            lg = L.g_step(zb=t(G[f's{s}.zg']).cuda(), gen_kwargs=kg)
            note('g.', L.gen_model)
            assert_close(lg, G[f's{s}.loss_g'], 1e-3, f'loss_g step {s}')
    finally:
        StyleAddNoise.honour_noise_in_training = False
    n_checked = 0
    for pre, tag, model in (('g1.', 'g.', L.gen_model), ('d1.', 'd.', L.disc_model)):
        cur = dict(model.state_dict())
        ref0 = sub(G, pre[0] + '0.')
        for k, v in sub(G, pre).items():
            du_ref, du = v - ref0[k], cur[k].detach().cpu() - ref0[k]
            if du_ref.abs().max() == 0:
                assert du.abs().max() == 0, k
            else:
                m = ok[tag + k]
                assert m.float().mean() > 0.5, k
                # Adam(beta1=0) normalises the step to ~lr*sign(g): 2% of lr is a tight bound on g parity
                assert_close(du[m], du_ref[m], 2e-2, 'update ' + pre + k)
                n_checked += int(m.sum())
    assert n_checked > 1000
    for k, v in sub(G, 'lag.').items():
        m = ok.get('g.' + k, torch.ones_like(v, dtype=torch.bool))
        assert_close(L.lagged_params[k].cpu()[m], v[m], 1e-4, 'ewma ' + k)


@pytest.mark.parametrize('kind', ['stylegan', 'progan'])
def test_train_loop_grows_and_stays_finite(kind):
    from gan_lab_amd.utils.data_utils import SyntheticImageLoader
    init = 8 if kind == 'stylegan' else 4
    kw = dict(loss='nonsaturating', gradient_penalty='r1') if kind == 'stylegan' else {}
    L = make_learner(kind, 16, init_res=init, batch=4, **kw)
    dl = SyntheticImageLoader(4096, 4, init)
    n_iters = 6 * (3 if kind == 'stylegan' else 5) + 2
    L.log_every = 1
    L.train(dl, num_main_iters=n_iters)
    assert L.gen_model.curr_res == 16 and not L.gen_model.fade_in_phase and L.gen_model.alpha == 1
    assert np.isfinite(L.last_losses['loss_d']) and np.isfinite(L.last_losses['loss_g'])
    assert L.sched.nimg_transition_lst[-1] == float('inf') and not L.progressively_grow
    # the loader was bumped to every resolution in order, at the scheduled batch sizes
    assert [r for _, r in dl.served] == sorted(r for _, r in dl.served)
    assert dl.served[-1] == (4, 16)
    img = L.gen_model(torch.randn(4, 16).cuda())
    assert img.shape == (4, 3, 16, 16) and torch.isfinite(img).all()
    keys = list(L.lagged_params.keys())
    assert keys == [k for k, _ in L.gen_model.named_parameters()]
    L.train(dl, num_main_iters=2)            # re-entrant: continues in the final phase
    assert L.gen_model.curr_res == 16
