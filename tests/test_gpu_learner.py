"""GPU tests of the learner: a D-iteration + G-iteration (fused Adam, EWMA) against training-step
vectors produced by the reference modules + torch.optim.Adam, and the full train() loop over a
growth schedule."""
import numpy as np
import pytest
import torch

from util import assert_close, load_golden, rel_err, sub, t

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


def test_load_reference_written_checkpoint(tmp_path):
    """tests/golden/ref_progan_ckpt.tar was written by the reference's own ProGANLearner.save_model after 9 main
    iterations (4x4 -> 8x8, mid fade-in).  Loading it must reproduce the reference's generator / EWMA generator /
    critic outputs, its next critic Adam step, and resume the phase machine where it stopped."""
    import os
    from gan_lab_amd.config import make_config
    from gan_lab_amd.progan.learner import ProGANLearner
    from gan_lab_amd.utils.data_utils import SyntheticImageLoader
    here = os.path.join(os.path.dirname(__file__), 'golden')
    E = load_golden('ref_progan_ckpt_expect.npz')
    bs = 4
    cfg = make_config('progan', dev='cuda', pin_memory=False, res_samples=16, res_dataset=16, init_res=4, batch_size=bs,
                      len_latent=16, nimg_transition=22, num_iters_save_model=10 ** 9, log_every=1,
                      bs_dict={4: bs, 8: bs, 16: bs // 2, 32: bs, 64: bs, 128: bs, 256: bs, 512: bs // 2,
                               1024: bs // 4})
    cfg.lr_fctr_dict = {4: 1, 8: 1.25, 16: 1.5, 32: 1, 64: 1, 128: 1.5, 256: 2, 512: 3, 1024: 3}
    L = ProGANLearner(cfg)
    L.load_model(os.path.join(here, 'ref_progan_ckpt.tar'))
    assert L.gen_model.curr_res == int(E['curr_res']) == 8 and L.gen_model.fade_in_phase
    assert abs(L.gen_model.alpha - float(E['alpha'])) < 1e-12
    assert L.curr_img_num == int(E['curr_img_num']) and L.curr_phase_num == int(E['curr_phase_num'])
    assert L.batch_size == int(E['batch_size']) and L.pretrained_model and not L.not_trained_yet
    z, real = t(E['z']).cuda(), t(E['real']).cuda()
    L.gen_model.eval()
    with torch.no_grad():
        assert_close(L.gen_model(z).cpu(), E['img'], 1e-3, 'G(z) eval')
        lag = L.materialize_lagged_generator().eval()
        assert_close(lag(z).cpu(), E['img_lagged'], 1e-3, 'EWMA G(z) eval')
    L.gen_model.train()
    L.disc_model.train()
    with torch.no_grad():
        assert_close(L.gen_model(z).cpu(), E['fake_train'], 1e-3, 'G(z) train')
    # the next critic step, on the reference's own latents, reals and interpolation draw
    before = {k: v.detach().clone() for k, v in L.disc_model.named_parameters()}
    L.set_requires_grad_disc(True)
    ld = L.d_step(real, zb=z, eps_interp=t(E['eps_interp']).cuda())
    assert_close(ld.cpu(), E['loss_d'], 2e-3, 'loss_d')
    n = 0
    for k, p in L.disc_model.named_parameters():
        if 'gd.' + k not in E:
            continue
        gref = t(E['gd.' + k])
        m = gref.abs() > 1e-3 * gref.abs().max()
        du, dref = (p.detach() - before[k]).cpu(), t(E['dd.' + k])
        bad = (du[m] - dref[m]).abs() > 3e-2 * dref.abs().max()
        assert bad.float().mean() <= 5e-3, k
        n += int(m.sum())
    assert n > 1000
    # resume: the phase machine continues into stabilisation and the 16x16 growth
    dl = SyntheticImageLoader(4096, 4, 4)
    L.train(dl, num_main_iters=14)
    assert L.gen_model.curr_res == 16 and dl.served[0] == (4, 8) and dl.served[-1] == (2, 16)
    assert np.isfinite(L.last_losses['loss_d'])
    # and this package's own checkpoint round-trips (Adam moments included)
    path = tmp_path / 'progan_model.tar'
    L.save_model(path)
    L2 = ProGANLearner(cfg)
    L2.load_model(path)
    assert L2.gen_model.curr_res == 16 and L2.curr_img_num == L.curr_img_num
    for a, b in ((L.arena_g, L2.arena_g), (L.arena_d, L2.arena_d)):
        assert torch.equal(a.flat, b.flat)
    x = (torch.rand(2, 3, 16, 16) * 2 - 1).cuda()
    z2, e2 = torch.randn(2, 16).cuda(), torch.rand(2, 1, 1, 1).cuda()
    L2.opt_disc.param_groups[0]['lr'] = L.opt_disc.param_groups[0]['lr']    # train() would set it (LambdaLR)
    for lr in (L, L2):
        lr.disc_model.train()
        lr.set_requires_grad_disc(True)
        lr.d_step(x, zb=z2, eps_interp=e2)
    assert rel_err(L2.arena_d.flat.cpu(), L.arena_d.flat.cpu()) < 1e-6


def test_keyboard_interrupt_saves_a_checkpoint(tmp_path):
    """Ctrl-C during train() saves the latest checkpoint before the exception propagates (progan/learner.py:986-1013)."""
    from gan_lab_amd.utils.data_utils import SyntheticImageLoader
    L = make_learner('progan', 8, init_res=8, batch=4, save_model_dir=tmp_path)
    dl = SyntheticImageLoader(4096, 4, 8)
    orig, n = L.g_step, [0]

    def g_step(*a, **k):
        n[0] += 1
        if n[0] == 3:
            raise KeyboardInterrupt
        return orig(*a, **k)
    L.g_step = g_step
    with pytest.raises(KeyboardInterrupt):
        L.train(dl, num_main_iters=10)
    ck = tmp_path / 'progan_model.tar'
    assert ck.exists()
    L2 = make_learner('progan', 8, init_res=8, batch=4)
    L2.load_model(ck)
    for (k, v), (_, v2) in zip(L.gen_model.state_dict().items(), L2.gen_model.state_dict().items()):
        assert torch.equal(v.cpu(), v2.cpu()), k


def test_training_is_bitwise_reproducible():
    """Every reduction in the HIP path has a fixed order (split-K partials, weight-gradient slots, channel sums, the
    InstanceNorm statistics) and the device RNG is counter-based: two learners built from the same seeds and fed the same
    batches must agree BIT FOR BIT after several G+D iterations (parameters, Adam moments, EWMA generator) - at a width
    where the stride-2, split-K, layer-tail and deferred-activation kernels are all on the path."""
    from gan_lab_amd import progressive as P, rng
    from gan_lab_amd.utils.data_utils import SyntheticImageLoader

    def run():
        P.FMAP_BASE, P.FMAP_MAX = 2048, 64          # 64 channels at 4x4 .. 32x32, 32 at 64x64
        try:
            torch.manual_seed(7)
            np.random.seed(7)         # the mixing-regularisation coin is np.random.rand(), as in the reference
            rng.manual_seed(99)       # overwritten by the learner: config.random_seed drives the device stream
            L = make_learner('stylegan', 64, batch=4, loss='nonsaturating', gradient_penalty='r1', random_seed=7)
            L.train(SyntheticImageLoader(64, 4, 64, seed=5), num_main_iters=3)
            torch.cuda.synchronize()
            state = {('g', k): v.detach().clone() for k, v in L.gen_model.state_dict().items()}
            state.update({('d', k): v.detach().clone() for k, v in L.disc_model.state_dict().items()})
            state.update({('lag', k): v.detach().clone() for k, v in L.lagged_params.items()})
            for name, opt in (('og', L.opt_gen), ('od', L.opt_disc)):
                for i, st in enumerate(opt.state_dict()['state'].values() if hasattr(opt, 'state_dict') else []):
                    for kk, vv in st.items():
                        if torch.is_tensor(vv):
                            state[(name, i, kk)] = vv.detach().clone()
            return state, dict(L.last_losses)
        finally:
            P.FMAP_BASE, P.FMAP_MAX = 8192, 512
    a, la = run()
    b, lb = run()
    assert a.keys() == b.keys() and len(a) > 100
    diff = [k for k in a if not torch.equal(a[k], b[k])]
    assert not diff, f'{len(diff)} of {len(a)} tensors differ between two identical runs, e.g. {diff[:3]}'
    assert la == lb


def test_direct_parameter_gradients_equal_the_accumulated_ones():
    """``ops.direct_param_grads`` (the learners' backward sweeps in a single-process run): the first gradient of a parameter
    in a step is written into its zeroed arena slot by the kernel that computes it; the others are summed by the engine
    and added by the parameter's AccumulateGrad node, which runs after every contributing Function - so nothing can be
    overwritten.  Only the order of the additions differs from the plain path (g1 + (g2 + g3) against ((g1 + g2) + g3)):
    both arenas and the updated parameters must agree to fp32 rounding - and most of the per-parameter ``grad += g`` launches must be gone."""
    from gan_lab_amd import ops, progressive as P, rng
    from gan_lab_amd.utils.data_utils import SyntheticImageLoader
    P.FMAP_BASE, P.FMAP_MAX = 2048, 64
    real = next(iter(SyntheticImageLoader(64, 4, 64, seed=3)))
    real = (real[0] if isinstance(real, (tuple, list)) else real).cuda()

    class Off(object):                                   # stands in for ops.direct_param_grads: never enables
        def __init__(self, *_):
            pass

        def __enter__(self):
            pass

        def __exit__(self, *exc):
            return False

    def run(direct):
        torch.manual_seed(3)
        np.random.seed(3)
        L = make_learner('stylegan', 64, batch=4, loss='nonsaturating', gradient_penalty='r1', random_seed=11)
        L.gen_model.train()
        L.disc_model.train()
        L.beta = 0.999
        rng.manual_seed(5)
        orig = ops.direct_param_grads
        taken = [0]
        if direct:
            take = ops._take

            def counting(name, shape, like):
                out = take(name, shape, like)
                taken[0] += int(bool(ops._TAKEN.get(name)))
                return out
            ops._take = counting
        else:
            ops.direct_param_grads = Off
        try:
            L.d_step(real, defer_update=True)
            gd = L.arena_d.gflat.clone()
            L.g_step(d_update_pending=True)
            gg = L.arena_g.gflat.clone()
        finally:
            ops.direct_param_grads = orig
            if direct:
                ops._take = take
        torch.cuda.synchronize()
        n_params = len(L.arena_d.params) + len(L.arena_g.params)
        return gd, gg, L.arena_d.flat.clone(), L.arena_g.flat.clone(), taken[0], n_params
    try:
        a, b = run(True), run(False)
    finally:
        P.FMAP_BASE, P.FMAP_MAX = 8192, 512
    for i, what in enumerate(('critic gradients', 'generator gradients', 'critic parameters', 'generator parameters')):
        assert a[i].abs().max() > 0
        if what == 'generator gradients':   # taken through the critic AFTER its update, which already differs by rounding
            assert (a[i] - b[i]).abs().max().item() <= 1e-3 * b[i].abs().max().item(), what
        elif 'gradients' in what:
            assert (a[i] - b[i]).abs().max().item() <= 4e-6 * b[i].abs().max().item(), what
        else:       # one Adam step of lr 1e-3: a sign-sized move where a gradient is rounding noise around zero
            assert (a[i] - b[i]).abs().max().item() <= 2.1e-3, what
    assert a[4] >= 0.6 * a[5], f'only {a[4]} of {a[5]} parameters took the direct path'
    assert b[4] == 0


@pytest.mark.parametrize('kind,flush', [('stylegan', None), ('progan', None), ('stylegan', 'keepalive'),
                                        ('stylegan', 'recapture')])
def test_graphed_step_equals_eager(kind, flush, monkeypatch):
    """graphs.GraphedStep: the stabilised iteration replayed as HIP graphs (device-resident Philox position and Adam
    scalars, one graph per style-mixing cut and half) against the same learner stepping eagerly - parameters, Adam moments,
    EWMA generator, the running w average and both losses BIT FOR BIT after 6 iterations, 4 of them replayed.

    ``flush`` (use-after-free): after the first replayed iteration the pack cache is flushed
    (``ops.bump_weight_epoch()``: what a cache overflow, a sampling graph or a growth event does) and the freed memory is
    recycled and scribbled over.  'keepalive': the graphs survive (the cache generation is pinned for the test) and must
    replay through the buffers and descriptor tables they hold references to; 'recapture': the generation in the
    signature drops them, two eager iterations and a fresh capture follow.  Both must still equal the eager run."""
    from gan_lab_amd import ops, progressive as P, rng
    from gan_lab_amd.graphs import GraphedStep
    gen = torch.Generator().manual_seed(17)
    reals = [(torch.rand(4, 3, 32, 32, generator=gen) * 2 - 1).cuda() for _ in range(6)]
    if flush == 'keepalive':
        monkeypatch.setattr(ops, 'pack_generation', lambda: 0)

    def run(graphed):
        P.FMAP_BASE, P.FMAP_MAX = 1024, 64
        torch.manual_seed(9)
        np.random.seed(9)
        kw = dict(loss='nonsaturating', gradient_penalty='r1') if kind == 'stylegan' else \
            dict(loss='wgan', gradient_penalty='wgan-gp')
        L = make_learner(kind, 32, batch=4, random_seed=21, **kw)
        L.gen_model.train()
        L.disc_model.train()
        L.beta = 0.99
        torch.manual_seed(10)               # the WGAN-GP interpolation weights come from torch's device generator
        stepper = GraphedStep(L, warmup=2)
        losses = []
        for it, x in enumerate(reals):
            if graphed and flush and it == 3:
                assert stepper.graphs, 'nothing was captured before the flush'
                ops.bump_weight_epoch()
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
                junk = [torch.full((1 << 18,), float('nan'), device='cuda') for _ in range(64)]   # recycle what was freed
                del junk
            if graphed:
                ld, lg = stepper(x)
            else:
                cut, kw_d = stepper._mix_kwargs()
                ld = stepper._d_half(x, kw_d)
                cut, kw_g = stepper._mix_kwargs()
                lg = stepper._g_half(kw_g)
            losses.append((float(ld), float(lg)))
        torch.cuda.synchronize()
        state = {'g': L.arena_g.flat.clone(), 'd': L.arena_d.flat.clone(), 'lag': L.ewma.flat.clone()}
        for name, opt in (('og', L.opt_gen), ('od', L.opt_disc)):
            ex = opt.export_moments(list(L.gen_model.named_parameters()) if name == 'og' else
                                    list(L.disc_model.named_parameters()))
            state[name + '.step'] = torch.tensor(ex['step'])
            for k2, v in ex['exp_avg'].items():
                state[f'{name}.m.{k2}'] = v
            for k2, v in ex['exp_avg_sq'].items():
                state[f'{name}.v.{k2}'] = v
        if getattr(L.gen_model, 'w_ewma', None) is not None:
            state['w_ewma'] = L.gen_model.w_ewma.clone()
        return state, losses, (len(stepper.graphs) if graphed else 0), rng._STATE['offset']
    try:
        a, la, n_graphs, off_a = run(True)
        b, lb, _, off_b = run(False)
    finally:
        P.FMAP_BASE, P.FMAP_MAX = 8192, 512
    assert n_graphs >= 2, 'nothing was captured'
    assert off_a == off_b, 'the device random stream advanced differently'
    assert la == lb, (la, lb)
    assert a.keys() == b.keys()
    diff = [k for k in a if not torch.equal(a[k].cpu(), b[k].cpu())]
    assert not diff, f'{len(diff)} of {len(a)} tensors differ between replayed and eager steps, e.g. {diff[:4]}'


def test_load_reference_written_stylegan_checkpoint(tmp_path):
    """tests/golden/ref_stylegan_ckpt.tar was written by the reference's own StyleGANLearner.save_model
    (stylegan/learner.py:432-501: 9 main iterations, 4x4 -> 8x8 mid fade-in, truncation trick on).  The expectations
    come from the reference's OWN load_model (:503-640) of that file: truncation state, the ``w_ewma`` the generator and
    the EWMA generator end up with, and their eval-mode images (truncation applied) on fixed latents / noise."""
    import os
    from gan_lab_amd.config import make_config
    from gan_lab_amd.stylegan.learner import StyleGANLearner
    from gan_lab_amd.utils.data_utils import SyntheticImageLoader
    here = os.path.join(os.path.dirname(__file__), 'golden')
    E = load_golden('ref_stylegan_ckpt_expect.npz')
    bs = 4
    cfg = make_config('stylegan', dev='cuda', pin_memory=False, res_samples=16, res_dataset=16, init_res=4,
                      batch_size=bs, len_latent=16, len_dlatent=16, mapping_num_fcs=2, nimg_transition=22,
                      beta_trunc_trick=.9, psi_trunc_trick=.7, cutoff_trunc_trick=1, loss='nonsaturating',
                      gradient_penalty='r1', num_iters_save_model=10 ** 9, log_every=1,
                      bs_dict={4: bs, 8: bs, 16: bs // 2, 32: bs, 64: bs, 128: bs, 256: bs, 512: bs // 2,
                               1024: bs // 4})
    cfg.lr_fctr_dict = {4: 1, 8: 1.25, 16: 1.5, 32: 1, 64: 1, 128: 1.5, 256: 2, 512: 3, 1024: 3}
    from gan_lab_amd import progressive as P
    P.FMAP_BASE, P.FMAP_MAX = 64, 16
    try:
        L = StyleGANLearner(cfg)
        L.load_model(os.path.join(here, 'ref_stylegan_ckpt.tar'))
        g = L.gen_model
        assert g.curr_res == int(E['curr_res']) == 8 and g.fade_in_phase == bool(E['fade_in'])
        assert abs(g.alpha - float(E['alpha'])) < 1e-12
        assert L.curr_img_num == int(E['curr_img_num']) and L.curr_phase_num == int(E['curr_phase_num'])
        assert g.use_truncation_trick == bool(E['use_truncation_trick'])
        assert g.trunc_cutoff_stage == int(E['trunc_cutoff_stage'])
        assert g.w_eval_psi == float(E['w_eval_psi']) and g.w_ewma_beta == float(E['w_ewma_beta'])
        assert g.pct_mixing_reg == float(E['pct_mixing_reg'])
        assert_close(g.w_ewma.cpu(), E['w_ewma'], 1e-6, 'generator w_ewma after load')
        assert_close(L.gen_model_lagged.w_ewma.cpu(), E['w_ewma_lagged_model'], 1e-6, 'EWMA generator w_ewma after load')
        assert_close(L.ds_mean, E['ds_mean'], 0, 'ds_mean')
        assert_close(L.ds_std, E['ds_std'], 0, 'ds_std')
        assert_close(L.valid_z.cpu(), E['valid_z'], 0, 'valid_z')
        z = t(E['z']).cuda()
        noise = [t(E[f'noise{i}']).cuda() for i in range(len(g.gen_layers))]
        g.eval()
        lag = L.gen_model_lagged.eval()
        with torch.no_grad():
            assert_close(g(z, noise=noise).cpu(), E['img'], 1e-3, 'G(z) eval, truncation trick')
            assert_close(lag(z, noise=noise).cpu(), E['img_lagged'], 1e-3, 'EWMA G(z) eval, truncation trick')
            g.use_truncation_trick = False
            assert_close(g(z, noise=noise).cpu(), E['img_no_trunc'], 1e-3, 'G(z) eval, no truncation')
            g.use_truncation_trick = True
        # resume training from it, then write / re-read this package's own format with the truncation state in it
        g.train()
        L.train(SyntheticImageLoader(4096, 4, 4), num_main_iters=3)
        assert np.isfinite(L.last_losses['loss_d'])
        path = tmp_path / 'stylegan_model.tar'
        L.save_model(path)
        L2 = StyleGANLearner(cfg)
        L2.load_model(path)
        assert torch.equal(L2.gen_model.w_ewma, L.gen_model.w_ewma) and L2.gen_model.trunc_cutoff_stage == 1
        assert torch.equal(L2.arena_g.flat, L.arena_g.flat) and torch.equal(L2.ds_mean, L.ds_mean)
        # and the reference-format writer runs on the device learner too (read back by this package's reader)
        L.save_model(tmp_path / 'ref_format.tar', reference_format=True)
        from gan_lab_amd import checkpoint as ckpt
        ck = ckpt.load_checkpoint(tmp_path / 'ref_format.tar')
        assert ckpt.is_reference_format(ck) and 'w_ewma_lagged' in ck and ck['opt_gen_state_dict']['state']
        L3 = StyleGANLearner(cfg)
        L3.load_model(tmp_path / 'ref_format.tar')
        assert torch.equal(L3.arena_d.flat, L.arena_d.flat)
    finally:
        P.FMAP_BASE, P.FMAP_MAX = 8192, 512


def test_training_step_frees_its_activations_without_the_cyclic_collector():
    """No autograd node may keep its own output alive through ``ctx`` attributes (a reference cycle: the activations of
    a step would then be released only when Python's cyclic collector happens to run - at StyleGAN-1024 that was 2-4 GiB
    of allocator growth per step).  With the collector DISABLED, device memory after a step must be what it was after the
    step before - on a network whose generator runs the deferred-InstanceNorm chain (16 channels at 64x64)."""
    import gc
    from gan_lab_amd import _lib, progressive as P
    from gan_lab_amd.utils.data_utils import SyntheticImageLoader
    P.FMAP_BASE, P.FMAP_MAX = 512, 64              # 64 channels up to 8x8, 32 at 16x16 ... 16 at 64x64
    calls = {'mod': 0}
    L_ = _lib.lib()
    orig = L_.ganlab_mod_conv_fwd_f32

    def counted(*a):
        calls['mod'] += 1
        return orig(*a)
    L_.ganlab_mod_conv_fwd_f32 = counted
    try:
        L = make_learner('stylegan', 64, batch=4, loss='nonsaturating', gradient_penalty='r1', random_seed=3)
        dl = SyntheticImageLoader(4096, 4, 64)
        L.train(dl, num_main_iters=2)
        torch.cuda.synchronize()
        gc.collect()
        gc.disable()
        try:
            L.train(dl, num_main_iters=1)
            torch.cuda.synchronize()
            a1 = torch.cuda.memory_allocated()
            L.train(dl, num_main_iters=2)
            torch.cuda.synchronize()
            a2 = torch.cuda.memory_allocated()
        finally:
            gc.enable()
    finally:
        L_.ganlab_mod_conv_fwd_f32 = orig
        P.FMAP_BASE, P.FMAP_MAX = 8192, 512
    assert calls['mod'] >= 4, calls            # the modulated layer was on the path
    assert a2 <= a1, (a1, a2)


@pytest.mark.parametrize('batch,paired', [(8, True), (4, True), (6, False)])
def test_progan_paired_critic_pass_equals_two_passes(batch, paired, monkeypatch):
    """The WGAN-GP critic iteration scores [generated; real] in ONE pass when the minibatch-stddev groups (contiguous
    groups of 4, custom_layers.py:117-140) stay inside each half - a batch that is a multiple of 4; loss and every critic
    gradient equal those of the reference's two passes (progan/learner.py:786-800) up to the order of the weight-gradient
    sums.  A batch of 6 (one group = the whole batch) must NOT pair."""
    from gan_lab_amd import rng
    out = {}
    for pair in ('0', '1'):
        monkeypatch.setenv('GANLAB_CRITIC_PAIR', pair)
        torch.manual_seed(5)
        rng.manual_seed(5)
        L = make_learner('progan', 16, batch=batch, loss='wgan', gradient_penalty='wgan-gp', random_seed=5)
        L.gen_model.train()
        L.disc_model.train()
        g = torch.Generator().manual_seed(9)
        real = (torch.rand(batch, 3, 16, 16, generator=g) * 2 - 1).cuda()
        zd = torch.randn(batch, 16, generator=g).cuda()
        eps = torch.rand(batch, 1, 1, 1, generator=g).cuda()
        if pair == '0':
            w0 = L.arena_d.flat.detach().clone(), L.arena_g.flat.detach().clone()
        else:
            with torch.no_grad():
                L.arena_d.flat.copy_(w0[0])
                L.arena_g.flat.copy_(w0[1])
            from gan_lab_amd import ops
            ops.bump_weight_epoch()
        assert L._pair_critic_batches(torch.empty_like(real), real) == (pair == '1' and paired)
        L.set_requires_grad_disc(True)
        ld = L.d_step(real, zb=zd, eps_interp=eps, defer_update=True)
        out[pair] = (ld.cpu(), L.arena_d.gflat.detach().cpu().clone())
    assert out['0'][1].abs().max() > 0
    assert_close(out['1'][0], out['0'][0], 1e-6, 'loss_d')
    assert_close(out['1'][1], out['0'][1], 1e-5, 'critic gradients')
