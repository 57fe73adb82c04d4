"""GPU tests of the ResNet GAN path (BASELINE config #5): BatchNorm / LayerNorm / tanh kernels incl.
the WGAN-GP double backward through LayerNorm, the 32 / 64 pixel nets against reference vectors, the
GANLearner iteration (generator first, then critic iterations) against reference + torch.optim.Adam,
and a full-width 64x64 step against the oracle.  Tolerance: 1e-3 relative fp32 (north star)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util import assert_close, load_golden, rel_err, resnet_zero_grad_key, sub, t

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _cmp_grads(model, ref_by_key, tol, what):
    gmax = max(float(np.abs(v).max()) for v in ref_by_key.values())
    named = dict(model.named_parameters())
    for k, ref in ref_by_key.items():
        b = torch.from_numpy(ref).double()
        if named[k].grad is None:      # no path to the loss: the reference holds exact zeros there
            assert b.abs().max() == 0, k
            continue
        a = named[k].grad.detach().double().cpu()
        den = gmax if resnet_zero_grad_key(k) else max(b.abs().max().item(), 1e-4 * gmax)
        e = (a - b).abs().max().item() / den
        assert e <= tol, f'{what} {k}: rel err {e:.3e} > {tol:.1e}'


# ---------------------------------------------------------------------------------------------- #
def test_chan_affine_mul_tanh():
    from gan_lab_amd import ops
    torch.manual_seed(0)
    for shape in [(3, 5, 7, 9), (2, 8, 16, 16), (4, 6, 1)]:
        x = torch.randn(*shape)
        s, b = torch.randn(shape[1]), torch.randn(shape[1])
        view = [1, shape[1]] + [1] * (len(shape) - 2)
        xg, sg, bg = (v.clone().cuda().requires_grad_(True) for v in (x, s, b))
        y = ops.chan_affine(xg, sg, bg)
        xr, sr, br = (v.clone().requires_grad_(True) for v in (x, s, b))
        yr = xr * sr.view(view) + br.view(view)
        assert_close(y.cpu(), yr, 1e-6, 'chan_affine')
        cot = torch.randn(*shape)
        y.backward(cot.cuda())
        yr.backward(cot)
        for a, r, n in ((xg, xr, 'gx'), (sg, sr, 'gscale'), (bg, br, 'gshift')):
            assert_close(a.grad.cpu(), r.grad, 1e-5, n)
        assert_close(ops.chan_affine(xg, None, bg).cpu(), x + b.view(view), 1e-6, 'shift only')
    a, b = torch.randn(1000), torch.randn(1000)
    assert_close(ops.mul(a.cuda(), b.cuda()).cpu(), a * b, 1e-7, 'mul')
    xg = a.clone().cuda().requires_grad_(True)
    y = ops.tanh(xg)
    assert_close(y.cpu(), torch.tanh(a), 1e-6, 'tanh')
    y.backward(b.cuda())
    assert_close(xg.grad.cpu(), b * (1 - torch.tanh(a) ** 2), 1e-5, 'tanh bwd')
    # second order (a gradient penalty through a tanh critic): L = sum (d<tanh(x), b> / dx)^2 against torch's own tanh
    xg = a.clone().cuda().requires_grad_(True)
    xr = a.clone().requires_grad_(True)
    gg, = torch.autograd.grad((ops.tanh(xg) * b.cuda()).sum(), xg, create_graph=True)
    gr, = torch.autograd.grad((torch.tanh(xr) * b).sum(), xr, create_graph=True)
    (gg ** 2).sum().backward()
    (gr ** 2).sum().backward()
    assert_close(xg.grad.cpu(), xr.grad, 1e-5, 'tanh second order')
    g = torch.randn(3, 4, 8, 8)
    assert_close(ops.global_avg_pool(g.cuda()).cpu(), g.mean(dim=(2, 3), keepdim=True), 1e-6, 'global avg pool')


def test_batch_norm_train_and_eval():
    from gan_lab_amd.utils.custom_layers import BatchNorm2d
    torch.manual_seed(1)
    ref = torch.nn.BatchNorm2d(6)
    mine = BatchNorm2d(6)
    with torch.no_grad():
        ref.weight.copy_(torch.randn(6) * .3 + 1)
        ref.bias.copy_(torch.randn(6) * .3)
    mine.load_state_dict(ref.state_dict())
    mine.cuda()
    for step in range(3):
        x = torch.randn(5, 6, 8, 12) * 2 + 0.5
        xr = x.clone().requires_grad_(True)
        xg = x.clone().cuda().requires_grad_(True)
        if step == 2:     # the ReLU behind it applied by the normalisation's own passes, forward and backward
            yr, y = F.relu(ref(xr)), mine(xg, act_slope=0.0)
        else:
            yr, y = ref(xr), mine(xg)
        assert_close(y.cpu(), yr, 1e-5, 'bn fwd')
        cot = torch.randn_like(x)
        ref.zero_grad()
        mine.zero_grad()
        yr.backward(cot)
        y.backward(cot.cuda())
        assert_close(xg.grad.cpu(), xr.grad, 1e-4, 'bn gx')
        assert_close(mine.weight.grad.cpu(), ref.weight.grad, 1e-4, 'bn gw')
        assert_close(mine.bias.grad.cpu(), ref.bias.grad, 1e-4, 'bn gb')
    for k, v in ref.state_dict().items():
        assert_close(mine.state_dict()[k].cpu().double(), v.double(), 1e-5, k)
    ref.eval()
    mine.eval()
    x = torch.randn(3, 6, 4, 4)
    assert_close(mine(x.cuda()).cpu(), ref(x), 1e-5, 'bn eval')


@pytest.mark.parametrize('fused_slope', [None, 0.0, 0.2])
@pytest.mark.parametrize('shape', [(4, 3, 8, 8), (2, 8, 16, 16), (3, 5, 4, 4),
                                   (3, 16, 32, 32), (2, 32, 64, 64)])      # the last two: several statistics blocks per row
def test_layer_norm_first_and_second_order(shape, fused_slope):
    """LayerNorm([C,R,R]) inside a WGAN-GP style double backward: d/dtheta of |d out / d x|^2.  ``fused_slope``: the
    LeakyReLU behind it applied by the normalisation's own passes (forward, backward and double backward) against
    nn.LayerNorm followed by leaky_relu."""
    from gan_lab_amd.utils.custom_layers import LayerNorm
    torch.manual_seed(2)
    ref = torch.nn.LayerNorm(list(shape[1:]))
    with torch.no_grad():
        ref.weight.copy_(torch.randn(shape[1:]) * .3 + 1)
        ref.bias.copy_(torch.randn(shape[1:]) * .3)
    mine = LayerNorm(list(shape[1:]))
    mine.load_state_dict(ref.state_dict())
    mine.cuda()
    x = torch.randn(*shape) * 1.5 + 0.2
    w2 = torch.randn(*shape)

    def run(mod, x, w2):
        x = x.clone().requires_grad_(True)
        if fused_slope is None:
            y = mod(x)
        elif mod is mine:
            y = mod(x, act_slope=fused_slope)
        else:
            y = F.leaky_relu(mod(x), fused_slope)
        out = (F.relu(y - 0.1) * w2).sum(dim=(1, 2, 3))
        gx, = torch.autograd.grad(out, x, torch.ones_like(out), create_graph=True)
        pen = (gx ** 2).sum() + out.sum()
        mod.zero_grad()
        pen.backward()
        return y, gx, x.grad, mod.weight.grad, mod.bias.grad

    r = run(ref, x, w2)
    m = run(mine, x.cuda(), w2.cuda())
    for a, b, n in zip(m, r, ('y', 'gx', 'd pen/dx', 'd pen/dw', 'd pen/db')):
        assert_close(a.detach().cpu(), b.detach(), 2e-4, 'layernorm ' + n)


# ---------------------------------------------------------------------------------------------- #
RESNET_CASES = [(32, None), (64, None), (32, 'tanh')]      # (resolution, --nonlinearity other than the default ReLU)
RESNET_IDS = ['32', '64', '32-tanh']


def _resnet_golden(res, nl):
    return load_golden(f'resnet{res}.npz' if nl is None else f'resnet{res}_{nl}.npz')


def _nets(G, res, nl=None):
    from torch import nn
    from gan_lab_amd.resnetgan import architectures as A
    kw = {} if nl is None else {'nl': {'tanh': nn.Tanh}[nl]()}     # the reference's own module instance is accepted
    if res == 64:
        g = A.Generator64PixResnet(len_latent=int(G['len_latent']), fmap=int(G['fmap_g']), **kw)
        d = A.Discriminator64PixResnet(fmap=int(G['fmap_d']), **kw)
    else:
        g = A.Generator32PixResnet(len_latent=int(G['len_latent']), fmap=int(G['fmap_g']), **kw)
        d = A.Discriminator32PixResnet(fmap=int(G['fmap_d']), **kw)
    g.load_state_dict(sub(G, 'g0.'))
    d.load_state_dict(sub(G, 'd0.'))
    return g.cuda().train(), d.cuda().train()


@pytest.mark.parametrize('res,nl', RESNET_CASES, ids=RESNET_IDS)
def test_resnet_nets_match_reference(res, nl):
    """Forward, generator gradients, WGAN-GP value and its double-backward gradients against the reference's fixtures;
    the tanh case is ``--nonlinearity tanh`` as the hidden activation (resnetgan/learner.py:180-181): first and second
    order of ``ops.tanh`` inside whole networks."""
    from gan_lab_amd.utils import backprop_utils as bp
    G = _resnet_golden(res, nl)
    g, d = _nets(G, res, nl)
    img = g(t(G['z']).cuda())
    assert_close(img.cpu(), G['img'], TOL, 'img')
    for k, v in sub(G, 'g_after_fwd.').items():
        assert_close(g.state_dict()[k].cpu(), v, 1e-4, 'running stat ' + k)
    for p in d.parameters():
        p.requires_grad_(False)
    dout = d(img)
    assert_close(dout.cpu(), G['d_of_img'], TOL, 'D(G(z))')
    g.zero_grad()
    bp.loss_gen('wgan', dout).backward()
    _cmp_grads(g, {k[3:]: v for k, v in G.items() if k.startswith('gg.')}, TOL, 'G grad')
    for p in d.parameters():
        p.requires_grad_(True)
    fake, real, eps = img.detach(), t(G['real']).cuda(), t(G['eps_interp']).cuda()
    assert_close(d(fake).cpu(), G['d_fake'], TOL, 'd_fake')
    assert_close(d(real).cpu(), G['d_real'], TOL, 'd_real')
    d.zero_grad()
    gpv = bp.calc_gp(d, 'wgan-gp', fake, real, lda=10., gamma=1., eps_interp=eps)
    assert_close(gpv.cpu(), G['gp'], TOL, 'gp')
    gpv.backward()
    _cmp_grads(d, {k[4:]: v for k, v in G.items() if k.startswith('ggp.')}, TOL, 'GP-only grad')
    d.zero_grad()
    ld = bp.loss_disc('wgan', d(fake), d(real)) + bp.calc_gp(d, 'wgan-gp', fake, real, lda=10., gamma=1.,
                                                              eps_interp=eps)
    assert_close(ld.cpu(), G['loss_d'], TOL, 'loss_d')
    ld.backward()
    _cmp_grads(d, {k[3:]: v for k, v in G.items() if k.startswith('gd.')}, TOL, 'D grad')


def make_learner(res, batch=4, **kw):
    from gan_lab_amd.config import make_config
    from gan_lab_amd.resnetgan.learner import GANLearner
    cfg = make_config('resnetgan', dev='cuda', pin_memory=False, res_samples=res, res_dataset=res, batch_size=batch,
                      num_iters_save_model=10 ** 9, log_every=0, **kw)
    return cfg, GANLearner


@pytest.mark.parametrize('res,nl', RESNET_CASES, ids=RESNET_IDS)
def test_resnet_training_iterations_match_reference(res, nl):
    """Two main iterations (G step, then 2 critic steps each) from the reference's fixture.  Losses are
    checked against the reference's own values.  Parameter updates are judged against a float64 replay
    of the oracle: on these narrow nets torch's CPU fp32 BatchNorm backward is itself ~3e-3 away from
    the float64 result for the deep generator parameters (measured; the HIP path is ~1e-6 away), and
    Adam(beta1=0) turns a relative gradient error straight into a relative update error."""
    from oracle import resnet
    G = _resnet_golden(res, nl)
    cfg, Learner = make_learner(res, len_latent=int(G['len_latent']), lr_base=float(G['lr']),
                                **({} if nl is None else {'nonlinearity': nl}))
    cfg.fmap_g, cfg.fmap_d = int(G['fmap_g']), int(G['fmap_d'])
    L = Learner(cfg)
    L.gen_model.load_state_dict(sub(G, 'g0.'))
    L.disc_model.load_state_dict(sub(G, 'd0.'))
    assert L.arena_g.is_attached() and L.arena_d.is_attached()
    L.gen_model.train()
    L.disc_model.train()
    dbl = lambda sd: {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}  # noqa: E731
    gan = resnet.ResnetFunctionalGAN(dbl(sub(G, 'g0.')), dbl(sub(G, 'd0.')), res, lr=float(G['lr']), nl=nl)
    ok = {}

    def note(tag, params):
        for k, p in params.items():
            if p.grad is not None and p.grad.abs().max() > 0:
                m = p.grad.abs() > 1e-4 * p.grad.abs().max()
                ok[tag + k] = m if tag + k not in ok else (ok[tag + k] & m)

    for it in range(int(G['n_iters'])):
        L.set_requires_grad_disc(False)
        zg = t(G[f'i{it}.zg'])
        lg = L.g_step(zb=zg.cuda())
        lg64 = gan.g_step(zg.double())
        note('g.', gan.g)
        assert_close(lg.cpu(), G[f'i{it}.loss_g'], TOL, f'loss_g {it}')
        assert_close(lg.cpu().double(), lg64, TOL, f'loss_g {it} (f64)')
        L.set_requires_grad_disc(True)
        for di in range(int(G['n_disc'])):
            p = f'i{it}.d{di}.'
            ld = L.d_step(t(G[p + 'real']).cuda(), zb=t(G[p + 'zd']).cuda(), eps_interp=t(G[p + 'eps_interp']).cuda())
            ld64 = gan.d_step(t(G[p + 'zd']).double(), t(G[p + 'real']).double(), t(G[p + 'eps_interp']).double())
            note('d.', gan.d)
            assert_close(ld.cpu(), G[p + 'loss_d'], 2e-3, f'loss_d {it}.{di}')
            assert_close(ld.cpu().double(), ld64, 2e-3, f'loss_d {it}.{di} (f64)')
    n_checked = 0
    for pre, tag, model, ref in (('g0.', 'g.', L.gen_model, {**gan.g, **gan.g_buf}), ('d0.', 'd.', L.disc_model, gan.d)):
        cur = dict(model.state_dict())
        ref0 = sub(G, pre)
        fix1 = sub(G, pre[0] + '1.')
        for k, v in ref.items():
            v = v.detach()
            if k.endswith('num_batches_tracked'):
                assert int(cur[k]) == int(v) == int(fix1[k]), k
                continue
            if 'running' in k:
                # the zero-gradient conv biases random-walk by +-lr per step and shift these means by O(lr)
                assert_close(cur[k].cpu().double(), v, 2e-2, 'running ' + k)
                assert_close(cur[k].cpu(), fix1[k], 2e-2, 'running (fixture) ' + k)
                continue
            if tag == 'g.' and resnet_zero_grad_key(k):
                continue
            du_ref, du = v - ref0[k].double(), cur[k].detach().cpu().double() - ref0[k].double()
            if tag + k not in ok:      # exactly-zero gradient (the critic's last bias cancels in the WGAN loss)
                assert du_ref.abs().max() == 0 and du.abs().max() == 0, k
                continue
            m = ok[tag + k]
            if m.float().mean() > 0.5:
                # Adam(beta1=0) steps by ~lr*sign(g); masked noise elements moved by +-lr in both runs, so
                # from the second iteration on a gradient element within ~1e-3 of zero may legitimately
                # flip: 99.5% of the elements must be within 3% of the largest update.
                bad = (du[m] - du_ref[m]).abs() > 3e-2 * du_ref[m].abs().max()
                assert bad.float().mean() <= 5e-3, f'update {pre}{k}: {int(bad.sum())}/{bad.numel()} off'
                n_checked += int(m.sum())
    assert n_checked > 1000


@pytest.mark.parametrize('res', [32, 64])
def test_resnet_paired_critic_pass_equals_two_passes(res, monkeypatch):
    """The critic iteration scores [generated; real] in ONE pass (every critic layer is per-sample); the loss, every
    critic gradient and the update equal those of the reference's two passes (resnetgan/learner.py:640-651) up to the
    order of the weight-gradient sums."""
    G = load_golden(f'resnet{res}.npz')
    out = {}
    for pair in ('0', '1'):
        monkeypatch.setenv('GANLAB_RESNET_PAIR', pair)
        cfg, Learner = make_learner(res, len_latent=int(G['len_latent']), lr_base=float(G['lr']))
        cfg.fmap_g, cfg.fmap_d = int(G['fmap_g']), int(G['fmap_d'])
        L = Learner(cfg)
        L.gen_model.load_state_dict(sub(G, 'g0.'))
        L.disc_model.load_state_dict(sub(G, 'd0.'))
        L.gen_model.train()
        L.disc_model.train()
        L.set_requires_grad_disc(True)
        p = 'i0.d0.'
        real, zd, eps = t(G[p + 'real']).cuda(), t(G[p + 'zd']).cuda(), t(G[p + 'eps_interp']).cuda()
        assert L._pair_critic_batches(torch.empty_like(real), real) == (pair == '1')
        ld = L.d_step(real, zb=zd, eps_interp=eps)
        out[pair] = (ld.cpu(), L.arena_d.gflat.detach().cpu().clone(),
                     {k: v.detach().cpu().clone() for k, v in L.disc_model.named_parameters()},
                     {k: v.grad.detach().cpu().clone() for k, v in L.disc_model.named_parameters()})
    assert_close(out['1'][0], out['0'][0], 1e-6, 'loss_d')
    assert_close(out['1'][1], out['0'][1], 1e-5, 'critic gradients')
    assert out['0'][1].abs().max() > 0
    checked = 0
    for k, v in out['0'][2].items():
        # Adam(beta1=0) moves an element by ~lr*sign(g): elements whose gradient is at rounding level may flip, the rest
        # must have moved alike
        g = out['0'][3][k]
        m = g.abs() > 1e-3 * g.abs().max()
        if g.abs().max() == 0 or not m.any():
            continue
        assert_close(out['1'][3][k], g, 1e-5, 'gradient ' + k)
        du0, du1 = (v - sub(G, 'd0.')[k])[m], (out['1'][2][k] - sub(G, 'd0.')[k])[m]
        assert (du1 - du0).abs().max() <= 1e-2 * du0.abs().max(), k
        checked += int(m.sum())
    assert checked > 1000


def test_resnet_full_width_step_vs_oracle():
    """Config #5 at its real width (fmap 64, 64x64, latent 128), batch 8: one generator iteration and one
    critic iteration (WGAN + WGAN-GP) against the oracle on the same weights and draws.  The oracle runs
    in float64, with the same step in CPU fp32 beside it as the yardstick of what fp32 can deliver here
    (BatchNorm backward sums that cancel; WGAN-GP through 8 LayerNorms: ~5e-3 on the CPU).
    lr = 0: Adam(beta1=0) moves every element by lr*sign(g), so after ONE update two implementations
    already differ in every element whose gradient is within rounding of zero (thousands at this width)
    and the next step is no longer comparable elementwise; the update itself is covered by
    test_resnet_training_iterations_match_reference."""
    from oracle import resnet
    torch.manual_seed(3)
    cfg, Learner = make_learner(64, batch=8, lr_base=0.)
    L = Learner(cfg)
    with torch.no_grad():
        for m in (L.gen_model, L.disc_model):
            for k, p in m.named_parameters():
                if k.endswith('bias'):
                    p.copy_(torch.randn_like(p) * 0.1)
    L.gen_model.train()
    L.disc_model.train()
    sd_g = {k: v.detach().cpu().clone() for k, v in L.gen_model.state_dict().items()}
    sd_d = {k: v.detach().cpu().clone() for k, v in L.disc_model.state_dict().items()}
    dbl = lambda sd: {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}  # noqa: E731
    gan = resnet.ResnetFunctionalGAN(dbl(sd_g), dbl(sd_d), 64, lr=cfg.lr_base)
    gan32 = resnet.ResnetFunctionalGAN(sd_g, sd_d, 64, lr=cfg.lr_base)     # what fp32 on the CPU achieves
    zg, zd = torch.randn(8, 128), torch.randn(8, 128)
    real, eps = torch.rand(8, 3, 64, 64) * 2 - 1, torch.rand(8, 1, 1, 1)
    L.set_requires_grad_disc(False)
    lg = L.g_step(zb=zg.cuda())
    lg_ref = gan.g_step(zg.double())
    gan32.g_step(zg)
    assert_close(lg.cpu().double(), lg_ref, TOL, 'loss_g')
    gmax = max(p.grad.abs().max().item() for p in gan.g.values())
    named = dict(L.gen_model.named_parameters())
    bad_g = {}
    for k, p in gan.g.items():
        den = gmax if resnet_zero_grad_key(k) else max(p.grad.abs().max().item(), 1e-3 * gmax)
        e = (named[k].grad.cpu().double() - p.grad).abs().max().item() / den
        e32 = (gan32.g[k].grad.double() - p.grad).abs().max().item() / den
        # behind a BatchNorm backward the per-channel gradient sums cancel almost exactly, so fp32 itself
        # (e32: the same step on the CPU in fp32) is only good to ~1e-3 on some parameters
        if e > max(TOL, 5 * e32):
            # ... unless it is the footprint of a ReLU tie (see the critic below): one activation within an fp32 ulp
            # of zero lands on the other side, which moves ONE bias element / ONE output channel's filter of the
            # layers next to it by O(1e-2) and nothing else - isolated (<= 0.5% of the elements) with the bulk (L1)
            # error still within tolerance
            err = (named[k].grad.cpu().double() - p.grad).abs()
            frac = (err > TOL * den).double().mean().item()
            l1 = err.sum().item() / max(p.grad.abs().sum().item(), 1e-3 * gmax * p.numel())
            # (measured when the 8-channel staging order of the tiny-geometry conv path changed the rounding: exactly
            #  one of 512 channels off, frac = 1/512 = 1.95e-3, L1 1.1e-3 .. 1.3e-3 - that one channel IS the L1 error)
            # (seed sweep, seeds 3 / 11 / 12 / 13 / 14 with and without the split-K conv path: every seed but one has a few
            #  such entries on BOTH paths - max error 4e-3 .. 2e-2 where fp32 on the CPU has 2e-4 .. 1.4e-3, L1 error
            #  8e-4 .. 3.3e-3, up to 8 of 512 channels - so "how many elements" depends on which activations happen to sit
            #  next to zero for that seed and rounding order; the bulk (L1) error is the stable quantity and is what is
            #  bounded here.  A wrong kernel shows up as an L1 error of order 1.)
            if not (frac <= 2e-2 and l1 <= 3 * TOL):
                bad_g[k] = (f'{e:.3e}', f'cpu fp32 {e32:.3e}', tuple(p.shape), f'frac {frac:.2e}', f'l1 {l1:.2e}')
    assert not bad_g, f'G grads: {bad_g}'
    L.set_requires_grad_disc(True)
    ld = L.d_step(real.cuda(), zb=zd.cuda(), eps_interp=eps.cuda())
    ld_ref = gan.d_step(zd.double(), real.double(), eps.double())
    gan32.d_step(zd, real, eps)
    assert_close(ld.cpu().double(), ld_ref, TOL, 'loss_d')
    # ReLU ties: with ~4M activations per critic pass a handful sit within one fp32 ulp of zero, and two
    # fp32 implementations (or fp32 vs fp64) put them on different sides; each flip changes the gradient in a
    # small neighbourhood by O(1) of its value (the CPU fp32 run shows the same isolated outliers against
    # float64).  So: at most 2% of the elements of any parameter may be off by more than 1e-3 of the
    # parameter's largest gradient, and the bulk (L1) error must be within 1e-3.
    gmax = max(p.grad.abs().max().item() for p in gan.d.values())
    named = dict(L.disc_model.named_parameters())
    for k, p in gan.d.items():
        if p.grad.abs().max() == 0:
            continue
        den = max(p.grad.abs().max().item(), 1e-3 * gmax)
        err = (named[k].grad.cpu().double() - p.grad).abs()
        frac = (err > TOL * den).double().mean().item()
        l1 = err.sum().item() / max(p.grad.abs().sum().item(), 1e-3 * gmax * p.numel())
        # (measured on resblocks.1's first LayerNorm bias, the worst entry: 0.72% / 0.83% / 1.14% of its elements for three
        #  roundings of the SAME step - critic scored in one or two passes, the generator's 1x1 skip before or after its
        #  upsample, i.e. generated images that differ by 1e-7 - with the L1 error at 4.4e-5 .. 5.4e-5 every time: the count
        #  moves with the rounding order, the bulk error is the stable quantity; same 2% cap as for the generator above)
        assert frac <= 2e-2 and l1 <= TOL, f'D grad {k}: {frac:.2e} of elements off, L1 rel err {l1:.2e}'


def test_resnet_train_loop_and_checkpoint(tmp_path):
    from gan_lab_amd.utils.data_utils import SyntheticImageLoader
    cfg, Learner = make_learner(32, batch=4, len_latent=16, num_disc_iters=2, lr_sched='linear decay')
    cfg.fmap_g = cfg.fmap_d = 8
    Ls = Learner(cfg)
    Ls.train(SyntheticImageLoader(64, 4, 32), num_main_iters=4)
    assert abs(Ls.opt_gen.param_groups[0]['lr']) < 1e-12 and abs(Ls.opt_disc.param_groups[0]['lr']) < 1e-12
    cfg, Learner = make_learner(32, batch=4, len_latent=16, num_disc_iters=2)
    cfg.fmap_g = cfg.fmap_d = 8
    L = Learner(cfg)
    dl = SyntheticImageLoader(64, 4, 32)
    L.log_every = 1
    L.train(dl, num_main_iters=3)
    assert np.isfinite(L.last_losses['loss_d']) and np.isfinite(L.last_losses['loss_g'])
    assert L.curr_img_num == 3 * 2 * 4 and not L.not_trained_yet
    path = tmp_path / 'resnetgan_model.tar'
    L.save_model(path)
    L2 = Learner(cfg)
    L2.load_model(path)
    for a, b in ((L.gen_model, L2.gen_model), (L.disc_model, L2.disc_model)):
        for (k, v), (_, v2) in zip(a.state_dict().items(), b.state_dict().items()):
            assert torch.equal(v, v2), k
    # the restored Adam moments reproduce the next update exactly
    z, x = torch.randn(4, 16).cuda(), (torch.rand(4, 3, 32, 32) * 2 - 1).cuda()
    e = torch.rand(4, 1, 1, 1).cuda()
    for lr in (L, L2):
        lr.gen_model.train()
        lr.disc_model.train()
        lr.set_requires_grad_disc(True)
        lr.d_step(x, zb=z, eps_interp=e)
    assert rel_err(L2.arena_d.flat.cpu(), L.arena_d.flat.cpu()) < 1e-6
    img = L.gen_model(torch.randn(4, 16).cuda())
    assert img.shape == (4, 3, 32, 32) and torch.isfinite(img).all() and img.abs().max() <= 1
