"""Pins the oracle (oracle/) against golden vectors captured from the imported reference
(tests/golden/make_golden.py).  CPU only.  Tolerance: 1e-5 relative (same ATen kernels, different
op order only), far inside the 1e-3 the north star allows the product."""
import numpy as np
import pytest
import torch

from oracle import nets, ops, step
from util import assert_close, load_golden, resnet_zero_grad_key, sub, t

TOL = 2e-5


@pytest.fixture(scope='module')
def G():
    return load_golden('ops.npz')


def test_wscale_tables(G):
    for ni, nf, ks, gain, ws in G['wscale_conv_table']:
        w = torch.empty(int(nf), int(ni), int(ks), int(ks))
        assert abs(ops.conv_wscale(w, gain) - ws) < 1e-12
    for ni, nf, gain, ws in G['wscale_linear_table']:
        assert abs(ops.linear_wscale(torch.empty(int(nf), int(ni)), gain) - ws) < 1e-12
    # known answers quoted in SURVEY.md §8a A2
    assert abs(ops.he_std(512 * 9, 2.0) - 0.020833333) < 1e-8
    assert abs(ops.he_std(512, 1.0) - 0.044194174) < 1e-8
    assert abs(ops.he_std(512, 2.0) - 0.0625) < 1e-12
    assert abs(ops.he_std(512, 2.0 / 16) - 0.015625) < 1e-12


def test_conv2d_ex(G):
    x = t(G['conv_x']).requires_grad_(True)
    w = t(G['conv_w']).requires_grad_(True)
    b = t(G['conv_b']).requires_grad_(True)
    y = ops.conv2d_ex(x, w, b, ops.conv_wscale(w, 2.0), padding=1)
    assert_close(y, G['conv_y'], TOL, 'y')
    (y * t(G['conv_cot'])).sum().backward()
    assert_close(x.grad, G['conv_gx'], TOL, 'gx')
    assert_close(w.grad, G['conv_gw'], TOL, 'gw')
    assert_close(b.grad, G['conv_gb'], TOL, 'gb')
    w4 = t(G['conv4_w'])
    assert_close(ops.conv2d_ex(t(G['conv4_x']), w4, t(G['conv4_b']), ops.conv_wscale(w4, 2.0)),
                 G['conv4_y'], TOL, 'conv4')
    w1 = t(G['conv1_w'])
    assert_close(ops.conv2d_ex(t(G['conv1_x']), w1, t(G['conv1_b']), ops.conv_wscale(w1, 1.0)),
                 G['conv1_y'], TOL, 'conv1')


def test_linear_ex_lrmul(G):
    x = t(G['lin_x']).requires_grad_(True)
    w = t(G['lin_w']).requires_grad_(True)
    b = t(G['lin_b']).requires_grad_(True)
    y = ops.linear_ex(x, w, b, ops.linear_wscale(w, 2.0), lrmul=0.01)
    assert_close(y, G['lin_y'], TOL, 'y')
    (y * t(G['lin_cot'])).sum().backward()
    assert_close(x.grad, G['lin_gx'], TOL, 'gx')
    assert_close(w.grad, G['lin_gw'], TOL, 'gw')
    assert_close(b.grad, G['lin_gb'], TOL, 'gb')


@pytest.mark.parametrize('name,fn', [('blur', ops.blur_binomial), ('pn', ops.pixelnorm),
                                     ('in', ops.instancenorm)])
def test_unary_ops(G, name, fn):
    x = t(G[f'{name}_x']).requires_grad_(True)
    y = fn(x)
    assert_close(y, G[f'{name}_y'], TOL, name)
    (y * t(G[f'{name}_cot'])).sum().backward()
    assert_close(x.grad, G[f'{name}_gx'], 5e-5, name + ' grad')


@pytest.mark.parametrize('tag', ['mb8', 'mb6'])
def test_mbstd_first_and_second_order(G, tag):
    x = t(G[f'{tag}_x']).requires_grad_(True)
    y = ops.mbstd_concat(x, 4)
    assert_close(y, G[f'{tag}_y'], TOL, 'y')
    gx, = torch.autograd.grad((y * t(G[f'{tag}_cot'])).sum(), x, create_graph=True)
    assert_close(gx, G[f'{tag}_gx'], TOL, 'gx')
    ggx, = torch.autograd.grad((gx * t(G[f'{tag}_cot2'])).sum(), x)
    assert_close(ggx, G[f'{tag}_ggx'], 1e-4, 'ggx')


def test_mbstd_known_answers():
    # constant tensor -> std channel == sqrt(1e-8); group of 1 -> zeros
    y = ops.mbstd_concat(torch.full((4, 3, 4, 4), 2.5), 4)
    assert torch.allclose(y[:, 3], torch.full((4, 4, 4), 1e-4), rtol=1e-6)
    assert ops.mbstd_concat(torch.randn(1, 3, 4, 4), 4)[:, 3].abs().max() == 0
    y = ops.pixelnorm(torch.ones(2, 5, 3, 3))
    assert torch.allclose(y, torch.full_like(y, 1 / np.sqrt(1 + 1e-8)))


def test_noise_and_losses(G):
    assert_close(ops.add_noise(t(G['noise_x']), t(G['noise_w']), t(G['noise_n'])), G['noise_y'], TOL)
    a, b = t(G['loss_a']), t(G['loss_b'])
    assert_close(ops.loss_disc('wgan', a, b), G['loss_wgan_d'], TOL)
    assert_close(ops.loss_gen('wgan', a), G['loss_wgan_g'], TOL)
    assert_close(ops.loss_gen('nonsaturating', a), G['loss_ns_g'], TOL)
    assert_close(ops.loss_gen('minimax', a), G['loss_mm_g'], TOL)
    assert_close(ops.loss_disc('minimax', a, b), G['loss_mm_d'], TOL)
    assert_close(ops.loss_disc('nonsaturating', a, b), G['loss_mm_d'], TOL)


# ---------------------------------------------------------------------------------------------- #
NETS = ['stylegan_stab16', 'stylegan_fade16', 'stylegan_stab32', 'stylegan_stab4', 'stylegan_r2_8', 'progan_stab16',
        'progan_fade8', 'stylegan_bilinear16', 'progan_nearest16', 'stylegan_bilinear8']


def _resample_cfg(g):
    """The fixture's (model_upsample_type, model_downsample_type, align_corners), absent = the defaults."""
    if 'resample' not in g:
        return {}
    up, down, align = [str(s) for s in g['resample']]
    return dict(upsample=up, downsample=down, align_corners=bool(int(align)))


def _run_gen(kind, sd, z, noise, cfg, alpha, fade):
    if kind == 'stylegan':
        return nets.stylegen_forward(sd, z, noise, cfg, alpha, fade)
    return nets.progen_forward(sd, z, cfg, alpha, fade)


@pytest.mark.parametrize('name', NETS)
def test_whole_nets_forward_backward_gp(name):
    g = load_golden(name + '.npz')
    kind, loss, gp = [str(s) for s in g['meta']]
    cfg = nets.make_cfg(use_pixelnorm=(kind == 'progan'), **_resample_cfg(g))
    alpha, fade = float(g['alpha']), bool(g['fade_in'])
    sd_g = {k: v.requires_grad_(True) for k, v in sub(g, 'g.').items()}
    sd_d = {k: v.requires_grad_(True) for k, v in sub(g, 'd.').items()}
    z, real = t(g['z']), t(g['real'])
    noise = None
    if kind == 'stylegan':
        noise = [t(g[f'noise{i}']) for i in range(nets.stylegen_num_layers(sd_g))]
    img = _run_gen(kind, sd_g, z, noise, cfg, alpha, fade)
    assert_close(img, g['img'], TOL, 'img')
    sd_d_frozen = {k: v.detach() for k, v in sd_d.items()}
    dout = nets.disc_forward(sd_d_frozen, img, cfg, alpha, fade)
    assert_close(dout, g['d_of_img'], TOL, 'D(G(z))')
    lg = ops.loss_gen(loss, dout)
    assert_close(lg, g['loss_g'], TOL, 'loss_g')
    lg.backward()
    ref_gg = sub(g, 'gg.')
    assert ref_gg, 'no G grads in fixture'
    for k, v in ref_gg.items():
        assert_close(sd_g[k].grad, v, 2e-4, 'G grad ' + k)
    for k, p in sd_g.items():
        if k not in ref_gg:
            assert p.grad is None or p.grad.abs().max() == 0, k
    # D step loss parts and gradients
    fake = img.detach()
    total, parts = step.d_loss(sd_d, cfg, fake, real, loss, gp, 10.0, 1.0, 0.001, alpha, fade,
                               t(g['eps_interp']), return_parts=True)
    assert_close(parts['adv'], g['loss_d_adv'], TOL, 'adv')
    assert_close(parts['gp'], g['gp'], 1e-4, 'gp')
    assert_close(total, g['loss_d'], 1e-4, 'loss_d')
    total.backward()
    for k, v in sub(g, 'gd.').items():
        assert_close(sd_d[k].grad, v, 5e-4, 'D grad ' + k)
    # GP-only double backward
    for p in sd_d.values():
        p.grad = None

    def D(x):
        return nets.disc_forward(sd_d, x, cfg, alpha, fade)
    step.calc_gp(D, gp, fake, real, 10.0, 1.0, t(g['eps_interp'])).backward()
    ref = sub(g, 'ggp.')
    assert ref
    for k, v in ref.items():
        assert_close(sd_d[k].grad, v, 5e-4, 'GP grad ' + k)


def test_stylegan_mixing_regularisation():
    g = load_golden('stylegan_mixing16.npz')
    cfg = nets.make_cfg()
    sd = sub(g, 'g.')
    noise = [t(g[f'noise{i}']) for i in range(nets.stylegen_num_layers(sd))]
    img, w = nets.stylegen_forward(sd, t(g['z']), noise, cfg, cutoff_idx=int(g['cutoff_idx']),
                                   z_mix=t(g['z_mix']), return_w=True)
    assert_close(img, g['img'], TOL, 'mixed img')
    # first training call: w_ewma = mean_b(w) (stylegan/architectures.py:429-430)
    assert_close(w.mean(dim=0), g['w_ewma'], TOL, 'w_ewma')
    # and a different cutoff must give a different image (the fixture pins the cutoff semantics)
    other = 1 if int(g['cutoff_idx']) != 1 else 2
    img2 = nets.stylegen_forward(sd, t(g['z']), noise, cfg, cutoff_idx=other, z_mix=t(g['z_mix']))
    assert (img2 - img).abs().max() > 1e-4


@pytest.mark.parametrize('name', ['step_stylegan16', 'step_stylegan8_fade', 'step_progan8'])
def test_training_steps(name):
    g = load_golden(name + '.npz')
    kind, loss, gp = [str(s) for s in g['meta']]
    cfg = nets.make_cfg(use_pixelnorm=(kind == 'progan'))
    alpha, fade = float(g['alpha']), bool(g['fade_in'])
    gan = step.FunctionalGAN(sub(g, 'g0.'), sub(g, 'd0.'), cfg, model=kind, loss=loss, gp=gp,
                             lr=float(g['lr']))
    # Adam (beta1 = 0) turns a numerically-zero gradient (|g| ~ 1e-8: e.g. a bias in front of an
    # InstanceNorm on a channel whose activations never change sign) into an O(lr) update whose
    # sign is rounding noise - such elements are not comparable between ANY two implementations,
    # so they are masked out by gradient magnitude.
    ok = {}

    def note(params):
        for k, p in params.items():
            if p.grad is not None:
                m = p.grad.abs() > 1e-5 * p.grad.abs().max().clamp_min(1e-30)
                ok[k] = m if k not in ok else (ok[k] & m)

    for s in range(int(g['n_steps'])):
        nd = ng = None
        if kind == 'stylegan':
            L = nets.stylegen_num_layers(gan.g)
            nd = [t(g[f's{s}.nd{i}']) for i in range(L)]
            ng = [t(g[f's{s}.ng{i}']) for i in range(L)]
        ld, _ = gan.d_step(t(g[f's{s}.zd']), t(g[f's{s}.real']), nd, alpha, fade,
                           eps_interp=t(g[f's{s}.eps_interp']))
        note({'d.' + k: v for k, v in gan.d.items()})
        assert_close(ld, g[f's{s}.loss_d'], 2e-4, f'loss_d step {s}')
        lg = gan.g_step(t(g[f's{s}.zg']), ng, alpha, fade, beta=float(g['beta']))
        note({'g.' + k: v for k, v in gan.g.items()})
        assert_close(lg, g[f's{s}.loss_g'], 2e-4, f'loss_g step {s}')
    # Adam moves every element by ~lr per step, so compare the *update* (p1 - p0), not p1
    n_checked = 0
    for pre, tag, cur, ref0 in (('g1.', 'g.', gan.g, sub(g, 'g0.')), ('d1.', 'd.', gan.d, sub(g, 'd0.'))):
        for k, v in sub(g, pre).items():
            du_ref = v - ref0[k]
            du = cur[k].detach() - ref0[k]
            if du_ref.abs().max() == 0:
                assert du.abs().max() == 0, k
            else:
                m = ok[tag + k]
                assert m.float().mean() > 0.5, k
                assert_close(du[m], du_ref[m], 2e-2, 'update ' + pre + k)
                n_checked += int(m.sum())
    assert n_checked > 1000
    for k, v in sub(g, 'lag.').items():
        m = ok['g.' + k] if 'g.' + k in ok else torch.ones_like(v, dtype=torch.bool)
        assert_close(gan.lagged[k][m], v[m], 1e-5, 'ewma ' + k)


# ---------------------------------------------------------------------------------------------- #
# ResNet GAN (config #5): BatchNorm generator, LayerNorm critic, WGAN + WGAN-GP
# ---------------------------------------------------------------------------------------------- #
def _grad_close(got, ref_by_key, tol, what):
    """Per-parameter relative error.  The conv biases inside the generator's residual blocks all
    feed a BatchNorm (directly, or through the residual sum), so their true gradient is zero and
    both sides only hold rounding noise: those are judged against the largest gradient of the net."""
    gmax = max(float(np.abs(v).max()) for v in ref_by_key.values())
    for k, ref in ref_by_key.items():
        a = got[k].grad.detach().double()
        b = torch.from_numpy(ref).double()
        den = gmax if resnet_zero_grad_key(k) else max(b.abs().max().item(), 1e-4 * gmax)
        e = (a - b).abs().max().item() / den
        assert e <= tol, f'{what} {k}: rel err {e:.3e} > {tol:.1e}'


RESNET_CASES = [(32, None), (64, None), (32, 'tanh')]      # (resolution, --nonlinearity other than the default ReLU)
RESNET_IDS = ['32', '64', '32-tanh']


def _resnet_golden(res, nl):
    return load_golden(f'resnet{res}.npz' if nl is None else f'resnet{res}_{nl}.npz')


@pytest.mark.parametrize('res,nl', RESNET_CASES, ids=RESNET_IDS)
def test_resnet_nets_forward_backward_gp(res, nl):
    from oracle import resnet
    g = _resnet_golden(res, nl)
    gan = resnet.ResnetFunctionalGAN(sub(g, 'g0.'), sub(g, 'd0.'), res, lr=float(g['lr']), nl=nl)
    img = gan.gen(t(g['z']))
    assert_close(img, g['img'], TOL, 'img')
    for k, v in sub(g, 'g_after_fwd.').items():
        assert_close(gan.g_buf[k], v, TOL, 'running stat ' + k)
    dout = gan.disc(img, {k: v.detach() for k, v in gan.d.items()})
    assert_close(dout, g['d_of_img'], TOL, 'D(G(z))')
    (-dout.mean()).backward()
    _grad_close(gan.g, {k[3:]: v for k, v in g.items() if k.startswith('gg.')}, 1e-4, 'G grad')
    fake, real, eps = img.detach(), t(g['real']), t(g['eps_interp'])
    gpv = step.calc_gp(gan.disc, 'wgan-gp', fake, real, 10.0, 1.0, eps)
    assert_close(gpv, g['gp'], 1e-4, 'gp')
    gpv.backward()
    _grad_close(gan.d, {k[4:]: v for k, v in g.items() if k.startswith('ggp.')}, 2e-4, 'GP-only grad')
    for p in gan.d.values():
        p.grad = None
    ld = gan.d_loss(fake, real, eps)
    assert_close(ld, g['loss_d'], 1e-4, 'loss_d')
    ld.backward()
    _grad_close(gan.d, {k[3:]: v for k, v in g.items() if k.startswith('gd.')}, 2e-4, 'D grad')


@pytest.mark.parametrize('res,nl', RESNET_CASES, ids=RESNET_IDS)
def test_resnet_training_iterations(res, nl):
    from oracle import resnet
    g = _resnet_golden(res, nl)
    gan = resnet.ResnetFunctionalGAN(sub(g, 'g0.'), sub(g, 'd0.'), res, lr=float(g['lr']), nl=nl)
    ok = {}

    def note(tag, params):
        for k, p in params.items():
            if p.grad is not None:
                m = p.grad.abs() > 1e-4 * p.grad.abs().max().clamp_min(1e-30)
                ok[tag + k] = m if tag + k not in ok else (ok[tag + k] & m)

    for it in range(int(g['n_iters'])):
        lg = gan.g_step(t(g[f'i{it}.zg']))
        note('g.', gan.g)
        assert_close(lg, g[f'i{it}.loss_g'], 2e-4, f'loss_g {it}')
        for di in range(int(g['n_disc'])):
            p = f'i{it}.d{di}.'
            ld = gan.d_step(t(g[p + 'zd']), t(g[p + 'real']), t(g[p + 'eps_interp']))
            note('d.', gan.d)
            assert_close(ld, g[p + 'loss_d'], 5e-4, f'loss_d {it}.{di}')
    n_checked = 0
    for pre, tag, cur, ref0 in (('g1.', 'g.', {**gan.g, **gan.g_buf}, sub(g, 'g0.')),
                                ('d1.', 'd.', gan.d, sub(g, 'd0.'))):
        for k, v in sub(g, pre).items():
            if k.endswith('num_batches_tracked'):
                assert int(cur[k]) == int(v), k
                continue
            if 'running' in k:
                # the +-lr noise updates of the zero-gradient biases shift these means by O(lr)
                assert_close(cur[k], v, 5e-3, pre + k)
                continue
            if tag == 'g.' and resnet_zero_grad_key(k):
                continue        # Adam(beta1=0) turns the rounding-noise gradient into +-lr: not comparable
            du_ref, du = v - ref0[k], cur[k].detach() - ref0[k]
            m = ok[tag + k]
            if m.float().mean() > 0.5:
                assert_close(du[m], du_ref[m], 3e-2, 'update ' + pre + k)
                n_checked += int(m.sum())
    assert n_checked > 1000


# ---------------------------------------------------------------------------------------------- #
# real-image input path: PIL BOX resize -> ToTensor -> Normalize
# ---------------------------------------------------------------------------------------------- #
def test_image_decode_oracle_matches_pil():
    from oracle import data
    g = load_golden('data_box.npz')
    for res in (64, 32, 16, 8, 4):
        assert np.array_equal(data.box_resize_u8(g['images'], res), g[f'u8_{res}']), res       # bit-exact
        assert np.array_equal(data.decode(g['images'], res, g['mean'], g['std']), g[f'x_{res}']), res
        assert np.array_equal(data.decode(g['images'], res, g['mean2'], g['std2']), g[f'x2_{res}']), res
    flip = np.array([1, 0, 1, 0, 0, 1], dtype=bool)
    x = data.decode(g['images'], 16, g['mean'], g['std'], flip)
    ref = g['x_16'].copy()
    ref[flip] = ref[flip][:, :, :, ::-1]
    assert np.array_equal(x, ref)
