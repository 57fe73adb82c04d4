#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by importing and RUNNING the upstream reference
(/root/reference, read-only) in the build container.  Usage:  python tests/golden/make_golden.py

What is stored is data only: seeded inputs, the (shrunken-width) state_dicts, and the outputs /
gradients / updated parameters the reference produced.  No reference source is copied.  The
reference has no tests or fixtures of its own, so these vectors are what pins the
oracle (oracle/) - see tests/test_oracle_golden.py.

Widths are shrunk by patching the reference's FMAP_BASE / FMAP_MAX module constants
(stylegan/base.py:16-17, progan/base.py:16-17) so the fixtures stay small.
"""
import argparse
import os
import pickle
import sys
import tempfile
import types
from abc import ABC

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refstub  # noqa: E402

ns = _refstub.import_reference_learners()
import torch  # noqa: E402
from torch import nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402

torch.set_num_threads(4)
FMAP_BASE, FMAP_MAX = 64, 16
LEN_LATENT = 16
NUM_FCS = 2


def T(x):
    return x.detach().cpu().numpy().copy()


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f'wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrs)} arrays')


def sd_arrays(prefix, module):
    return {prefix + k: T(v) for k, v in module.state_dict().items()}


def randomize_zero_params(module, gen):
    """Biases / noise weights are zero-initialised and const_input is ones in the reference
    (stylegan/architectures.py:110,199-201; custom_layers.py:198-200) - give them random values so
    the fixtures exercise them."""
    with torch.no_grad():
        for k, p in module.named_parameters():
            if k.endswith('bias') or k.endswith('noise_weight') or k == 'const_input':
                p.copy_(torch.randn(p.shape, generator=gen) * 0.5 + (1.0 if k == 'const_input' else 0.0))


# ---------------------------------------------------------------------------------------------- #
def ref_resamplers(up, down, align):
    """The generator's upsampler and the critic's pooler as resnetgan/learner.py:147-173 builds them from
    config.model_upsample_type / model_downsample_type / align_corners."""
    upsampler = nn.Upsample(scale_factor=2, mode=up, align_corners=(align if up == 'bilinear' else None))
    if down in ('average', 'box'):
        pooler = nn.AvgPool2d(kernel_size=2, stride=2)
    elif down == 'nearest':
        pooler = ns.cl.NearestPool2d()
    else:
        pooler = ns.cl.BilinearPool2d(align_corners=align)
    return upsampler, pooler


def make_nets(kind, res, blur='binomial', mbstd=4, resample=None, nl=None, **gkw):
    """Build reference G and D the way the learners do (stylegan/learner.py:114-163,
    progan/learner.py:120-160) and grow them to `res` (fade_in_phase left True)."""
    dkw = {}
    if resample is not None:
        gkw['upsampler'], dkw['pooler'] = ref_resamplers(*resample)
    if nl == 'tanh':                       # --nonlinearity tanh (config.py:208; the learners hand nn.Tanh() to both networks)
        gkw['nl'], dkw['nl'] = nn.Tanh(), nn.Tanh()
    if kind == 'stylegan':
        ns.sb.FMAP_BASE, ns.sb.FMAP_MAX = FMAP_BASE, FMAP_MAX
        Base = type('StyleGAN', (nn.Module, ABC,), dict(ns.sb.StyleGAN.__dict__))
        Base.reset_state()
        G = type('StyleGenerator', (Base,), dict(ns.sa.StyleGenerator.__dict__))
        D = type('StyleDiscriminator', (Base,), dict(ns.pa.ProDiscriminator.__dict__))
        kw = dict(final_res=64, len_latent=LEN_LATENT, len_dlatent=LEN_LATENT, mapping_num_fcs=NUM_FCS,
                  blur_type=blur, truncation_trick_params={'beta': .995, 'psi': .7, 'cutoff_stage': 4})
        kw.update(gkw)
        g = G(**kw)
    else:
        ns.pb.FMAP_BASE, ns.pb.FMAP_MAX = FMAP_BASE, FMAP_MAX
        Base = type('ProGAN', (nn.Module, ABC,), dict(ns.pb.ProGAN.__dict__))
        Base.reset_state()
        G = type('ProGenerator', (Base,), dict(ns.pa.ProGenerator.__dict__))
        D = type('ProDiscriminator', (Base,), dict(ns.pa.ProDiscriminator.__dict__))
        kw = dict(final_res=64, len_latent=LEN_LATENT, blur_type=blur)
        kw.update(gkw)
        g = G(**kw)
    d = D(final_res=64, blur_type=blur, mbstd_group_size=mbstd, **dkw)
    for _ in range(int(np.log2(res)) - 2):
        g.increase_scale()
        d.increase_scale()
    return g, d


def set_phase(g, fade_in, alpha):
    g.fade_in_phase = fade_in
    if fade_in:
        g.alpha = alpha
    else:
        g.alpha = 1


def gen_noise(g, b, gen):
    return [torch.randn(b, 1, 4 * 2 ** ((n) // 2), 4 * 2 ** ((n) // 2), generator=gen)
            for n in range(len(g.gen_layers))]


# ---------------------------------------------------------------------------------------------- #
def golden_ops():
    gen = torch.Generator().manual_seed(1234)
    cl = ns.cl
    out = {}

    def rnd(*s):
        return torch.randn(*s, generator=gen)

    # A1 Conv2dEx eq-LR 3x3 with bias
    torch.manual_seed(1)
    m = cl.Conv2dEx(ni=6, nf=5, ks=3, padding=1, init='He', init_type='StyleGAN', gain_sq_base=2.,
                    equalized_lr=True)
    m.conv2d.bias.data = rnd(5)
    x = rnd(3, 6, 7, 7).requires_grad_(True)
    cot = rnd(3, 5, 7, 7)
    y = m(x)
    (y * cot).sum().backward()
    out.update(conv_x=T(x), conv_w=T(m.conv2d.weight), conv_b=T(m.conv2d.bias), conv_cot=T(cot),
               conv_y=T(y), conv_gx=T(x.grad), conv_gw=T(m.conv2d.weight.grad),
               conv_gb=T(m.conv2d.bias.grad), conv_wscale=np.float64(m.wscale))
    # A1 4x4 valid conv, 1x1 toRGB (gain 1)
    torch.manual_seed(2)
    m = cl.Conv2dEx(ni=4, nf=3, ks=4, padding=0, init='He', init_type='ProGAN', gain_sq_base=2.,
                    equalized_lr=True)
    x = rnd(2, 4, 4, 4)
    out.update(conv4_x=T(x), conv4_w=T(m.conv2d.weight), conv4_b=T(m.conv2d.bias), conv4_y=T(m(x)),
               conv4_wscale=np.float64(m.wscale))
    m = cl.Conv2dEx(ni=8, nf=3, ks=1, padding=0, init='He', init_type='StyleGAN', gain_sq_base=1.,
                    equalized_lr=True)
    x = rnd(2, 8, 5, 5)
    out.update(conv1_x=T(x), conv1_w=T(m.conv2d.weight), conv1_b=T(m.conv2d.bias), conv1_y=T(m(x)),
               conv1_wscale=np.float64(m.wscale))
    # A3 LinearEx with lrmul
    torch.manual_seed(3)
    m = cl.LinearEx(nin_feat=16, nout_feat=12, init='He', init_type='StyleGAN', gain_sq_base=2.,
                    equalized_lr=True, lrmul=.01)
    m.linear.bias.data = rnd(12) * 10
    x = rnd(5, 16).requires_grad_(True)
    cot = rnd(5, 12)
    y = m(x)
    (y * cot).sum().backward()
    out.update(lin_x=T(x), lin_w=T(m.linear.weight), lin_b=T(m.linear.bias), lin_cot=T(cot), lin_y=T(y),
               lin_gx=T(x.grad), lin_gw=T(m.linear.weight.grad), lin_gb=T(m.linear.bias.grad),
               lin_wscale=np.float64(m.wscale))
    # A2 wscale table (known answers from initializer.py)
    tab = []
    for (ni, nf, ks, gain) in [(512, 512, 3, 2.), (512, 3, 1, 1.), (513, 512, 3, 2.), (512, 512, 4, 2.),
                               (3, 16, 1, 2.), (32, 16, 3, 2.)]:
        m = cl.Conv2dEx(ni=ni, nf=nf, ks=ks, init='He', init_type='StyleGAN', gain_sq_base=gain,
                        equalized_lr=True)
        tab.append([ni, nf, ks, gain, m.wscale])
    out['wscale_conv_table'] = np.array(tab, dtype=np.float64)
    tab = []
    for (ni, nf, gain) in [(512, 512, 2.), (512, 1024, 1.), (512, 8192, 2. / 16), (512, 1, 1.)]:
        m = cl.LinearEx(nin_feat=ni, nout_feat=nf, init='He', init_type='ProGAN', gain_sq_base=gain,
                        equalized_lr=True)
        tab.append([ni, nf, gain, m.wscale])
    out['wscale_linear_table'] = np.array(tab, dtype=np.float64)
    # A4 blur
    x = rnd(2, 5, 9, 9).requires_grad_(True)
    cot = rnd(2, 5, 9, 9)
    y = cl.get_blur_op('binomial', 5)(x)
    (y * cot).sum().backward()
    out.update(blur_x=T(x), blur_cot=T(cot), blur_y=T(y), blur_gx=T(x.grad))
    # A5 PixelNorm
    x = rnd(3, 7, 4, 4).requires_grad_(True)
    cot = rnd(3, 7, 4, 4)
    y = cl.PixelNorm2d()(x)
    (y * cot).sum().backward()
    out.update(pn_x=T(x), pn_cot=T(cot), pn_y=T(y), pn_gx=T(x.grad))
    # A6 InstanceNorm
    x = (rnd(3, 4, 6, 6) * 3 + 1).requires_grad_(True)
    cot = rnd(3, 4, 6, 6)
    y = cl.NormalizeLayer('InstanceNorm')(x)
    (y * cot).sum().backward()
    out.update(in_x=T(x), in_cot=T(cot), in_y=T(y), in_gx=T(x.grad))
    # A13 mbstd: B=8 gs=4, and the B % gs != 0 fallback (B=6), first and second derivatives
    for tag, b in (('mb8', 8), ('mb6', 6)):
        x = rnd(b, 5, 4, 4).requires_grad_(True)
        cot = rnd(b, 6, 4, 4)
        y = cl.concat_mbstd_layer(x, 4)
        gx, = torch.autograd.grad((y * cot).sum(), x, create_graph=True)
        cot2 = rnd(b, 5, 4, 4)
        ggx, = torch.autograd.grad((gx * cot2).sum(), x)
        out.update({f'{tag}_x': T(x), f'{tag}_cot': T(cot), f'{tag}_y': T(y), f'{tag}_gx': T(gx),
                    f'{tag}_cot2': T(cot2), f'{tag}_ggx': T(ggx)})
    # A7 StyleAddNoise (eval mode honours the supplied noise) + A8 AdaIN via a tiny explicit chain
    m = ns.sa.StyleAddNoise(nf=4).eval()
    m.noise_weight.data = rnd(1, 4, 1, 1)
    x, nz = rnd(2, 4, 5, 5), rnd(2, 1, 5, 5)
    out.update(noise_x=T(x), noise_w=T(m.noise_weight), noise_n=T(nz), noise_y=T(m(x, noise=nz)))
    # losses (backprop_utils.py:19-49) on (B,) logits
    a, b_ = rnd(8), rnd(8)
    out.update(loss_a=T(a), loss_b=T(b_),
               loss_wgan_d=T(ns.bp.wasserstein_distance_disc(a, b_)),
               loss_wgan_g=T(ns.bp.wasserstein_distance_gen(a)),
               loss_ns_g=T(ns.bp.nonsaturating_loss_gen(a)),
               loss_mm_g=T(ns.bp.minimax_loss_gen(a)),
               loss_mm_d=T(ns.bp.minimax_loss_disc(a, b_)))
    save('ops.npz', **out)


# ---------------------------------------------------------------------------------------------- #
def ref_calc_gp(d, kind, fake, real, lda=10., gamma=1.):
    """Call the reference's GANLearner.calc_gp (resnetgan/learner.py:780-827) unbound."""
    fake_self = types.SimpleNamespace(
        gradient_penalty=kind, batch_size=real.shape[0], disc_model=d,
        config=types.SimpleNamespace(dev=torch.device('cpu'), lda=lda, gamma=gamma))
    return ns.rl.GANLearner.calc_gp(fake_self, fake, real)


def golden_nets(kind, res, fade_in, alpha, tag, b=4, loss='nonsaturating', gp='r1', seed=0, **gkw):
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(100 + seed)
    g, d = make_nets(kind, res, **gkw)
    randomize_zero_params(g, gen)
    randomize_zero_params(d, gen)
    set_phase(g, fade_in, alpha)
    g.eval()            # eval + explicit noise == the train-mode math with that noise
    d.train()
    if kind == 'stylegan':
        g.use_truncation_trick = False
    out = {}
    out.update(sd_arrays('g.', g))
    out.update(sd_arrays('d.', d))
    z = torch.randn(b, LEN_LATENT, generator=gen)
    real = torch.rand(b, 3, res, res, generator=gen) * 2 - 1
    out.update(z=T(z), real=T(real), alpha=np.float64(g.alpha), fade_in=np.bool_(fade_in),
               res=np.int64(res))
    if kind == 'stylegan':
        noise = gen_noise(g, b, gen)
        for i, nz in enumerate(noise):
            out[f'noise{i}'] = T(nz)
        img = g(z, noise=noise)
    else:
        img = g(z)
    out['img'] = T(img)
    # G-side gradients through D (G-step, progan/learner.py:857-904)
    for p in d.parameters():
        p.requires_grad_(False)
    dout = d(img)
    out['d_of_img'] = T(dout)
    if loss == 'wgan':
        lg = -dout.mean()
    else:
        lg = F.binary_cross_entropy_with_logits(dout, torch.ones(b))
    g.zero_grad()
    lg.backward()
    out['loss_g'] = T(lg)
    for k, p in g.named_parameters():
        if p.grad is not None:
            out['gg.' + k] = T(p.grad)
    for p in d.parameters():
        p.requires_grad_(True)
    # D-side: loss + GP + drift (progan/learner.py:788-815)
    fake = img.detach()
    d.zero_grad()
    d_fake, d_real = d(fake), d(real)
    if loss == 'wgan':
        ld = (d_fake - d_real).mean()
    else:
        ld = F.binary_cross_entropy_with_logits(d_fake, torch.zeros(b)) + \
            F.binary_cross_entropy_with_logits(d_real, torch.ones(b))
    out['loss_d_adv'] = T(ld)
    torch.manual_seed(777)
    gpv = ref_calc_gp(d, gp, fake, real)
    torch.manual_seed(777)
    out['eps_interp'] = T(torch.rand(b, 1, 1, 1))
    out['gp'] = T(gpv)
    total = ld + gpv + (d_real ** 2).mean() * 0.001
    out['loss_d'] = T(total)
    total.backward()
    for k, p in d.named_parameters():
        if p.grad is not None:
            out['gd.' + k] = T(p.grad)
    # GP-only gradients (isolates the double backward)
    d.zero_grad()
    torch.manual_seed(777)
    ref_calc_gp(d, gp, fake, real).backward()
    for k, p in d.named_parameters():
        if p.grad is not None:
            out['ggp.' + k] = T(p.grad)
    out['meta'] = np.array([kind, loss, gp], dtype='U16')
    if gkw.get('resample') is not None:
        up, down, align = gkw['resample']
        out['resample'] = np.array([up, down, str(int(bool(align)))], dtype='U16')
    if gkw.get('nl') is not None:
        out['nl'] = np.array([gkw['nl']], dtype='U16')
    save(f'{tag}.npz', **out)


def golden_mixing():
    """Train-mode StyleGenerator forward with mixing regularisation (stylegan/architectures.py:
    415-422, 507-512): replay the torch RNG stream to recover cutoff_idx, noise and the second z."""
    torch.manual_seed(5)
    gen = torch.Generator().manual_seed(55)
    g, d = make_nets('stylegan', 16)
    randomize_zero_params(g, gen)
    set_phase(g, False, 1)
    g.train()
    g.pct_mixing_reg = 1.0     # force the branch (np.random.rand() < 1.0)
    g._use_mixing_reg = True
    b = 4
    z = torch.randn(b, LEN_LATENT, generator=gen)
    torch.manual_seed(4242)
    img = g(z)
    # replay
    torch.manual_seed(4242)
    L = len(g.gen_layers)
    cutoff = torch.randint(1, 2 * g.scale_stage, (1,)).item()
    noise, z2 = [], None
    for n in range(L):
        r = 4 * 2 ** (n // 2)
        noise.append(torch.randn(b, 1, r, r, dtype=torch.float32))
        if n == cutoff:
            z2 = torch.randn(b, LEN_LATENT, dtype=torch.float32)
    out = sd_arrays('g.', g)
    out.update(z=T(z), z_mix=T(z2), cutoff_idx=np.int64(cutoff), img=T(img),
               w_ewma=T(g.w_ewma))
    for i, nz in enumerate(noise):
        out[f'noise{i}'] = T(nz)
    save('stylegan_mixing16.npz', **out)


def golden_step(kind, res, tag, loss, gp, fade_in=False, alpha=1.0, b=4, lr=1e-3, n_steps=2):
    """A D-iteration + G-iteration with torch.optim.Adam(betas=(0,.99)) on the reference modules,
    every random draw explicit (progan/learner.py:734-943)."""
    torch.manual_seed(11)
    gen = torch.Generator().manual_seed(111)
    g, d = make_nets(kind, res)
    randomize_zero_params(g, gen)
    randomize_zero_params(d, gen)
    set_phase(g, fade_in, alpha)
    g.eval()
    d.train()
    if kind == 'stylegan':
        g.use_truncation_trick = False
    out = {}
    out.update(sd_arrays('g0.', g))
    out.update(sd_arrays('d0.', d))
    excl_g = [] if fade_in else ['prev_torgb.conv2d.weight', 'prev_torgb.conv2d.bias']
    excl_d = [] if fade_in else ['prev_fromrgb.0.conv2d.weight', 'prev_fromrgb.0.conv2d.bias']
    opt_g = torch.optim.Adam(g.most_parameters(excluded_params=excl_g), lr=lr, betas=(0., .99), eps=1e-8)
    opt_d = torch.optim.Adam(d.most_parameters(excluded_params=excl_d), lr=lr, betas=(0., .99), eps=1e-8)
    beta = .5 ** (b / 10000.)
    lagged = {k: p.detach().clone() for k, p in g.named_parameters()}
    for s in range(n_steps):
        zd = torch.randn(b, LEN_LATENT, generator=gen)
        zg = torch.randn(b, LEN_LATENT, generator=gen)
        real = torch.rand(b, 3, res, res, generator=gen) * 2 - 1
        out.update({f's{s}.zd': T(zd), f's{s}.zg': T(zg), f's{s}.real': T(real)})
        nd = ng = None
        if kind == 'stylegan':
            nd, ng = gen_noise(g, b, gen), gen_noise(g, b, gen)
            for i in range(len(nd)):
                out[f's{s}.nd{i}'] = T(nd[i])
                out[f's{s}.ng{i}'] = T(ng[i])
        # D step
        for p in d.parameters():
            p.requires_grad_(True)
        d.zero_grad()
        fake = (g(zd, noise=nd) if kind == 'stylegan' else g(zd)).detach()
        if fade_in:
            # real images are faded in like the generator's output (progan/learner.py:771-779)
            with torch.no_grad():
                real = F.interpolate(F.avg_pool2d(real, kernel_size=2, stride=2), scale_factor=2, mode='nearest') * \
                    (1. - g.alpha) + real * g.alpha
        d_fake, d_real = d(fake), d(real)
        if loss == 'wgan':
            ld = (d_fake - d_real).mean()
        else:
            ld = F.binary_cross_entropy_with_logits(d_fake, torch.zeros(b)) + \
                F.binary_cross_entropy_with_logits(d_real, torch.ones(b))
        torch.manual_seed(900 + s)
        ld = ld + ref_calc_gp(d, gp, fake, real)
        torch.manual_seed(900 + s)
        out[f's{s}.eps_interp'] = T(torch.rand(b, 1, 1, 1))
        ld = ld + (d_real ** 2).mean() * 0.001
        ld.backward()
        opt_d.step()
        out[f's{s}.loss_d'] = T(ld)
        # G step
        for p in d.parameters():
            p.requires_grad_(False)
        g.zero_grad()
        dout = d(g(zg, noise=ng) if kind == 'stylegan' else g(zg))
        lg = -dout.mean() if loss == 'wgan' else F.binary_cross_entropy_with_logits(dout, torch.ones(b))
        lg.backward()
        opt_g.step()
        out[f's{s}.loss_g'] = T(lg)
        with torch.no_grad():
            for k, p in g.named_parameters():
                lagged[k] = p * (1. - beta) + lagged[k] * beta
    out.update(sd_arrays('g1.', g))
    out.update(sd_arrays('d1.', d))
    out.update({'lag.' + k: T(v) for k, v in lagged.items()})
    out['meta'] = np.array([kind, loss, gp], dtype='U16')
    out.update(alpha=np.float64(g.alpha), fade_in=np.bool_(fade_in), res=np.int64(res),
               lr=np.float64(lr), beta=np.float64(beta), n_steps=np.int64(n_steps))
    save(f'{tag}.npz', **out)


# ---------------------------------------------------------------------------------------------- #
class _FakeDL:
    """Duck-typed train_dl (what ProGANLearner.train touches: SURVEY.md §8b level 1)."""

    def __init__(self, n_images, bs, res, resize_cls, gen):
        self.gen = gen
        self.batch_sampler = types.SimpleNamespace(batch_size=bs)
        tf = types.SimpleNamespace(transforms=[resize_cls(size=(res, res))])
        DS = type('DS', (types.SimpleNamespace,), {'__len__': lambda s: s.n})
        self.dataset = DS(transforms=types.SimpleNamespace(transform=tf), n=n_images)
        self.batches = []

    def __iter__(self):
        n = len(self.dataset)
        i = 0
        while i + self.batch_sampler.batch_size <= n:
            bs = self.batch_sampler.batch_size
            res = self.dataset.transforms.transform.transforms[0].kwargs['size'][0]
            xb = torch.rand(bs, 3, res, res, generator=self.gen) * 2 - 1
            self.batches.append((bs, res))
            yield xb, torch.zeros(bs, dtype=torch.int64)
            i += bs


def _ref_progan_setup(num_main_iters=40):
    """Namespace + pickled config files the reference's ProGANLearner needs (temp HOME)."""
    import argparse as ap
    from pathlib import Path
    from PIL import Image
    tmp = tempfile.mkdtemp(prefix='ganlab_golden_')
    os.environ['HOME'] = tmp
    ns.pb.FMAP_BASE, ns.pb.FMAP_MAX = FMAP_BASE, FMAP_MAX
    bs = 4
    cfg = ap.Namespace(
        model='ProGAN', dev=torch.device('cpu'), n_gpu=1, enable_cudnn_autotuner=False, random_seed=0,
        gen_bs_mult=1, num_gen_iters=1, num_disc_iters=1, loss='wgan', gradient_penalty='wgan-gp',
        lda=10., gamma=1., eps_drift=.001, optimizer='adam', lr_base=.001, beta1=0., beta2=.99,
        eps=1e-8, wd=0., lr_sched='resolution dependent', lr_sched_custom=None,
        lr_fctr_dict={4: 1, 8: 1.25, 16: 1.5, 32: 1, 64: 1, 128: 1.5, 256: 2, 512: 3, 1024: 3},
        batch_size=bs, bs_dict={4: bs, 8: bs, 16: bs // 2, 32: bs, 64: bs, 128: bs, 256: bs, 512: bs // 2,
                                1024: bs // 4},
        nimg_transition=22, num_main_iters=num_main_iters, res_samples=16, res_dataset=16, init_res=4,
        model_upsample_type='nearest', model_downsample_type='average', align_corners=False,
        blur_type='binomial', bit_exact_resampling=False, nonlinearity='leaky relu', leakiness=.2,
        use_equalized_lr=True, normalize_z=True, len_latent=LEN_LATENT, latent_distribution='normal',
        use_pixelnorm=True, mbstd_group_size=4, use_ewma_gen=True, num_classes=0, class_condition=False,
        use_auxiliary_classifier=False, ac_disc_scale=1., ac_gen_scale=.1, num_iters_valid=1000,
        metrics_dev=torch.device('cpu'), gen_metrics=[], disc_metrics=[], img_grid_sz=4,
        img_grid_show_labels=True, save_samples_dir=Path(tmp) / 'samples',
        save_model_dir=Path(tmp) / 'models', num_iters_save_model=10 ** 9, num_workers=0,
        pin_memory=False)
    dcfg = ap.Namespace(dataset='custom', dataset_dir=tmp, ds_mean=[.5, .5, .5], ds_std=[.5, .5, .5],
                        dataset_downsample_type=Image.BOX, include_valid_set=False)
    with open(os.path.join(tmp, '.configs_dir.txt'), 'wb') as f:
        f.write(tmp.encode())
    with open(os.path.join(tmp, '.config.p'), 'wb') as f:
        pickle.dump(cfg, f)
    with open(os.path.join(tmp, '.data_config.p'), 'wb') as f:
        pickle.dump(dcfg, f)
    return cfg, bs


def golden_checkpoint():
    """A checkpoint file written by the REAL reference ``ProGANLearner.save_model`` (progan/learner.py:1238-1298)
    after 9 main iterations of its own ``train`` (4x4 stabilised -> 8x8 mid fade-in), plus what a loader must
    reproduce from it: generator / EWMA-generator / critic outputs on fixed inputs, and the critic parameters
    after ONE more Adam step taken with the restored optimiser state."""
    cfg, bs = _ref_progan_setup(num_main_iters=9)
    torch.manual_seed(0)
    np.random.seed(0)
    learner = ns.pl.ProGANLearner(cfg)
    import torchvision.transforms as tvt
    gen = torch.Generator().manual_seed(7)
    dl = _FakeDL(64, bs, 4, tvt.Resize, gen)
    learner.train(dl, num_main_iters=cfg.num_main_iters)
    learner.valid_z = torch.zeros(16, LEN_LATENT)          # save_model() dereferences it (:1281)
    path = os.path.join(HERE, 'ref_progan_ckpt.tar')
    learner.save_model(path)
    print(f'wrote ref_progan_ckpt.tar: {os.path.getsize(path) / 1024:.1f} KiB')
    g, d = learner.gen_model, learner.disc_model
    res = g.curr_res
    z = torch.randn(4, LEN_LATENT, generator=gen)
    real = torch.rand(4, 3, res, res, generator=gen) * 2 - 1
    out = dict(z=T(z), real=T(real), curr_res=np.int64(res), alpha=np.float64(g.alpha),
               fade_in=np.bool_(g.fade_in_phase), curr_img_num=np.int64(learner.curr_img_num),
               curr_phase_num=np.int64(learner.curr_phase_num), batch_size=np.int64(learner.batch_size),
               lr_gen=np.float64(learner.opt_gen.param_groups[0]['lr']),
               nimg_transition_lst=np.array([x if np.isfinite(x) else -1 for x in learner.nimg_transition_lst],
                                            dtype=np.float64))
    g.eval()
    with torch.no_grad():
        out['img'] = T(g(z))
        learner._update_gen_lagged()
        learner.gen_model_lagged.eval()
        out['img_lagged'] = T(learner.gen_model_lagged(z))
    g.train()
    with torch.no_grad():
        fake = g(z)
    out['fake_train'] = T(fake)
    for p in d.parameters():
        p.requires_grad_(True)
    d.zero_grad()
    if g.fade_in_phase:   # the D-step fades the reals in like the generator (progan/learner.py:771-779)
        with torch.no_grad():
            real = F.interpolate(F.avg_pool2d(real, kernel_size=2, stride=2), scale_factor=2, mode='nearest') * \
                (1. - g.alpha) + real * g.alpha
    d_fake, d_real = d(fake), d(real)
    out.update(d_fake=T(d_fake), d_real=T(d_real))
    torch.manual_seed(4321)
    loss = (d_fake - d_real).mean() + ref_calc_gp(d, 'wgan-gp', fake, real) + (d_real ** 2).mean() * 0.001
    torch.manual_seed(4321)
    out['eps_interp'] = T(torch.rand(4, 1, 1, 1))
    out['loss_d'] = T(loss)
    before = {k: v.detach().clone() for k, v in d.named_parameters()}
    loss.backward()
    learner.opt_disc.step()
    out.update({'dd.' + k: T(v.detach() - before[k]) for k, v in d.named_parameters()})
    out.update({'gd.' + k: T(v.grad) for k, v in d.named_parameters() if v.grad is not None})
    save('ref_progan_ckpt_expect.npz', **out)


def _ref_stylegan_setup(num_main_iters=9):
    """The ProGAN Namespace of ``_ref_progan_setup`` turned into a StyleGAN one (config.py:291-325 fields)."""
    cfg, bs = _ref_progan_setup(num_main_iters=num_main_iters)
    ns.sb.FMAP_BASE, ns.sb.FMAP_MAX = FMAP_BASE, FMAP_MAX
    cfg.model = 'StyleGAN'
    for k, v in dict(init_res=4, len_dlatent=LEN_LATENT, mapping_num_fcs=NUM_FCS, mapping_lrmul=.01, use_noise=True,
                     use_pixelnorm=False, use_instancenorm=True, pct_mixing_reg=.9, beta_trunc_trick=.9,
                     psi_trunc_trick=.7, cutoff_trunc_trick=1, loss='nonsaturating', gradient_penalty='r1').items():
        setattr(cfg, k, v)
    with open(os.path.join(os.environ['HOME'], '.config.p'), 'wb') as f:
        pickle.dump(cfg, f)
    return cfg, bs


def golden_checkpoint_stylegan():
    """A checkpoint written by the REAL reference ``StyleGANLearner.save_model`` (stylegan/learner.py:432-501) after 9
    main iterations of its own ``train`` (4x4 stabilised -> 8x8 mid fade-in, truncation trick on: cutoff stage 1,
    psi 0.7, w-average beta 0.9), and what the reference's OWN ``load_model`` (:503-640) makes of it: a second
    reference learner loads the file and its eval-mode generator / EWMA generator (truncation applied, explicit
    noise) are evaluated - including the ``w_ewma`` each of them ends up with."""
    cfg, bs = _ref_stylegan_setup(num_main_iters=9)
    torch.manual_seed(0)
    np.random.seed(0)
    learner = ns.sl.StyleGANLearner(cfg)
    import torchvision.transforms as tvt
    gen = torch.Generator().manual_seed(7)
    dl = _FakeDL(64, bs, 4, tvt.Resize, gen)
    learner.train(dl, num_main_iters=cfg.num_main_iters)
    learner.valid_z = torch.zeros(16, LEN_LATENT)
    path = os.path.join(HERE, 'ref_stylegan_ckpt.tar')
    learner.save_model(path)
    print(f'wrote ref_stylegan_ckpt.tar: {os.path.getsize(path) / 1024:.1f} KiB')
    # the reference's own reader.  It calls torch.load(path, map_location=...) as torch 1.x allowed; torch >= 2.6
    # defaults to weights_only=True, which refuses the pickled config / module objects the file holds.
    L2 = ns.sl.StyleGANLearner(cfg)
    import functools
    _orig_load = torch.load
    torch.load = functools.partial(_orig_load, weights_only=False)
    try:
        L2.load_model(path)
    finally:
        torch.load = _orig_load
    g, gl = L2.gen_model, L2.gen_model_lagged
    res = g.curr_res
    z = torch.randn(4, LEN_LATENT, generator=gen)
    noise = [torch.randn(4, 1, 4 * 2 ** (n // 2), 4 * 2 ** (n // 2), generator=gen) for n in range(len(g.gen_layers))]
    out = dict(z=T(z), curr_res=np.int64(res), alpha=np.float64(g.alpha), fade_in=np.bool_(g.fade_in_phase),
               curr_img_num=np.int64(L2.curr_img_num), curr_phase_num=np.int64(L2.curr_phase_num),
               batch_size=np.int64(L2.batch_size), use_truncation_trick=np.bool_(g.use_truncation_trick),
               trunc_cutoff_stage=np.int64(g.trunc_cutoff_stage), w_eval_psi=np.float64(g.w_eval_psi),
               w_ewma_beta=np.float64(g.w_ewma_beta), w_ewma=T(g.w_ewma), w_ewma_lagged_model=T(gl.w_ewma),
               w_ewma_saved=T(learner.gen_model.w_ewma), pct_mixing_reg=np.float64(g.pct_mixing_reg),
               ds_mean=T(L2.ds_mean), ds_std=T(L2.ds_std), valid_z=T(L2.valid_z))
    out.update({f'noise{i}': T(n) for i, n in enumerate(noise)})
    g.eval()
    gl.eval()
    with torch.no_grad():
        out['img'] = T(g(z, noise=noise))
        out['img_lagged'] = T(gl(z, noise=noise))
        g.use_truncation_trick = False
        out['img_no_trunc'] = T(g(z, noise=noise))
        g.use_truncation_trick = True
    save('ref_stylegan_ckpt_expect.npz', **out)


def golden_schedule():
    """Host-logic trace of the REAL reference ProGANLearner.train over a 4 -> 8 -> 16 schedule
    (BASELINE config #1 shape: 64 random images, batch 4): per main iteration the resolution, phase,
    alpha, batch size and LR seen by the G-step, plus phase bookkeeping at the end."""
    cfg, bs = _ref_progan_setup(num_main_iters=40)
    torch.manual_seed(0)
    np.random.seed(0)
    learner = ns.pl.ProGANLearner(cfg)
    import torchvision.transforms as tvt
    gen = torch.Generator().manual_seed(7)
    dl = _FakeDL(64, bs, 4, tvt.Resize, gen)
    rows = []
    calls = {'n': 0}

    def hook(mod, inp, outp):
        calls['n'] += 1
        if calls['n'] % 2 == 0:      # 2nd G forward of a main iteration == the G-step
            with __import__('warnings').catch_warnings():
                __import__('warnings').simplefilter('ignore')
                lr_g = learner.opt_gen.param_groups[0]['lr']
                lr_d = learner.opt_disc.param_groups[0]['lr']
            rows.append([calls['n'] // 2 - 1, mod.curr_res, int(mod.fade_in_phase), float(mod.alpha),
                         inp[0].shape[0], lr_g, lr_d, learner.curr_phase_num, learner.curr_img_num,
                         len(list(learner.opt_gen.param_groups[0]['params'])),
                         len(list(learner.opt_disc.param_groups[0]['params'])),
                         outp.shape[-1]])

    learner.gen_model.register_forward_hook(hook)
    learner.train(dl, num_main_iters=cfg.num_main_iters)
    trace = np.array(rows, dtype=np.float64)
    save('schedule_progan_4to16.npz',
         trace=trace,
         columns=np.array(['itr', 'curr_res', 'fade_in', 'alpha', 'batch', 'lr_gen', 'lr_disc',
                           'phase_num', 'curr_img_num_after_dstep', 'n_opt_params_g', 'n_opt_params_d',
                           'img_res'], dtype='U32'),
         nimg_transition_lst=np.array([x if np.isfinite(x) else -1 for x in learner.nimg_transition_lst],
                                      dtype=np.float64),
         real_batches=np.array(dl.batches, dtype=np.int64),
         cfg_nimg_transition=np.int64(cfg.nimg_transition), cfg_batch=np.int64(bs),
         bs_dict=np.array(sorted(cfg.bs_dict.items()), dtype=np.int64),
         lr_fctr=np.array(sorted(cfg.lr_fctr_dict.items()), dtype=np.float64),
         lagged_keys=np.array(list(learner.lagged_params.keys()), dtype='U64'),
         g_keys=np.array([k for k, _ in learner.gen_model.named_parameters()], dtype='U64'),
         d_keys=np.array([k for k, _ in learner.disc_model.named_parameters()], dtype='U64'),
         final_beta=np.float64(learner.beta))


# ---------------------------------------------------------------------------------------------- #
# ResNet GAN (BASELINE config #5): non-progressive 32 / 64 pixel nets, BatchNorm G, LayerNorm D, WGAN-GP
# ---------------------------------------------------------------------------------------------- #
RESNET_LATENT = 16


def make_resnets(res, fmap_g, fmap_d, nl=None):
    """``nl``: None = the constructors' default nn.ReLU(); 'tanh' = nn.Tanh(), what resnetgan/learner.py:180-181 passes
    for --nonlinearity tanh."""
    import resnetgan.architectures as ra
    kw = {} if nl is None else {'nl': {'tanh': torch.nn.Tanh}[nl]()}
    if res == 64:
        g = ra.Generator64PixResnet(len_latent=RESNET_LATENT, fmap=fmap_g, **kw)
        d = ra.Discriminator64PixResnet(fmap=fmap_d, **kw)
    else:
        g = ra.Generator32PixResnet(len_latent=RESNET_LATENT, fmap=fmap_g, **kw)
        d = ra.Discriminator32PixResnet(fmap=fmap_d, **kw)
    return g, d


def randomize_resnet(module, gen):
    """Zero biases -> random; unit BatchNorm/LayerNorm gains -> random around 1."""
    with torch.no_grad():
        for k, p in module.named_parameters():
            if k.endswith('bias'):
                p.copy_(torch.randn(p.shape, generator=gen) * 0.3)
            elif '.norm.weight' in k:
                p.copy_(1. + torch.randn(p.shape, generator=gen) * 0.2)


def golden_resnet(res, tag, fmap_g, fmap_d, b=4, lr=1e-3, n_iters=2, n_disc=2, nl=None):
    """Forward / gradient vectors plus `n_iters` main iterations of GANLearner.train's loop body
    (resnetgan/learner.py:538-684: one G iteration with D frozen, then `n_disc` D iterations with
    WGAN + WGAN-GP), torch.optim.Adam(betas=(0,.9)), every random draw explicit."""
    torch.manual_seed(21 + res)
    gen = torch.Generator().manual_seed(2100 + res)
    g, d = make_resnets(res, fmap_g, fmap_d, nl)
    randomize_resnet(g, gen)
    randomize_resnet(d, gen)
    g.train()
    d.train()
    out = {}
    out.update(sd_arrays('g0.', g))
    out.update(sd_arrays('d0.', d))
    # ---- net-level vectors (buffers restored afterwards so the step part starts from g0/d0) ----
    sd_g0 = {k: v.clone() for k, v in g.state_dict().items()}
    z = torch.randn(b, RESNET_LATENT, generator=gen)
    real = torch.rand(b, 3, res, res, generator=gen) * 2 - 1
    img = g(z)
    out.update(z=T(z), real=T(real), img=T(img))
    for p in d.parameters():
        p.requires_grad_(False)
    dout = d(img)
    lg = -dout.mean()
    g.zero_grad()
    lg.backward()
    out.update(d_of_img=T(dout), loss_g=T(lg))
    out.update({'gg.' + k: T(p.grad) for k, p in g.named_parameters()})
    out.update({'g_after_fwd.' + k: T(v) for k, v in g.state_dict().items() if 'running' in k})
    for p in d.parameters():
        p.requires_grad_(True)
    fake = img.detach()
    d.zero_grad()
    d_fake, d_real = d(fake), d(real)
    ld = (d_fake - d_real).mean()
    torch.manual_seed(777)
    gpv = ref_calc_gp(d, 'wgan-gp', fake, real)
    torch.manual_seed(777)
    out['eps_interp'] = T(torch.rand(b, 1, 1, 1))
    out.update(d_fake=T(d_fake), d_real=T(d_real), loss_d_adv=T(ld), gp=T(gpv), loss_d=T(ld + gpv))
    (ld + gpv).backward()
    out.update({'gd.' + k: T(p.grad) for k, p in d.named_parameters()})
    d.zero_grad()
    torch.manual_seed(777)
    ref_calc_gp(d, 'wgan-gp', fake, real).backward()
    out.update({'ggp.' + k: T(p.grad) for k, p in d.named_parameters() if p.grad is not None})
    g.load_state_dict(sd_g0)
    g.zero_grad()
    d.zero_grad()
    # ---- training iterations ----
    opt_g = torch.optim.Adam(g.parameters(), lr=lr, betas=(0., .9), eps=1e-8)
    opt_d = torch.optim.Adam(d.parameters(), lr=lr, betas=(0., .9), eps=1e-8)
    for it in range(n_iters):
        for p in d.parameters():
            p.requires_grad_(False)
        g.zero_grad()
        zg = torch.randn(b, RESNET_LATENT, generator=gen)
        out[f'i{it}.zg'] = T(zg)
        lg = -d(g(zg)).mean()
        lg.backward()
        opt_g.step()
        out[f'i{it}.loss_g'] = T(lg)
        for p in d.parameters():
            p.requires_grad_(True)
        for di in range(n_disc):
            d.zero_grad()
            zd = torch.randn(b, RESNET_LATENT, generator=gen)
            real = torch.rand(b, 3, res, res, generator=gen) * 2 - 1
            fake = g(zd).detach()
            ld = (d(fake) - d(real)).mean()
            torch.manual_seed(900 + 10 * it + di)
            ld = ld + ref_calc_gp(d, 'wgan-gp', fake, real)
            torch.manual_seed(900 + 10 * it + di)
            eps_i = torch.rand(b, 1, 1, 1)
            ld.backward()
            opt_d.step()
            out.update({f'i{it}.d{di}.zd': T(zd), f'i{it}.d{di}.real': T(real), f'i{it}.d{di}.eps_interp': T(eps_i),
                        f'i{it}.d{di}.loss_d': T(ld)})
    out.update(sd_arrays('g1.', g))
    out.update(sd_arrays('d1.', d))
    out.update(res=np.int64(res), lr=np.float64(lr), n_iters=np.int64(n_iters), n_disc=np.int64(n_disc),
               fmap_g=np.int64(fmap_g), fmap_d=np.int64(fmap_d), len_latent=np.int64(RESNET_LATENT))
    save(f'{tag}.npz', **out)


def golden_data():
    """The host image chain of data_config.py:307-341 evaluated with the real PIL + torch arithmetic
    (torchvision is absent here; its Resize / ToTensor / Normalize are thin wrappers over exactly these calls):
    Image.resize(BOX) -> uint8 HWC -> float/255 -> (x - mean)/std."""
    from PIL import Image
    rng = np.random.default_rng(7)
    imgs = rng.integers(0, 256, (6, 64, 64, 3), dtype=np.uint8)
    imgs[0] = 255
    imgs[1] = 0
    imgs[2, ::2] = 255          # worst case for the double rounding
    imgs[2, 1::2] = 0
    mean, std = [0.5, 0.5, 0.5], [0.5, 0.5, 0.5]
    mean2, std2 = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    out = dict(images=imgs, mean=np.float32(mean), std=np.float32(std), mean2=np.float32(mean2), std2=np.float32(std2))
    for res in (64, 32, 16, 8, 4):
        u8 = np.stack([np.asarray(Image.fromarray(im).resize((res, res), Image.BOX)) for im in imgs])
        out[f'u8_{res}'] = u8
        t_ = torch.from_numpy(u8).permute(0, 3, 1, 2).to(torch.float32).div(255)
        out[f'x_{res}'] = T((t_ - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1))
        out[f'x2_{res}'] = T((t_ - torch.tensor(mean2).view(1, 3, 1, 1)) / torch.tensor(std2).view(1, 3, 1, 1))
    import PIL
    out['pil_version'] = np.array(PIL.__version__)
    save('data_box.npz', **out)


if __name__ == '__main__':
    p = argparse.ArgumentParser()
    p.add_argument('--only', default=None)
    a = p.parse_args()
    jobs = {
        'ops': golden_ops,
        'sg_stab16': lambda: golden_nets('stylegan', 16, False, 1.0, 'stylegan_stab16'),
        'sg_fade16': lambda: golden_nets('stylegan', 16, True, 0.3, 'stylegan_fade16', seed=1),
        'sg_stab32_b8': lambda: golden_nets('stylegan', 32, False, 1.0, 'stylegan_stab32', b=8, seed=2),
        'sg_stab4': lambda: golden_nets('stylegan', 4, False, 1.0, 'stylegan_stab4', seed=3),
        'pg_stab16': lambda: golden_nets('progan', 16, False, 1.0, 'progan_stab16', loss='wgan',
                                         gp='wgan-gp', seed=4),
        'pg_fade8': lambda: golden_nets('progan', 8, True, 0.6, 'progan_fade8', loss='wgan', gp='wgan-gp',
                                        seed=5),
        'sg_r2_8': lambda: golden_nets('stylegan', 8, False, 1.0, 'stylegan_r2_8', gp='r2', seed=6),
        # the other resamplers (custom_layers.py:59-75, resnetgan/learner.py:147-173): (upsample, downsample, align_corners)
        'sg_bilinear16': lambda: golden_nets('stylegan', 16, False, 1.0, 'stylegan_bilinear16', seed=7,
                                             resample=('bilinear', 'bilinear', True)),
        'pg_nearest16': lambda: golden_nets('progan', 16, True, 0.4, 'progan_nearest16', loss='wgan', gp='wgan-gp',
                                            seed=8, resample=('bilinear', 'nearest', False)),
        'sg_bilinear8': lambda: golden_nets('stylegan', 8, False, 1.0, 'stylegan_bilinear8', seed=9,
                                            resample=('bilinear', 'bilinear', False)),
        # --nonlinearity tanh in the progressive networks (config.py:208, stylegan/learner.py, progan/learner.py: nl=nn.Tanh())
        'sg_tanh8': lambda: golden_nets('stylegan', 8, False, 1.0, 'stylegan_tanh8', seed=10, nl='tanh'),
        'pg_tanh8': lambda: golden_nets('progan', 8, True, 0.7, 'progan_tanh8', loss='wgan', gp='wgan-gp', seed=11, nl='tanh'),
        'mixing': golden_mixing,
        'step_sg': lambda: golden_step('stylegan', 16, 'step_stylegan16', 'nonsaturating', 'r1'),
        'step_sg_fade': lambda: golden_step('stylegan', 8, 'step_stylegan8_fade', 'nonsaturating', 'r1',
                                            fade_in=True, alpha=0.25),
        'step_pg': lambda: golden_step('progan', 8, 'step_progan8', 'wgan', 'wgan-gp'),
        'schedule': golden_schedule,
        'data': golden_data,
        'checkpoint': golden_checkpoint,
        'checkpoint_sg': golden_checkpoint_stylegan,
        'resnet64': lambda: golden_resnet(64, 'resnet64', fmap_g=2, fmap_d=2),
        'resnet32': lambda: golden_resnet(32, 'resnet32', fmap_g=8, fmap_d=8),
        # --nonlinearity tanh as the hidden activation (config.py:208, resnetgan/learner.py:180-181)
        'resnet32_tanh': lambda: golden_resnet(32, 'resnet32_tanh', fmap_g=8, fmap_d=8, n_iters=1, nl='tanh'),
    }
    for name, fn in jobs.items():
        if a.only is None or a.only == name:
            fn()
