"""GPU parity of the bf16-compute convolutions (csrc/conv_bf16.hip; BASELINE config #2 "bf16 compute / fp32
master") through the C-ABI.

Two yardsticks, both written out here:
  * EXACT arithmetic check: the kernel rounds its operands to bf16 (RNE) and accumulates the exact products in
    fp32, so against the oracle's conv evaluated in float64 on the SAME bf16-rounded operands only the fp32
    summation order differs -> 2e-5 relative.
  * PRECISION check against the plain fp32 oracle (the reference's arithmetic): bf16 operands carry 8 significand
    bits, a K-term dot product of random data is off by ~2^-9/sqrt(1) per term averaged -> we assert 1e-2 relative
    to the tensor's max (measured ~3e-3), the tolerance of config #2.
"""
import zlib

import pytest
import torch
import torch.nn.functional as F

from util import assert_close

pytestmark = pytest.mark.gpu
TOL_EXACT = 2e-5
TOL_BF16 = 1e-2


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    from gan_lab_amd import ops as _ops, _lib
    _lib.lib()
    return _ops


def bf(x):
    return x.to(torch.bfloat16).to(torch.float64)


CASES = [
    # N, Cin, H, W, Cout, up, bias, act
    (2, 64, 8, 32, 64, False, False, None),       # one tile, one co block, two K chunks
    (2, 128, 16, 64, 64, False, True, 'lrelu'),   # several tiles, 4 K chunks, fused epilogue
    (1, 64, 32, 32, 192, False, True, None),      # 3 co blocks
    (2, 64, 16, 16, 128, True, False, None),      # nearest upsample in front (materialised, then bf16 conv)
    (3, 192, 24, 96, 128, False, True, 'lrelu'),  # nothing a power of two except the tile
    (8, 320, 16, 96, 320, False, False, None),    # 25 channel tiles: every weight-gradient pipeline walks 2 segments
    (3, 128, 16, 16, 64, False, True, 'lrelu'),   # 16-wide maps: 16 x 16 pixel tiles; the weight gradient stays fp32
    (2, 64, 32, 48, 128, False, False, None),     # width a multiple of 16 only
    (3, 128, 12, 48, 64, True, True, 'lrelu'),    # upsample folded in: taps at 24 x 96 (3 strips, one 24-row segment)
    (2, 256, 16, 16, 128, False, True, 'lrelu'),  # 4 output tiles, 8 K chunks: split-K (2 splits) forward and input gradient
    (1, 512, 16, 16, 64, False, False, None),     # 1 tile, 16 chunks: 4 splits
    (2, 384, 8, 8, 128, True, True, 'lrelu'),     # split-K behind the folded upsample (taps at 16 x 16), 12 chunks: 3 splits
]


@pytest.mark.parametrize('case', CASES, ids=[str(c) for c in CASES])
def test_bf16_conv_fwd_dgrad_wgrad(ops, case):
    n, cin, h, w, cout, up, has_b, act = case
    gen = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()))
    x = torch.randn(n, cin, h, w, generator=gen)
    wt = torch.randn(cout, cin, 3, 3, generator=gen)
    b = torch.randn(cout, generator=gen) if has_b else None
    scale = 1.0 / (cin * 9) ** 0.5
    xg = x.clone().cuda().requires_grad_(True)
    wg = wt.clone().cuda().requires_grad_(True)
    bg = b.clone().cuda().requires_grad_(True) if has_b else None
    with ops.compute_dtype('bf16'):
        y = ops.conv2d(xg, wg, bg, scale=scale, padding=1, up=up, act=act, slope=0.2)
    assert ops.get_compute_dtype() == 'f32'
    gy = torch.randn(y.shape, generator=gen)
    y.backward(gy.cuda())          # backward OUTSIDE the block: must still run the bf16 kernels of its forward

    def ref(xr, wr, dt, round_ops):
        xi = F.interpolate(xr, scale_factor=2, mode='nearest') if up else xr
        r = (lambda v: bf(v)) if round_ops else (lambda v: v.to(dt))
        pre = F.conv2d(r(xi), r(wr * scale), None, padding=1)
        if b is not None:
            pre = pre + b.to(pre.dtype).view(1, -1, 1, 1)
        return F.leaky_relu(pre, 0.2) if act == 'lrelu' else pre, pre, xi

    # forward, exact
    y64, pre64, xi = ref(x, wt, torch.float64, True)
    assert_close(y.detach().cpu(), y64, TOL_EXACT, 'bf16 fwd vs bf16-operand float64')
    # forward, precision vs the fp32 oracle arithmetic
    y32, _, _ = ref(x, wt, torch.float32, False)
    assert_close(y.detach().cpu(), y32, TOL_BF16, 'bf16 fwd vs fp32')

    # gradients: the backward kernels round THEIR operands (gz, w*scale, x) to bf16
    # the LeakyReLU backward is the fp32 pointwise kernel: gz = gy * (y > 0 ? 1 : slope) from the kernel's own output
    mask = torch.where(y.detach().cpu() > 0, 1.0, 0.2).float() if act == 'lrelu' else torch.ones_like(gy)
    gz32 = gy * mask
    gz = gz32.double()
    gxi = F.conv_transpose2d(bf(gz32), bf(wt * scale), None, padding=1)
    gx_ref = F.avg_pool2d(gxi, 2) * 4 if up else gxi
    assert_close(xg.grad.cpu(), gx_ref, 5e-5, 'bf16 dgrad vs bf16-operand float64')
    if xi.shape[-1] % 32 == 0:
        gw_ref = torch.nn.grad.conv2d_weight(bf(xi), wt.shape, bf(gz32), padding=1) * scale
        assert_close(wg.grad.cpu(), gw_ref, 5e-5, 'bf16 wgrad vs bf16-operand float64')
    else:       # the bf16 weight-gradient kernels walk 32-pixel strips: narrower maps take the exact fp32 kernel
        gw_ref = torch.nn.grad.conv2d_weight(xi.double(), wt.shape, gz, padding=1) * scale
        assert_close(wg.grad.cpu(), gw_ref, 2e-5, 'fp32 wgrad of a 16-wide bf16 layer vs float64')
    if has_b:
        assert_close(bg.grad.cpu(), gz.sum(dim=(0, 2, 3)), 2e-4, 'bias grad')
    # precision vs the fp32 arithmetic on unrounded operands (same LeakyReLU mask: a bf16-sized forward error flips the
    # sign of ~1% of the near-zero pre-activations, which is a property of the activation, not of the conv kernels)
    gxi32 = F.conv_transpose2d(gz32, wt * scale, None, padding=1)
    assert_close(xg.grad.cpu(), F.avg_pool2d(gxi32, 2) * 4 if up else gxi32, TOL_BF16, 'bf16 dgrad vs fp32')
    gw32 = torch.nn.grad.conv2d_weight(xi, wt.shape, gz32, padding=1) * scale
    assert_close(wg.grad.cpu(), gw32, TOL_BF16, 'bf16 wgrad vs fp32')


def test_bf16_mode_leaves_unsupported_shapes_exact(ops):
    """Shapes outside ganlab_conv_bf16_supported (thin / ragged / 1x1) keep the exact fp32 kernels in bf16 mode."""
    gen = torch.Generator().manual_seed(7)
    x = torch.randn(2, 16, 16, 16, generator=gen)
    wt = torch.randn(16, 16, 3, 3, generator=gen)
    with ops.compute_dtype('bf16'):
        y = ops.conv2d(x.cuda(), wt.cuda(), None, scale=0.1, padding=1)
    assert_close(y.cpu(), F.conv2d(x.double() * 0.1, wt.double(), padding=1), 2e-5, 'fp32 kernel in bf16 mode')


POOL_CASES = [(2, 64, 16, 64, 128, True, 'lrelu'), (3, 128, 32, 32, 64, False, None), (2, 64, 32, 16, 64, True, 'lrelu'),
              (3, 64, 24, 96, 128, True, None), (2, 256, 16, 16, 128, True, 'lrelu')]     # last: split-K, pooled in the finish


@pytest.mark.parametrize('case', POOL_CASES, ids=[str(c) for c in POOL_CASES])
def test_bf16_pooled_conv_fwd_dgrad_wgrad(ops, case):
    """conv -> AvgPool2d(2) -> +bias -> LeakyReLU as ONE bf16 kernel (the 2 x 2 sum in the epilogue), its input gradient
    (the upsample / 4 folded into the staging) and its weight gradient (on the materialised up2(gy) / 4): exact against
    float64 on bf16-rounded operands - the pooled gradient up2(gz)/4 is itself rounded to bf16 by the kernels, as the
    reference composition would round it."""
    n, cin, h, w, cout, has_b, act = case
    gen = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()))
    x = torch.randn(n, cin, h, w, generator=gen)
    wt = torch.randn(cout, cin, 3, 3, generator=gen)
    b = torch.randn(cout, generator=gen) if has_b else None
    scale = 1.0 / (cin * 9) ** 0.5
    xg, wg = x.clone().cuda().requires_grad_(True), wt.clone().cuda().requires_grad_(True)
    bg = b.clone().cuda().requires_grad_(True) if has_b else None
    with ops.compute_dtype('bf16'):
        assert ops.pool_fusable(n, cin, h, w, cout, 3, 1)
        y = ops.conv2d(xg, wg, bg, scale=scale, padding=1, act=act, slope=0.2, pool=True)
    assert tuple(y.shape) == (n, cout, h // 2, w // 2)
    gy = torch.randn(y.shape, generator=gen)
    y.backward(gy.cuda())
    pre = F.avg_pool2d(F.conv2d(bf(x), bf(wt * scale), None, padding=1), 2)
    if has_b:
        pre = pre + b.double().view(1, -1, 1, 1)
    ref = F.leaky_relu(pre, 0.2) if act == 'lrelu' else pre
    assert_close(y.detach().cpu(), ref, TOL_EXACT, 'bf16 pooled conv vs bf16-operand float64')
    mask = torch.where(y.detach().cpu() > 0, 1.0, 0.2).float() if act == 'lrelu' else torch.ones_like(gy)
    gz = gy * mask                                                    # fp32, as the pointwise kernel computes it
    gzu = F.interpolate(gz, scale_factor=2, mode='nearest') * 0.25    # adjoint of the average pool (exact in fp32)
    gx_ref = F.conv_transpose2d(bf(gzu), bf(wt * scale), None, padding=1)
    assert_close(xg.grad.cpu(), gx_ref, 5e-5, 'bf16 pooled-conv dgrad vs bf16-operand float64')
    gw_ref = torch.nn.grad.conv2d_weight(bf(x), wt.shape, bf(gzu), padding=1) * scale
    if w % 32 == 0:
        assert_close(wg.grad.cpu(), gw_ref, 5e-5, 'bf16 pooled-conv wgrad vs bf16-operand float64')
    else:
        gw64 = torch.nn.grad.conv2d_weight(x.double(), wt.shape, gzu.double(), padding=1) * scale
        assert_close(wg.grad.cpu(), gw64, 2e-5, 'fp32 wgrad of a 16-wide pooled bf16 layer')
    if has_b:
        assert_close(bg.grad.cpu(), gz.double().sum(dim=(0, 2, 3)), 2e-4, 'bias grad')


FULL = [(8, 128, 128, 128, 128), (8, 256, 64, 64, 512), (8, 512, 32, 32, 512), (8, 512, 16, 16, 512)]


@pytest.mark.parametrize('shape', FULL, ids=[str(c) for c in FULL])
def test_bf16_full_size_triple_product(ops, shape):
    """BASELINE config #2's own layer sizes (batch 8), where a float64 reference is out of reach: with operands that ARE
    bf16 numbers every product inside the three kernels is exact, so <conv(x, w), gy> = <x, dgrad(gy, w)> =
    <w, wgrad(gy, x)> - one triple sum, accumulated in fp32 in three different orders by three different kernels
    (forward tile kernel, its input-gradient twin, the rolling-row weight gradient or - at 16 pixels - the fp32 one)."""
    n, cin, h, w, cout = shape
    gen = torch.Generator().manual_seed(n * cin + h)
    bfr = lambda t: t.to(torch.bfloat16).float().cuda()
    x, wt = bfr(torch.randn(n, cin, h, w, generator=gen)), bfr(torch.randn(cout, cin, 3, 3, generator=gen))
    gy = bfr(torch.randn(n, cout, h, w, generator=gen))
    with ops.compute_dtype('bf16'):
        g = ops.Geom(n, cin, h, w, cout, 3, 1)
    assert g.bf is not None
    y = ops.k_conv_fwd(x, wt, None, g, 1.0)
    gx = ops.k_conv_dgrad(gy, wt, g, 1.0)
    gw = ops.k_conv_wgrad(gy, x, g, 1.0)
    dot = lambda a, b: float((a.double() * b.double()).sum())
    t_y, t_x, t_w = dot(y, gy), dot(x, gx), dot(wt, gw)
    scale = (float(y.double().pow(2).sum()) * float(gy.double().pow(2).sum())) ** 0.5     # |y| |gy|: the sum's natural size
    assert abs(t_y - t_x) <= 1e-5 * scale and abs(t_y - t_w) <= 1e-5 * scale, (t_y, t_x, t_w, scale)
    # and the weight gradient is not merely consistent in the mean: a random projection onto a second bf16 weight
    w2 = bfr(torch.randn(cout, cin, 3, 3, generator=gen))
    y2 = ops.k_conv_fwd(x, w2, None, g, 1.0)
    assert abs(dot(y2, gy) - dot(w2, gw)) <= 1e-5 * scale


def test_bf16_double_backward_closed(ops):
    """R1-style double backward through bf16 layers: d/dw of |d y/d x|^2 exists and is close to fp32 autograd."""
    gen = torch.Generator().manual_seed(11)
    n, c, h, w = 2, 64, 8, 32
    x = torch.randn(n, c, h, w, generator=gen)
    w1 = torch.randn(64, c, 3, 3, generator=gen)
    w2 = torch.randn(64, 64, 3, 3, generator=gen)
    s = 1.0 / (c * 9) ** 0.5

    def run(conv, dev):
        xr = x.clone().to(dev).requires_grad_(True)
        a, b_ = w1.clone().to(dev).requires_grad_(True), w2.clone().to(dev).requires_grad_(True)
        out = conv(conv(xr, a), b_).sum(dim=(1, 2, 3))
        g, = torch.autograd.grad(out.sum(), xr, create_graph=True)
        pen = (g ** 2).sum() if dev == 'cpu' else ops.sumsq_all(g)
        pen.backward()
        return pen.detach().cpu(), a.grad.cpu(), b_.grad.cpu()

    with ops.compute_dtype('bf16'):
        pg, ag, bgr = run(lambda t_, w_: ops.conv2d(t_, w_, None, scale=s, padding=1, act='lrelu'), 'cuda')
    pc, ac, bc = run(lambda t_, w_: F.leaky_relu(F.conv2d(t_ * s, w_, padding=1), 0.2), 'cpu')
    assert_close(pg, pc, 2e-2, 'penalty value')
    assert_close(ag, ac, 3e-2, 'penalty grad w1')
    assert_close(bgr, bc, 3e-2, 'penalty grad w2')


def test_bf16_stylegan64_step_vs_fp32_oracle(ops, capsys):
    """BASELINE config #2 in miniature: StyleGAN at REAL channel widths (512 ... 256 at 64^2), batch 4, one D step
    (nonsaturating + R1 + drift) and one G step with the eligible 3x3 layers (32^2 and 64^2: 8 of G's and D's
    convolutions, ~85% of the FLOPs) on the bf16 kernels - against the CPU oracle in fp32 on identical weights,
    latents and noise.  bf16 operands (8 significand bits) through ~40 layers: the image / logits agree to ~1e-2 of
    their range and every parameter-gradient tensor points the same way (cosine >= 0.99); tolerances below are the
    measured values with ~3x margin."""
    from gan_lab_amd import progressive as P
    from gan_lab_amd.progan.architectures import StyleDiscriminator
    from gan_lab_amd.stylegan.architectures import StyleGenerator
    from gan_lab_amd.utils import backprop_utils as bp
    from oracle import nets, ops as O, step
    old = (P.FMAP_BASE, P.FMAP_MAX)
    P.FMAP_BASE, P.FMAP_MAX = 8192, 512
    try:
        torch.manual_seed(5)
        P.StyleGAN.reset_state()
        g = StyleGenerator(final_res=64, blur_type='binomial')
        d = StyleDiscriminator(final_res=64, blur_type='binomial')
        for _ in range(4):
            g.increase_scale()
            d.increase_scale()
    finally:
        P.FMAP_BASE, P.FMAP_MAX = old
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
    b = 4
    z, real = torch.randn(b, 512), torch.rand(b, 3, 64, 64) * 2 - 1
    noise = [torch.randn(b, 1, 4 * 2 ** (n // 2), 4 * 2 ** (n // 2)) for n in range(len(g.gen_layers))]
    with ops.compute_dtype('bf16'):
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
    cfg = nets.make_cfg()
    og = {k: v.clone().requires_grad_(True) for k, v in sd_g.items()}
    od = {k: v.clone().requires_grad_(True) for k, v in sd_d.items()}
    oimg = nets.stylegen_forward(og, z, noise, cfg)
    ototal, parts = step.d_loss(od, cfg, oimg.detach(), real, 'nonsaturating', 'r1', 10.0, 1.0, 0.001,
                                return_parts=True)
    ototal.backward()
    olg = O.loss_gen('nonsaturating', nets.disc_forward({k: v.detach() for k, v in od.items()}, oimg, cfg))
    olg.backward()

    from util import rel_err
    rep = {'img': rel_err(img, oimg), 'gp': rel_err(gp, parts['gp']), 'loss_d': rel_err(loss_d, ototal),
           'loss_g': rel_err(loss_g, olg)}

    def cos(a, ref):
        a, ref = a.detach().cpu().double().flatten(), ref.double().flatten()
        return (a @ ref / (a.norm() * ref.norm()).clamp_min(1e-300)).item()
    gmax_d = max(v.grad.abs().max().item() for v in od.values() if v.grad is not None)
    gmax_g = max(v.grad.abs().max().item() for v in og.values() if v.grad is not None)
    worst_cos, worst_key = 1.0, None
    for net, ref, gmax, tag in ((d, od, gmax_d, 'd.'), (g, og, gmax_g, 'g.')):
        for k, p in net.named_parameters():
            r = ref[k].grad
            if r is None or r.abs().max() < 1e-3 * gmax:     # numerically-zero gradients (bias before InstanceNorm)
                continue
            c = cos(p.grad, r)
            if c < worst_cos:
                worst_cos, worst_key = c, tag + k
    rep['worst_grad_cosine'] = (worst_cos, worst_key)
    with capsys.disabled():
        print('\nbf16 StyleGAN-64 step vs fp32 oracle:', rep)
    # measured on MI355X: img 6.0e-3, R1 3.4e-3, loss_d 7e-4, loss_g 1.4e-3, worst cosine 0.9936 (a noise weight)
    assert rep['img'] < 2e-2 and rep['loss_d'] < 1e-2 and rep['loss_g'] < 1e-2 and rep['gp'] < 2e-2, rep
    assert worst_cos > 0.98, rep


def test_bf16_learner_trains_and_uses_the_bf16_kernels(ops, monkeypatch):
    """config.compute_dtype='bf16' through the learner (BASELINE config #2's switch): a StyleGAN at 64 channels,
    32^2, trains two iterations with finite losses, the eligible layers really run on the bf16 entry points, and a
    later fp32 learner is back on the exact kernels."""
    from gan_lab_amd import _lib, progressive as P
    from gan_lab_amd.config import make_config
    from gan_lab_amd.stylegan.learner import StyleGANLearner
    from gan_lab_amd.utils.data_utils import SyntheticImageLoader
    calls = {'fwd': 0, 'dgrad': 0, 'wgrad': 0}
    L_ = _lib.lib()
    for name in ('fwd', 'dgrad', 'wgrad'):
        orig = getattr(L_, f'ganlab_conv_{name}_bf16')

        def counted(*a, _o=orig, _n=name):
            calls[_n] += 1
            return _o(*a)
        monkeypatch.setattr(L_, f'ganlab_conv_{name}_bf16', counted, raising=False)
    old = (P.FMAP_BASE, P.FMAP_MAX)
    P.FMAP_BASE, P.FMAP_MAX = 1024, 64
    try:
        def cfg(dt):
            return make_config('stylegan', dev='cuda', pin_memory=False, res_samples=32, res_dataset=32, init_res=32,
                               batch_size=8, len_latent=64, len_dlatent=64, mapping_num_fcs=2, loss='nonsaturating',
                               gradient_penalty='r1', cutoff_trunc_trick=None, num_iters_save_model=10 ** 9,
                               log_every=1, compute_dtype=dt)
        L = StyleGANLearner(cfg('bf16'))
        assert ops.get_compute_dtype() == 'bf16'
        L.train(SyntheticImageLoader(1024, 8, 32), num_main_iters=2)
        import math
        assert math.isfinite(L.last_losses['loss_d']) and math.isfinite(L.last_losses['loss_g'])
        assert calls['fwd'] > 10 and calls['dgrad'] > 5 and calls['wgrad'] > 5, calls
        n_bf16 = dict(calls)
        L2 = StyleGANLearner(cfg('f32'))
        assert ops.get_compute_dtype() == 'f32'
        L2.train(SyntheticImageLoader(1024, 8, 32), num_main_iters=1)
        assert calls == n_bf16
    finally:
        P.FMAP_BASE, P.FMAP_MAX = old
        ops.set_compute_dtype('f32')
