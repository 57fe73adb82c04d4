"""GPU parity of every HIP op (through the C-ABI) against the CPU oracle, on seeded inputs.
Tolerance: the north star allows 1e-3 relative fp32; we assert 2e-4 (fp32 re-association only)."""
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util import assert_close, load_golden, t

pytestmark = pytest.mark.gpu
TOL = 2e-4


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    from gan_lab_amd import ops as _ops, _lib
    _lib.lib()  # raises if the HIP library is missing - no fallback
    return _ops


def gpu(x):
    return x.detach().clone().cuda()


def rnd(gen, *shape):
    return torch.randn(*shape, generator=gen)


CONV_CASES = [
    # N, Cin, H, W, Cout, ks, pad, up, bias, act
    (3, 6, 7, 7, 5, 3, 1, False, True, None),          # odd sizes, channel padding
    (2, 16, 64, 64, 16, 3, 1, False, False, None),     # thin 16->16 (the 1024^2 layer shape, small image)
    (2, 32, 32, 32, 16, 3, 1, True, False, None),      # upsample folded in: 32@32^2 -> 16@64^2
    (2, 3, 64, 64, 16, 1, 0, False, True, 'lrelu'),    # fromRGB
    (2, 16, 64, 64, 3, 1, 0, False, True, None),       # toRGB
    (4, 64, 16, 16, 64, 3, 1, False, True, 'lrelu'),   # thick, 16x16 tile geometry
    (4, 128, 8, 8, 96, 3, 1, False, True, 'lrelu'),    # 8x8x4 geometry, 2 co tiles with a ragged one
    (8, 256, 4, 4, 256, 3, 1, False, True, 'lrelu'),   # 4x4x16 geometry
    (4, 257, 4, 4, 64, 3, 1, False, True, 'lrelu'),    # mbstd-style odd Cin (513 in the real net)
    (2, 32, 8, 8, 32, 3, 1, True, False, None),        # up at small res
    (1, 16, 128, 96, 16, 3, 1, False, False, None),    # non-square, 32x8 geometry with ragged tiles
    (5, 24, 20, 20, 40, 3, 1, False, True, None),      # nothing a power of two
    (8, 16, 16, 16, 16, 3, 1, False, True, 'lrelu'),   # thin layer, 16x16-tile variant of the vertical strip kernel
    (2, 12, 64, 16, 16, 3, 1, False, False, None),     # thin, channel padding (12 -> 16), tall narrow image
    (2, 16, 24, 64, 16, 3, 1, False, True, None),      # thin, three tile rows of 8: strips with a top / bottom edge each
    # split-K (few output tiles, long contraction): forward and input gradient share tiles between S workgroups
    (32, 512, 8, 8, 512, 3, 1, False, True, 'lrelu'),  # the benchmark's 8x8 layer: 256 workgroups -> S = 2
    (32, 513, 4, 4, 512, 3, 1, False, True, 'lrelu'),  # critic's last 3x3 (mbstd channel): S = 4, odd Cin
    (3, 200, 5, 7, 72, 3, 1, False, True, None),       # ragged everything: 13 K-chunks of 16 -> 25 of 8, S = 4
    (2, 72, 8, 8, 200, 3, 1, False, False, 'lrelu'),   # 9 K-chunks: S is cut back so no workgroup gets an empty range
    (8, 256, 16, 16, 128, 3, 1, False, True, 'lrelu'), # 16x16 tiles on the vector-staged kernel, S = 4
    (32, 512, 16, 16, 512, 3, 1, False, True, None),   # the benchmark's 16x16 layer, S = 2
    # rolling-window weight gradient (wgrad_roll.hip: thin layers, W % 64 == 0, H % 4 == 0)
    (3, 12, 40, 64, 10, 3, 1, False, True, None),      # channel padding on both sides, 10 steps in one strip
    (2, 32, 16, 64, 24, 3, 1, False, True, 'lrelu'),   # 2 x 2 channel-tile pairs, ragged co tile
    (1, 5, 72, 192, 32, 3, 1, False, False, None),     # three columns, 18 steps = two strips (16 + 2)
    (2, 16, 8, 128, 16, 3, 1, False, True, None),      # two steps only: prologue + one prefetch
    (2, 16, 68, 64, 16, 3, 1, False, False, None),     # 17 steps: a strip of one step
]


@pytest.mark.parametrize('case', CONV_CASES, ids=[str(c) for c in CONV_CASES])
def test_conv_fwd_dgrad_wgrad(ops, case):
    n, cin, h, w, cout, ks, pad, up, has_b, act = case
    gen = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()))   # deterministic across processes
    x = rnd(gen, n, cin, h, w).requires_grad_(True)
    wt = rnd(gen, cout, cin, ks, ks).requires_grad_(True)
    b = (rnd(gen, cout) if has_b else None)
    if b is not None:
        b.requires_grad_(True)
    scale = 1.0 / np.sqrt(cin * ks * ks)
    xi = F.interpolate(x, scale_factor=2, mode='nearest') if up else x
    z_ref = F.conv2d(xi * scale, wt, b, padding=pad)
    y_ref = F.leaky_relu(z_ref, 0.2) if act else z_ref
    cot = rnd(gen, *y_ref.shape)

    xg, wg = gpu(x).requires_grad_(True), gpu(wt).requires_grad_(True)
    bg = gpu(b).requires_grad_(True) if has_b else None
    y = ops.conv2d(xg, wg, bg, scale=scale, padding=pad, up=up, act=act)
    assert_close(y, y_ref, TOL, 'y')
    (y * cot.cuda()).sum().backward()
    # the reference backward takes the LeakyReLU mask of the kernel's own output: among a million outputs a few sit within
    # rounding of zero, and one flipped sign moves 9*Cin input gradients by ~1/sqrt(9*Cin) of their size
    mask = torch.where(y.detach().cpu() > 0, 1.0, 0.2) if act else torch.ones_like(cot)
    z_ref.backward(cot * mask)
    assert_close(xg.grad, x.grad, TOL, 'dgrad')
    assert_close(wg.grad, wt.grad, TOL, 'wgrad')
    if has_b:
        assert_close(bg.grad, b.grad, TOL, 'bias grad')


LIN_CASES = [(8, 512, 512, 0.01), (4, 16, 32, 1.0), (32, 512, 1, 1.0), (5, 24, 70, 1.0), (32, 512, 1024, 1.0),
             # long contraction, few rows: the critic's 4x4 valid conv as a linear (8192 -> 512), split over workgroups
             (32, 8192, 512, 1.0), (4, 8192, 512, 1.0), (64, 2048, 40, 1.0)]


@pytest.mark.parametrize('case', LIN_CASES, ids=[str(c) for c in LIN_CASES])
def test_linear(ops, case):
    n, cin, cout, lrmul = case
    gen = torch.Generator().manual_seed(cin * 7 + cout)
    x = rnd(gen, n, cin).requires_grad_(True)
    wt = (rnd(gen, cout, cin) / lrmul).requires_grad_(True)
    b = rnd(gen, cout).requires_grad_(True)
    ws = np.sqrt(2.0 / cin)
    y_ref = F.leaky_relu((F.linear(x * ws, wt, b)) * lrmul, 0.2)
    cot = rnd(gen, n, cout)
    (y_ref * cot).sum().backward()
    xg, wg, bg = gpu(x).requires_grad_(True), gpu(wt).requires_grad_(True), gpu(b).requires_grad_(True)
    y = ops.linear(xg, wg, bg, scale=ws * lrmul, bias_scale=lrmul, act='lrelu')
    assert_close(y, y_ref, TOL, 'y')
    (y * cot.cuda()).sum().backward()
    assert_close(xg.grad, x.grad, TOL, 'gx')
    assert_close(wg.grad, wt.grad, TOL, 'gw')
    assert_close(bg.grad, b.grad, TOL, 'gb')


def test_conv4x4_valid_as_linear(ops):
    gen = torch.Generator().manual_seed(44)
    x = rnd(gen, 6, 32, 4, 4).requires_grad_(True)
    wt = rnd(gen, 48, 32, 4, 4).requires_grad_(True)
    b = rnd(gen, 48).requires_grad_(True)
    y_ref = F.leaky_relu(F.conv2d(x * 0.05, wt, b), 0.2)
    cot = rnd(gen, *y_ref.shape)
    (y_ref * cot).sum().backward()
    xg, wg, bg = gpu(x).requires_grad_(True), gpu(wt).requires_grad_(True), gpu(b).requires_grad_(True)
    y = ops.conv2d(xg, wg, bg, scale=0.05, padding=0, act='lrelu')
    assert_close(y, y_ref, TOL)
    (y * cot.cuda()).sum().backward()
    assert_close(xg.grad, x.grad, TOL)
    assert_close(wg.grad, wt.grad, TOL)
    assert_close(bg.grad, b.grad, TOL)


def test_conv_golden_vectors(ops):
    """Directly against the reference's own Conv2dEx / LinearEx outputs (tests/golden/ops.npz)."""
    from oracle import ops as O
    G = load_golden('ops.npz')
    x, w, b = gpu(t(G['conv_x'])).requires_grad_(True), gpu(t(G['conv_w'])).requires_grad_(True), \
        gpu(t(G['conv_b'])).requires_grad_(True)
    y = ops.conv2d(x, w, b, scale=float(G['conv_wscale']), padding=1)
    assert_close(y, G['conv_y'], TOL)
    (y * t(G['conv_cot']).cuda()).sum().backward()
    assert_close(x.grad, G['conv_gx'], TOL)
    assert_close(w.grad, G['conv_gw'], TOL)
    assert_close(b.grad, G['conv_gb'], TOL)
    y4 = ops.conv2d(gpu(t(G['conv4_x'])), gpu(t(G['conv4_w'])), gpu(t(G['conv4_b'])), scale=float(G['conv4_wscale']))
    assert_close(y4, G['conv4_y'], TOL)
    y1 = ops.conv2d(gpu(t(G['conv1_x'])), gpu(t(G['conv1_w'])), gpu(t(G['conv1_b'])), scale=float(G['conv1_wscale']))
    assert_close(y1, G['conv1_y'], TOL)
    xl, wl, bl = gpu(t(G['lin_x'])).requires_grad_(True), gpu(t(G['lin_w'])).requires_grad_(True), \
        gpu(t(G['lin_b'])).requires_grad_(True)
    yl = ops.linear(xl, wl, bl, scale=float(G['lin_wscale']) * 0.01, bias_scale=0.01)
    assert_close(yl, G['lin_y'], TOL)
    (yl * t(G['lin_cot']).cuda()).sum().backward()
    assert_close(xl.grad, G['lin_gx'], TOL)
    assert_close(wl.grad, G['lin_gw'], TOL)
    assert_close(bl.grad, G['lin_gb'], TOL)


def test_conv_double_backward(ops):
    """R1-shaped second order: d/dw of || d(sum lrelu(conv(x,w)+b))/dx ||^2 through the HIP kernels."""
    gen = torch.Generator().manual_seed(9)
    for (n, cin, h, cout, up) in [(2, 8, 8, 12, False), (2, 16, 16, 16, False), (2, 8, 4, 8, True)]:
        x = rnd(gen, n, cin, h, h).requires_grad_(True)
        w1 = rnd(gen, cout, cin, 3, 3).requires_grad_(True)
        b1 = rnd(gen, cout).requires_grad_(True)
        w2 = rnd(gen, 4, cout, 3, 3).requires_grad_(True)

        def net_ref(x):
            xi = F.interpolate(x, scale_factor=2, mode='nearest') if up else x
            hmid = F.leaky_relu(F.conv2d(xi * 0.1, w1, b1, padding=1), 0.2)
            return F.conv2d(hmid * 0.2, w2, None, padding=1)
        out = net_ref(x)
        g, = torch.autograd.grad(out.sum(), x, create_graph=True)
        pen = (g ** 2).sum()
        pen.backward()

        xg = gpu(x).requires_grad_(True)
        w1g, b1g, w2g = gpu(w1).requires_grad_(True), gpu(b1).requires_grad_(True), gpu(w2).requires_grad_(True)
        hmid = ops.conv2d(xg, w1g, b1g, scale=0.1, padding=1, up=up, act='lrelu')
        outg = ops.conv2d(hmid, w2g, None, scale=0.2, padding=1)
        gg, = torch.autograd.grad(ops.sum_all(outg), xg, create_graph=True)
        assert_close(gg, g, TOL, 'first-order grad')
        peng = ops.sumsq_all(gg)
        assert_close(peng, pen, TOL, 'penalty')
        peng.backward()
        assert_close(w1g.grad, w1.grad, 5e-4, 'ggw1')
        assert_close(w2g.grad, w2.grad, 5e-4, 'ggw2')
        assert b1g.grad is None or b1g.grad.abs().max() < 1e-6  # LeakyReLU'' == 0


def test_blur_pool_up(ops):
    from oracle import ops as O
    G = load_golden('ops.npz')
    x = gpu(t(G['blur_x'])).requires_grad_(True)
    y = ops.blur(x)
    assert_close(y, G['blur_y'], TOL)
    (y * t(G['blur_cot']).cuda()).sum().backward()
    assert_close(x.grad, G['blur_gx'], TOL)
    gen = torch.Generator().manual_seed(3)
    for shape in [(2, 5, 8, 8), (1, 3, 64, 32), (3, 16, 4, 4)]:
        a = rnd(gen, *shape).requires_grad_(True)
        ag = gpu(a).requires_grad_(True)
        for f_ref, f_gpu in ((O.avgpool2, ops.avg_pool2), (O.upsample2, ops.upsample2), (O.blur_binomial, ops.blur)):
            a.grad = ag.grad = None
            r = f_ref(a)
            cot = rnd(gen, *r.shape)
            (r * cot).sum().backward()
            o = f_gpu(ag)
            assert_close(o, r, TOL)
            (o * cot.cuda()).sum().backward()
            assert_close(ag.grad, a.grad, TOL)


def test_bias_act_noise(ops):
    gen = torch.Generator().manual_seed(5)
    for shape in [(2, 6, 5, 5), (2, 16, 32, 32)]:
        n, c, h, w = shape
        x = rnd(gen, *shape).requires_grad_(True)
        b = rnd(gen, 1, c, 1, 1).requires_grad_(True)
        nw = rnd(gen, 1, c, 1, 1).requires_grad_(True)
        nz = rnd(gen, n, 1, h, w)
        ref = F.leaky_relu(x + nw * nz + b, 0.2)
        cot = rnd(gen, *shape)
        (ref * cot).sum().backward()
        xg, bg, nwg = gpu(x).requires_grad_(True), gpu(b).requires_grad_(True), gpu(nw).requires_grad_(True)
        y = ops.bias_act(xg, bg, nz.cuda(), nwg, act='lrelu')
        assert_close(y, ref, TOL)
        (y * cot.cuda()).sum().backward()
        assert_close(xg.grad, x.grad, TOL)
        assert_close(bg.grad, b.grad, TOL)
        assert_close(nwg.grad, nw.grad, TOL)


PN_SHAPES = [(2, 16, 32, 32), (2, 32, 16, 24), (3, 64, 16, 16), (2, 128, 8, 8),      # channels in registers
             (4, 512, 8, 8), (3, 256, 64, 64), (2, 512, 4, 4), (2, 320, 8, 8),       # channel-split blocks (S = 16 / 4)
             (5, 512, 1, 1), (2, 96, 8, 8), (1, 256, 512, 512)]                      # one thread per pixel


@pytest.mark.parametrize('shape', PN_SHAPES, ids=[str(c) for c in PN_SHAPES])
def test_pixelnorm_every_kernel_vs_float64(ops, shape):
    """PixelNorm2d (custom_layers.py:81-86) forward and backward on each of its three kernel families (chosen by the
    shape inside ganlab_pixelnorm_*_f32): against float64, 2e-6 relative - the sums have at most 512 terms."""
    gen = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=gen) * 1.5 + 0.2
    cot = torch.randn(*shape, generator=gen)
    xd = x.double().requires_grad_(True)
    ref = xd * ((xd ** 2).mean(dim=1, keepdim=True) + 1e-8).rsqrt()
    (ref * cot.double()).sum().backward()
    xg = x.cuda().requires_grad_(True)
    y = ops.pixelnorm(xg)
    assert_close(y, ref.detach(), 2e-6, 'pixelnorm forward')
    (y * cot.cuda()).sum().backward()
    assert_close(xg.grad, xd.grad, 2e-6, 'pixelnorm backward')


def test_instnorm_style_and_pixelnorm(ops):
    from oracle import ops as O
    G = load_golden('ops.npz')
    x = gpu(t(G['in_x'])).requires_grad_(True)
    y = ops.instnorm_style(x, None)
    assert_close(y, G['in_y'], TOL)
    (y * t(G['in_cot']).cuda()).sum().backward()
    assert_close(x.grad, G['in_gx'], TOL)
    x = gpu(t(G['pn_x'])).requires_grad_(True)
    y = ops.pixelnorm(x)
    assert_close(y, G['pn_y'], TOL)
    (y * t(G['pn_cot']).cuda()).sum().backward()
    assert_close(x.grad, G['pn_gx'], TOL)
    gen = torch.Generator().manual_seed(8)
    for shape in [(3, 8, 4, 4), (2, 16, 64, 64), (2, 5, 7, 9)]:
        n, c = shape[0], shape[1]
        a = (rnd(gen, *shape) * 2 + 0.5).requires_grad_(True)
        st = rnd(gen, n, 2 * c).requires_grad_(True)
        ref = O.adain_affine(O.instancenorm(a), st)
        cot = rnd(gen, *shape)
        (ref * cot).sum().backward()
        ag, sg = gpu(a).requires_grad_(True), gpu(st).requires_grad_(True)
        o = ops.instnorm_style(ag, sg)
        assert_close(o, ref, TOL)
        (o * cot.cuda()).sum().backward()
        assert_close(ag.grad, a.grad, 5e-4)
        assert_close(sg.grad, st.grad, TOL)
    z = rnd(gen, 6, 32).requires_grad_(True)
    ref = O.pixelnorm(z)
    cot = rnd(gen, 6, 32)
    (ref * cot).sum().backward()
    zg = gpu(z).requires_grad_(True)
    o = ops.pixelnorm(zg)
    assert_close(o, ref, TOL)
    (o * cot.cuda()).sum().backward()
    assert_close(zg.grad, z.grad, TOL)


@pytest.mark.parametrize('tag', ['mb8', 'mb6'])
def test_mbstd_first_and_second_order(ops, tag):
    from gan_lab_amd.utils.custom_layers import concat_mbstd_layer
    G = load_golden('ops.npz')
    x = gpu(t(G[f'{tag}_x'])).requires_grad_(True)
    y = concat_mbstd_layer(x, 4)
    assert_close(y, G[f'{tag}_y'], TOL)
    gx, = torch.autograd.grad((y * t(G[f'{tag}_cot']).cuda()).sum(), x, create_graph=True)
    assert_close(gx, G[f'{tag}_gx'], TOL)
    ggx, = torch.autograd.grad((gx * t(G[f'{tag}_cot2']).cuda()).sum(), x)
    assert_close(ggx, G[f'{tag}_ggx'], 5e-4)


def test_mbstd_known_answer(ops):
    from gan_lab_amd.utils.custom_layers import concat_mbstd_layer
    y = concat_mbstd_layer(torch.full((4, 3, 4, 4), 2.5).cuda(), 4)
    assert torch.allclose(y[:, 3].cpu(), torch.full((4, 4, 4), 1e-4), rtol=1e-5)


def test_losses_and_reductions(ops):
    G = load_golden('ops.npz')
    a, b = gpu(t(G['loss_a'])).requires_grad_(True), gpu(t(G['loss_b'])).requires_grad_(True)
    assert_close(ops.bce_logits_mean(a, 1.0), G['loss_ns_g'], TOL)
    assert_close(ops.bce_logits_mean(a, 0.0) + ops.bce_logits_mean(b, 1.0), G['loss_mm_d'], TOL)
    assert_close(ops.sum_all(a, 1 / 8) - ops.sum_all(b, 1 / 8), G['loss_wgan_d'], TOL)
    ac = t(G['loss_a']).requires_grad_(True)
    F.binary_cross_entropy_with_logits(ac, torch.ones(8)).backward()
    ops.bce_logits_mean(a, 1.0).backward()
    assert_close(a.grad, ac.grad, TOL)
    gen = torch.Generator().manual_seed(12)
    g = rnd(gen, 4, 3, 16, 16).requires_grad_(True)
    ref = ((g.norm(2, dim=1) - 1.0) ** 2).mean() * 10.0 / 2
    ref.backward()
    gg = gpu(g).requires_grad_(True)
    o = ops.chnorm_penalty(gg, 1.0, 10.0 / 2 / (4 * 16 * 16))
    assert_close(o, ref, TOL)
    o.backward()
    assert_close(gg.grad, g.grad, TOL)
    big = rnd(gen, 3, 3, 64, 64)
    assert_close(ops.sumsq_all(big.cuda(), 0.5), (big ** 2).sum() * 0.5, TOL)


def test_adam_ewma_randn(ops):
    from oracle import step as S
    gen = torch.Generator().manual_seed(21)
    p, g = rnd(gen, 1000), rnd(gen, 1000)
    st = S.new_adam_state(p)
    pg, m, v = p.cuda(), torch.zeros(1000).cuda(), torch.zeros(1000).cuda()
    pc = p.clone()
    for step in range(1, 4):
        S.adam_update(pc, g, st, 1e-3, 0.0, 0.99, 1e-8)
        ops.adam_step(pg, g.cuda(), m, v, 1e-3, 0.0, 0.99, 1e-8, 0.0, 1 - 0.0 ** step, 1 - 0.99 ** step)
        g = g * 0.5 + 0.1
    assert_close(pg - p.cuda(), pc - p, 1e-4, 'adam update')
    lag, cur = rnd(gen, 777), rnd(gen, 777)
    lg = lag.cuda()
    ops.ewma_step(lg, cur.cuda(), 0.999)
    assert_close(lg, S.ewma_update(lag, cur, 0.999), 1e-6)
    z = ops.randn((1 << 20,), 1234, 0, 'cuda')
    assert abs(z.mean().item()) < 5e-3 and abs(z.std().item() - 1) < 5e-3
    assert abs((z ** 4).mean().item() - 3.0) < 0.1
    z2 = ops.randn((1 << 20,), 1234, 0, 'cuda')
    assert torch.equal(z, z2)
    assert not torch.equal(z, ops.randn((1 << 20,), 1234, 1 << 18, 'cuda'))


def test_cpu_tensor_is_rejected(ops):
    with pytest.raises(TypeError):
        ops.blur(torch.zeros(1, 1, 4, 4))


def test_instnorm_near_constant_planes(ops):
    """|mean| >> std (the generator's constant 4x4 input + weak noise): the statistics must not lose the
    variance to fp32 cancellation, and an exactly constant plane must normalise to exactly 0."""
    from oracle import ops as O
    gen = torch.Generator().manual_seed(31)
    for shape, std in [((4, 8, 4, 4), 1e-3), ((2, 4, 64, 64), 1e-2), ((3, 5, 16, 16), 3e-4)]:
        a = (1.0 + std * torch.randn(*shape, generator=gen)).requires_grad_(True)
        st = torch.randn(shape[0], 2 * shape[1], generator=gen)
        ref = O.adain_affine(O.instancenorm(a), st)
        cot = torch.randn(*shape, generator=gen)
        (ref * cot).sum().backward()
        ag = gpu(a).requires_grad_(True)
        o = ops.instnorm_style(ag, st.cuda())
        assert_close(o, ref, TOL, f'near-constant fwd {shape} {std}')
        (o * cot.cuda()).sum().backward()
        assert_close(ag.grad, a.grad, 2e-3, f'near-constant bwd {shape} {std}')
    c = torch.full((2, 3, 4, 4), 1.0)
    o = ops.instnorm_style(c.cuda(), None)
    assert o.abs().max().item() == 0.0


S2_CASES = [
    # N, Cin, H, W, Cout, kind   (kind: 'pool' = conv3x3 -> AvgPool2d(2) -> +bias -> lrelu ; 'up' = Upsample2x -> conv3x3)
    (2, 16, 64, 64, 32, 'pool'),      # D top-layer shape (16 -> 32), thin S / thin T(dgrad) / W
    (2, 32, 32, 32, 64, 'pool'),
    (1, 64, 32, 32, 128, 'pool'),     # thick configs
    (3, 16, 40, 48, 24, 'pool'),      # ragged tiles, nothing a power of two
    (2, 128, 32, 32, 80, 'pool'),     # several K-chunks, ragged co tile
    (2, 32, 32, 32, 16, 'up'),        # G top-layer shape (32 -> 16)
    (2, 64, 16, 16, 32, 'up'),
    (1, 128, 16, 16, 64, 'up'),
    (1, 96, 20, 24, 48, 'up'),
    # rolling-window 16-tap weight gradient (wgrad_roll.hip): low-resolution width a multiple of 32
    (3, 24, 24, 128, 40, 'pool'),     # channel padding on both operands, two columns, 6 steps
    (1, 40, 6, 64, 12, 'up'),         # three steps, NBA = 2 with a ragged low-channel tile
    (2, 16, 68, 32, 16, 'up'),        # NBA = 1, 34 steps = two strips (32 + 2)
    (2, 64, 36, 64, 64, 'pool'),      # 2 x 4 channel-tile pairs
    # rolling-window S / T kernels (conv_s2_roll.hip: 16 <-> 32 channels, low-resolution width a multiple of 32)
    (3, 12, 24, 128, 20, 'pool'),     # channel padding on both sides, two column strips, 6 steps; T as its input gradient
    (1, 16, 136, 64, 32, 'pool'),     # 34 steps: several strips per column, a ragged last one
    (2, 24, 10, 64, 10, 'up'),        # T forward with channel padding (24 of 32 in, 10 of 16 out), 5 steps; S as dgrad
    (1, 32, 68, 32, 16, 'up'),        # one column, 34 steps = strips with a top and a bottom edge each
]


@pytest.mark.parametrize('case', S2_CASES, ids=[str(c) for c in S2_CASES])
def test_stride2_fused_layers(ops, case):
    """conv+avgpool and upsample+conv as 4x4 stride-2 kernels (csrc/conv_s2.hip) vs the unfused torch ops."""
    n, cin, h, w, cout, kind = case
    gen = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()))   # deterministic across processes
    x = rnd(gen, n, cin, h, w).requires_grad_(True)
    wt = rnd(gen, cout, cin, 3, 3).requires_grad_(True)
    b = rnd(gen, cout).requires_grad_(True)
    scale = 1.0 / np.sqrt(cin * 9)
    if kind == 'pool':
        y_ref = F.leaky_relu(F.avg_pool2d(F.conv2d(x * scale, wt, None, padding=1), 2) + b.view(1, -1, 1, 1), 0.2)
    else:
        y_ref = F.leaky_relu(F.conv2d(F.interpolate(x, scale_factor=2, mode='nearest') * scale, wt, b, padding=1), 0.2)
    cot = rnd(gen, *y_ref.shape)
    (y_ref * cot).sum().backward()
    xg, wg, bg = gpu(x).requires_grad_(True), gpu(wt).requires_grad_(True), gpu(b).requires_grad_(True)
    g = ops.Geom(n, cin, h, w, cout, 3, 1, up=(kind == 'up'), pool=(kind == 'pool'))
    assert g.s2, 'this shape must take the stride-2 fast path'
    y = ops.conv2d(xg, wg, bg, scale=scale, padding=1, up=(kind == 'up'), pool=(kind == 'pool'), act='lrelu')
    assert_close(y, y_ref, TOL, 'y')
    (y * cot.cuda()).sum().backward()
    assert_close(xg.grad, x.grad, TOL, 'dgrad')
    assert_close(wg.grad, wt.grad, TOL, 'wgrad')
    assert_close(bg.grad, b.grad, TOL, 'bias grad')


def test_stride2_double_backward(ops):
    """R1-shaped second order through a D down layer (conv -> pool -> bias -> lrelu) and a following conv."""
    gen = torch.Generator().manual_seed(19)
    x = rnd(gen, 2, 16, 32, 32).requires_grad_(True)
    w1 = rnd(gen, 24, 16, 3, 3).requires_grad_(True)
    b1 = rnd(gen, 24).requires_grad_(True)
    w2 = rnd(gen, 8, 24, 3, 3).requires_grad_(True)
    hmid = F.leaky_relu(F.avg_pool2d(F.conv2d(x * 0.1, w1, None, padding=1), 2) + b1.view(1, -1, 1, 1), 0.2)
    out = F.conv2d(hmid * 0.2, w2, None, padding=1)
    g, = torch.autograd.grad(out.sum(), x, create_graph=True)
    pen = (g ** 2).sum()
    pen.backward()
    xg = gpu(x).requires_grad_(True)
    w1g, b1g, w2g = gpu(w1).requires_grad_(True), gpu(b1).requires_grad_(True), gpu(w2).requires_grad_(True)
    hm = ops.conv2d(xg, w1g, b1g, scale=0.1, padding=1, pool=True, act='lrelu')
    outg = ops.conv2d(hm, w2g, None, scale=0.2, padding=1)
    gg, = torch.autograd.grad(ops.sum_all(outg), xg, create_graph=True)
    assert_close(gg, g, TOL, 'first-order grad')
    peng = ops.sumsq_all(gg)
    assert_close(peng, pen, TOL, 'penalty')
    peng.backward()
    assert_close(w1g.grad, w1.grad, 5e-4, 'ggw1')
    assert_close(w2g.grad, w2.grad, 5e-4, 'ggw2')


def test_pool_fallback_small_shapes(ops):
    """Shapes the stride-2 kernels do not cover (8x8 -> 4x4) compose the plain kernels - same result."""
    gen = torch.Generator().manual_seed(23)
    x = rnd(gen, 4, 32, 8, 8).requires_grad_(True)
    wt = rnd(gen, 32, 32, 3, 3).requires_grad_(True)
    b = rnd(gen, 32).requires_grad_(True)
    y_ref = F.leaky_relu(F.avg_pool2d(F.conv2d(x * 0.05, wt, None, padding=1), 2) + b.view(1, -1, 1, 1), 0.2)
    y_ref.sum().backward()
    xg, wg, bg = gpu(x).requires_grad_(True), gpu(wt).requires_grad_(True), gpu(b).requires_grad_(True)
    y = ops.conv2d(xg, wg, bg, scale=0.05, padding=1, pool=True, act='lrelu')
    assert_close(y, y_ref, TOL)
    ops.sum_all(y).backward()
    assert_close(xg.grad, x.grad, TOL)
    assert_close(wg.grad, wt.grad, TOL)
    assert_close(bg.grad, b.grad, TOL)


# ---------------------------------------------------------------------------------------------- #
# blur fused with its pointwise neighbours
# ---------------------------------------------------------------------------------------------- #
def _blur_ref(x):
    k = torch.tensor([1., 2., 1.], dtype=x.dtype)
    k = (k[:, None] * k[None, :] / 16.).expand(x.shape[1], 1, 3, 3)
    return F.conv2d(x, k, padding=1, groups=x.shape[1])


@pytest.mark.parametrize('shape', [(2, 5, 8, 12), (3, 16, 32, 32), (2, 4, 4, 4), (2, 3, 6, 10)])
@pytest.mark.parametrize('with_noise', [True, False])
def test_blur_bias_act_fused(ops, shape, with_noise):
    """act(blur(x) + nw*noise + b*bs): forward, all first-order gradients, and the double backward of the
    input gradient; (2,3,6,10) is not 4-column aligned and takes the composed path."""
    gen = torch.Generator().manual_seed(zlib.crc32(repr((shape, with_noise)).encode()))
    n, c, h, w = shape
    x, b = rnd(gen, *shape), rnd(gen, 1, c, 1, 1)
    nz, nw = (rnd(gen, n, 1, h, w), rnd(gen, c)) if with_noise else (None, None)
    cot, cot2 = rnd(gen, *shape), rnd(gen, *shape)

    def run(dev):
        to = (lambda v: gpu(v).requires_grad_(True)) if dev == 'gpu' else (lambda v: v.clone().requires_grad_(True))
        xs, bs = to(x), to(b)
        nws = to(nw) if with_noise else None
        if dev == 'gpu':
            y = ops.bias_act(xs, bs, gpu(nz) if with_noise else None, nws, bias_scale=0.7, act='lrelu', slope=0.2,
                             blur=True)
            c1, c2 = gpu(cot), gpu(cot2)
        else:
            pre = _blur_ref(xs) + bs * 0.7
            if with_noise:
                pre = pre + nws.view(1, c, 1, 1) * nz
            y = F.leaky_relu(pre, 0.2)
            c1, c2 = cot, cot2
        ins = [xs, bs] + ([nws] if with_noise else [])
        grads = torch.autograd.grad(y, ins, c1, create_graph=True)
        # second order: d/dx of <gx, c2> is zero almost everywhere for a piecewise-linear map, so probe the
        # linear operator instead: differentiate <gx, c2> w.r.t. the cotangent c1
        c1v = c1.clone().requires_grad_(True)
        gx2, = torch.autograd.grad(y, xs, c1v, create_graph=True)
        gc, = torch.autograd.grad((gx2 * c2).sum(), c1v)
        return [y] + list(grads) + [gc]

    for a, r, name in zip(run('gpu'), run('cpu'), ['y', 'gx', 'gb', 'gnw' if with_noise else 'adj', 'adj']):
        assert_close(a.detach().cpu(), r.detach(), TOL, name)


@pytest.mark.parametrize('case', [(2, 8, 16, 16, 8, 3), (2, 16, 32, 32, 24, 3), (2, 3, 8, 8, 16, 1),
                                  (2, 8, 6, 6, 8, 3)])
def test_conv_act_blur_fused_backward(ops, case):
    """blur(lrelu(conv(x) + b)) as the D block uses it: the fused backward (blur^T, LeakyReLU', bias gradient in
    one pass) against torch, including the R1-style double backward."""
    n, cin, hw, _, cout, ks = case
    gen = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()))
    x, w, b = rnd(gen, n, cin, hw, hw), rnd(gen, cout, cin, ks, ks) * 0.2, rnd(gen, cout) * 0.3
    cot = rnd(gen, n, cout, hw, hw)

    def run(dev):
        to = (lambda v: gpu(v).requires_grad_(True)) if dev == 'gpu' else (lambda v: v.clone().requires_grad_(True))
        xs, ws, bs = to(x), to(w), to(b)
        if dev == 'gpu':
            y = ops.conv2d(xs, ws, bs, scale=0.5, padding=ks // 2, act='lrelu', slope=0.2, blur=True)
            c1 = gpu(cot)
        else:
            y = _blur_ref(F.leaky_relu(F.conv2d(xs * 0.5, ws, bs, padding=ks // 2), 0.2))
            c1 = cot
        out = (y * c1).sum(dim=(1, 2, 3))
        gx, = torch.autograd.grad(out, xs, torch.ones_like(out), create_graph=True)
        pen = (gx ** 2).sum() * 0.5 + out.sum()
        gw, gb = torch.autograd.grad(pen, [ws, bs])
        return y, gx, gw, gb

    for a, r, name in zip(run('gpu'), run('cpu'), ['y', 'gx', 'd pen/dw', 'd pen/db']):
        assert_close(a.detach().cpu(), r.detach(), TOL, name)


@pytest.mark.parametrize('n,r', [(12, 512), (8, 1024)], ids=['12x512^2-strips-of-2', '8x1024^2-strips-of-5-and-3'])
def test_thin_layer_multi_tile_strips(ops, n, r):
    """The strip kernels (a workgroup walks several 32x8 tiles of one tile row, double-buffered patch, weights in
    registers) only engage when a layer has >= 6144 strips: 12 x 16ch x 512^2 gives strips of 2 tiles.  Forward with
    the fused bias + LeakyReLU epilogue and the input gradient (same kernel, dgrad-packed weights) against the
    oracle's conv on the CPU; the edges of every strip (zero padding, last tile of a row) are in the comparison."""
    gen = torch.Generator().manual_seed(2024)
    c = 16           # (8, 1024): the north-star layer at its real width, ragged last strip (128 tile rows = 25*5 + 3)
    x = rnd(gen, n, c, r, r)
    wt = rnd(gen, c, c, 3, 3)
    b = rnd(gen, c)
    scale = 1.0 / np.sqrt(c * 9)
    xg = gpu(x).requires_grad_(True)
    y = ops.conv2d(xg, gpu(wt), gpu(b), scale=scale, padding=1, act='lrelu', slope=0.2)
    pre = F.conv2d(x * scale, wt, b, padding=1)
    assert_close(y.detach().cpu(), F.leaky_relu(pre, 0.2), TOL, 'strip fwd')
    gy = rnd(gen, n, c, r, r)
    y.backward(gpu(gy))
    gz = gy * torch.where(y.detach().cpu() > 0, 1.0, 0.2)    # the kernel's own sign pattern (|pre| ~ 1e-7 ties)
    gx = F.conv_transpose2d(gz, wt * scale, padding=1)
    assert_close(xg.grad.cpu(), gx, TOL, 'strip dgrad')


@pytest.mark.parametrize('shape,blur', [((2, 5, 32, 32), False), ((2, 5, 32, 32), True), ((3, 4, 64, 128), True),
                                        ((1, 3, 256, 256), True), ((1, 2, 512, 512), True), ((2, 3, 40, 52), False),
                                        ((2, 3, 16, 16), True)])
def test_bias_act_with_fused_instnorm_statistics(ops, shape, blur):
    """The generator layer tail (stylegan/architectures.py:497-526): noise + bias + LeakyReLU (+ blur in front) whose
    kernel also accumulates the InstanceNorm statistics of its output.  Same result as the separate statistics pass
    (both fp64 sums of the same fp32 values) and as the float64 oracle, forward and backward; near-constant planes
    (|mean| >> std) keep their variance."""
    from oracle import ops as O
    gen = torch.Generator().manual_seed(zlib.crc32(repr((shape, blur)).encode()))
    n, c, h, w = shape
    for offset, std in [(0.0, 1.0), (3.0, 1e-2)]:
        x = (offset + std * torch.randn(*shape, generator=gen)).requires_grad_(True)
        b = (0.1 * torch.randn(c, generator=gen)).requires_grad_(True)
        nz = std * torch.randn(n, 1, h, w, generator=gen)
        nw = (0.3 * torch.randn(c, generator=gen)).requires_grad_(True)
        st = torch.randn(n, 2 * c, generator=gen)
        cot = torch.randn(*shape, generator=gen)
        xd = x.double()
        pre = (_blur_ref(xd) if blur else xd) + nw.double().view(1, c, 1, 1) * nz.double() + b.double().view(1, c, 1, 1)
        st = st.requires_grad_(True)
        ref = O.adain_affine(O.instancenorm(F.leaky_relu(pre, 0.2)), st.double())
        (ref * cot.double()).sum().backward()
        want = [x.grad.clone(), b.grad.clone(), nw.grad.clone(), st.grad.clone()]
        st = st.detach()

        outs = []
        for variant in ('stats', 'separate', 'tail'):
            xg, bg, nwg, sg = (gpu(v).requires_grad_(True) for v in (x, b, nw, st))
            if variant == 'tail':       # the whole layer tail as one autograd node (fused backward as well)
                o = ops.layer_tail(xg, bg, nz.cuda(), nwg, sg, act='lrelu', blur=blur, eps=1e-8)
            else:
                fused = variant == 'stats'
                y = ops.bias_act(xg, bg, nz.cuda(), nwg, act='lrelu', blur=blur, stats_eps=1e-8 if fused else None)
                y, stats = y if fused else (y, None)
                if fused:
                    assert (stats is not None) == (h * w >= ops.STATS_MIN_PLANE and (h * w) % 4 == 0)
                o = ops.instnorm_style(y, sg, 1e-8, stats)
            (o * cot.cuda()).sum().backward()
            outs.append((o.detach(), xg.grad, bg.grad, nwg.grad, sg.grad))
        s = outs[1]
        tol = TOL if std == 1.0 else 2e-3
        for f, tag in ((outs[0], 'fused-stats'), (outs[2], 'layer-tail')):
            assert_close(f[0], ref.float(), tol, f'{tag} fwd {shape} blur={blur} std={std}')
            assert_close(f[0], s[0], 1e-6, f'{tag} vs separate statistics {shape} blur={blur} std={std}')
            for got, sep, w_, name in zip(f[1:], s[1:], want, ('dx', 'dbias', 'dnoise_w', 'dstyle')):
                if name == 'dbias' and std != 1.0:
                    continue     # all-positive planes: a channel shift in front of the IN has an exactly zero gradient
                assert_close(got, w_.float(), tol, f'{tag} {name} {shape} blur={blur} std={std}')
                # the fused tail evaluates its element formula in fp64 for the bias / noise-weight sums (cancelling sums over
                # the plane), the separate kernels add float-rounded elements: the two agree to the fp32 noise of the latter
                assert_close(got, sep, 1e-4 if name in ('dbias', 'dnoise_w') else 1e-5,
                             f'{tag} vs separate {name} {shape} blur={blur} std={std}')


@pytest.mark.parametrize('n,cin,cout,h,w', [(2, 3, 16, 64, 64), (3, 3, 32, 64, 96), (1, 1, 8, 128, 64), (2, 3, 16, 32, 32)])
def test_fromrgb_activation_backward_folded_into_gradient_kernels(ops, n, cin, cout, h, w):
    """fromRGB (1x1 conv from the image + bias + LeakyReLU, progan/architectures.py:232-237): at >= 64x64 its backward
    takes (gy, y) straight into the dgrad / wgrad kernels (no separate gy * lrelu'(y) pass) and the bias gradient comes
    out of the weight-gradient pass.  First order against float64, and the R1-shaped second order (d/dw, d/dw2 of
    ||d out / d image||^2); the last case is below the streaming kernels' size and takes the composed path."""
    gen = torch.Generator().manual_seed(zlib.crc32(repr((n, cin, cout, h, w)).encode()))
    x = rnd(gen, n, cin, h, w).requires_grad_(True)
    w1 = rnd(gen, cout, cin, 1, 1).requires_grad_(True)
    b1 = rnd(gen, cout).requires_grad_(True)
    w2 = rnd(gen, 8, cout, 3, 3).requires_grad_(True)
    cot = rnd(gen, n, cout, h, w)

    def ref_net(x_, w1_, b1_):
        return F.leaky_relu(F.conv2d(x_.double() * 0.5, w1_.double(), b1_.double() * 0.7), 0.2)

    (ref_net(x, w1, b1) * cot.double()).sum().backward()
    want1 = [t_.grad.clone() for t_ in (x, w1, b1)]
    for t_ in (x, w1, b1):
        t_.grad = None
    out = F.conv2d(ref_net(x, w1, b1) * 0.2, w2.double(), None, padding=1)
    g, = torch.autograd.grad(out.sum(), x, create_graph=True)
    pen = (g ** 2).sum()
    pen.backward()

    xg, w1g, b1g, w2g = (gpu(t_).requires_grad_(True) for t_ in (x, w1, b1, w2))
    from gan_lab_amd import ops as raw
    geom_fused = h * w >= 4096
    y = ops.conv2d(xg, w1g, b1g, scale=0.5, bias_scale=0.7, act='lrelu')
    (y * cot.cuda()).sum().backward()
    for got, want, name in zip((xg.grad, w1g.grad, b1g.grad), want1, ('dx', 'dw', 'dbias')):
        assert_close(got, want.float(), 2e-5, f'fromRGB first order {name}')
    xg.grad = w1g.grad = b1g.grad = None
    hmid = ops.conv2d(xg, w1g, b1g, scale=0.5, bias_scale=0.7, act='lrelu')
    assert bool(raw.conv_act_bwd_fusable(raw.Geom(n, cin, h, w, cout, 1, 0))) == geom_fused
    outg = ops.conv2d(hmid, w2g, None, scale=0.2, padding=1)
    gg, = torch.autograd.grad(ops.sum_all(outg), xg, create_graph=True)
    assert_close(gg, g.float(), TOL, 'fromRGB R1 first-order grad')
    peng = ops.sumsq_all(gg)
    assert_close(peng, pen.float(), TOL, 'fromRGB R1 penalty')
    peng.backward()
    assert_close(w1g.grad, w1.grad, 5e-4, 'fromRGB R1 ggw1')
    assert_close(w2g.grad, w2.grad, 5e-4, 'fromRGB R1 ggw2')


@pytest.mark.parametrize('blur', [True, False])
def test_activation_gradient_deferred_to_next_conv(ops, blur):
    """D block k's pooled conv + LeakyReLU feeding block k+1's first conv (progan/architectures.py:280-293): the first
    layer leaves the multiplication by lrelu'(y) to the second one's input-gradient kernel (epilogue mask).  Gradients of
    every input, first order and R1-shaped second order, against float64 autograd of the plain composition."""
    gen = torch.Generator().manual_seed(77 + int(blur))
    x = rnd(gen, 2, 16, 64, 64).requires_grad_(True)
    wa, ba = rnd(gen, 32, 16, 3, 3).requires_grad_(True), rnd(gen, 1, 32, 1, 1).requires_grad_(True)
    wb, bb = rnd(gen, 32, 32, 3, 3).requires_grad_(True), rnd(gen, 32).requires_grad_(True)
    wc = rnd(gen, 8, 32, 3, 3).requires_grad_(True)
    leaves = (x, wa, ba, wb, bb, wc)

    def ref(x_):
        a = F.leaky_relu(F.avg_pool2d(F.conv2d(x_.double() * 0.3, wa.double(), None, padding=1), 2) + ba.double(), 0.2)
        b = F.leaky_relu(F.conv2d(a * 0.2, wb.double(), bb.double(), padding=1), 0.2)
        if blur:
            b = _blur_ref(b)
        return F.conv2d(b * 0.1, wc.double(), None, padding=1)

    def hip(xg, wag, bag, wbg, bbg, wcg):
        a = ops.conv2d(xg, wag, bag, scale=0.3, padding=1, act='lrelu', pool=True, defer_act_grad=True)
        assert getattr(a, ops.ACT_DEFERRED, False), 'the pooled conv should have deferred its activation gradient'
        b = ops.conv2d(a, wbg, bbg, scale=0.2, padding=1, act='lrelu', blur=blur, in_act_slope=0.2)
        return ops.conv2d(b, wcg, None, scale=0.1, padding=1)

    cot = rnd(gen, 2, 8, 32, 32)
    (ref(x) * cot.double()).sum().backward()
    want = [t_.grad.clone() for t_ in leaves]
    for t_ in leaves:
        t_.grad = None
    dev = [gpu(t_).requires_grad_(True) for t_ in leaves]
    (hip(*dev) * cot.cuda()).sum().backward()
    for got, w_, name in zip(dev, want, ('dx', 'dwa', 'dba', 'dwb', 'dbb', 'dwc')):
        assert_close(got.grad, w_.float(), 5e-5, f'deferred act grad, first order {name}')

    g, = torch.autograd.grad(ref(x).sum(), x, create_graph=True)
    pen = (g ** 2).sum()
    pen.backward()
    dev = [gpu(t_).requires_grad_(True) for t_ in leaves]
    gg, = torch.autograd.grad(ops.sum_all(hip(*dev)), dev[0], create_graph=True)
    assert_close(gg, g.float(), TOL, 'deferred act grad, R1 first-order grad')
    peng = ops.sumsq_all(gg)
    assert_close(peng, pen.float(), TOL, 'deferred act grad, R1 penalty')
    peng.backward()
    for got, t_, name in zip(dev[1:], leaves[1:], ('wa', 'ba', 'wb', 'bb', 'wc')):
        if name in ('ba', 'bb'):
            continue        # LeakyReLU'' == 0: no second-order bias gradient
        assert_close(got.grad, t_.grad.float(), 5e-4, f'deferred act grad, R1 d/d{name}')


FULL_SIZE_LAYERS = [
    # the benchmark network's own layer shapes at its FULL batch (32): Cin, Cout, H(in), ks, up, pool
    (16, 16, 1024, 3, False, False),     # north-star layer: rolling-window kernel forward / dgrad, thin wgrad
    (16, 32, 1024, 3, False, True),      # D top block: conv + 2x2 average pool as one stride-2 kernel
    (32, 16, 512, 3, True, False),       # G top block: nearest upsample + conv as one stride-2 kernel
    (3, 16, 1024, 1, False, False),      # fromRGB (streaming kernels)
    (16, 3, 1024, 1, False, False),      # toRGB
    (256, 256, 64, 3, False, False),     # the kernel with the largest share of the step
    (512, 512, 32, 3, False, True),      # thick stride-2 down layer, 16x16 output tiles
    (512, 512, 8, 3, False, False),      # split-K, S = 2
    (513, 512, 4, 3, False, False),      # split-K, S = 4, mbstd channel
]


@pytest.mark.parametrize('layer', FULL_SIZE_LAYERS, ids=[str(c) for c in FULL_SIZE_LAYERS])
def test_full_size_layers_adjointness_and_linearity(ops, layer):
    """BASELINE-size checks that need no CPU oracle (a 1024^2 batch-32 layer is 155 GFLOP): the three kernels of a layer
    must be each other's adjoints - <conv(x, w), g> = <x, dgrad(g, w)> = <w, wgrad(g, x)> - and the forward must be
    linear in x; inner products in float64 on the device.  Together with the small-size oracle parity of the same kernels
    this pins the full-size launches (tile / strip / split-K / slot decompositions only exist at these sizes)."""
    cin, cout, h, ks, up, pool = layer
    n = 32
    gen = torch.Generator(device='cuda').manual_seed(zlib.crc32(repr(layer).encode()))

    def rn(*shape):
        return torch.randn(*shape, device='cuda', generator=gen)
    x = rn(n, cin, h, h).requires_grad_(True)
    w = (rn(cout, cin, ks, ks) / np.sqrt(cin * ks * ks)).requires_grad_(True)
    pad = ks // 2
    y = ops.conv2d(x, w, None, scale=1.0, padding=pad, up=up, pool=pool)
    g = rn(*y.shape)
    y.backward(g)

    def dot(a, b):
        return (a.detach().double() * b.detach().double()).sum().item()
    lhs, via_x, via_w = dot(y, g), dot(x, x.grad), dot(w, w.grad)
    scale = np.sqrt(dot(y, y) * dot(g, g))           # |<y, g>| <= |y| |g|: errors relative to that
    assert abs(lhs - via_x) <= 2e-6 * scale, ('dgrad is not the adjoint', lhs, via_x, scale)
    assert abs(lhs - via_w) <= 2e-6 * scale, ('wgrad is not the adjoint', lhs, via_w, scale)
    with torch.no_grad():
        x2 = rn(n, cin, h, h)
        y12 = ops.conv2d(x.detach() + 2.0 * x2, w.detach(), None, scale=1.0, padding=pad, up=up, pool=pool)
        y2 = ops.conv2d(x2, w.detach(), None, scale=1.0, padding=pad, up=up, pool=pool)
        err = (y12 - (y.detach() + 2.0 * y2)).abs().max().item()
    assert err <= 2e-5 * y12.abs().max().item(), ('forward is not linear in x', err)


@pytest.mark.parametrize('blur', [False, True])
def test_full_size_layer_tail_fused_equals_composed(ops, blur):
    """The generator's top layer tail at the benchmark's full size (32 x 16 x 1024^2): the fused node (statistics in the
    bias / blur pass, LeakyReLU derivative and channel sums in the InstanceNorm backward) against the composition of the
    separate kernels on the same inputs - outputs and all four gradients - plus the size-independent property of the
    result: every plane has mean 0 and variance 1 before the style is applied."""
    gen = torch.Generator(device='cuda').manual_seed(91 + int(blur))
    n, c, r = 32, 16, 1024

    def rn(*shape):
        return torch.randn(*shape, device='cuda', generator=gen)
    x0, b0, nw0, st0 = rn(n, c, r, r), 0.1 * rn(c), 0.3 * rn(c), rn(n, 2 * c)
    nz, cot = rn(n, 1, r, r), rn(n, c, r, r)
    res = []
    for fused in (True, False):
        x, b, nw, st = (t_.clone().requires_grad_(True) for t_ in (x0, b0, nw0, st0))
        if fused:
            o = ops.layer_tail(x, b, nz, nw, st, act='lrelu', blur=blur, eps=1e-8)
        else:
            o = ops.instnorm_style(ops.bias_act(x, b, nz, nw, act='lrelu', blur=blur), st, 1e-8)
        o.backward(cot)
        res.append((o.detach(), x.grad, b.grad, nw.grad, st.grad))
        del o, x
    # dbias / dnoise_w are sums of 2^25 cancelling terms per channel: the fused tail evaluates each term and the partial
    # sums in fp64, the composed kernels add float-rounded terms, so the two differ by the composed path's rounding
    # (a few 1e-5 of the largest entry); every other output is elementwise and agrees to fp32 rounding.
    for a, bb, name in zip(res[0], res[1], ('out', 'dx', 'dbias', 'dnoise_w', 'dstyle')):
        den = bb.abs().max().item()
        tol = 1e-4 if name in ('dbias', 'dnoise_w') else 2e-5
        assert (a - bb).abs().max().item() <= tol * den, f'fused vs composed {name}'
    plain = ops.layer_tail(x0, b0, nz, nw0, None, act='lrelu', blur=blur, eps=1e-8)
    m = plain.double().mean(dim=(2, 3))
    v = plain.double().var(dim=(2, 3), unbiased=False)
    assert m.abs().max().item() < 1e-5 and (v - 1).abs().max().item() < 1e-4


def test_full_size_critic_top_fused_backward_equals_composed(ops):
    """The critic's top at the benchmark's full size (batch 32, 1024^2): fromRGB -> conv + LeakyReLU + blur -> pooled conv
    + LeakyReLU -> next block's first conv, with the LeakyReLU derivatives folded into the gradient kernels (fromRGB's
    streaming kernels, the deferred derivative in the dgrad epilogue) against the same chain with every derivative as
    its own pass: first-order gradients of all weights and of the image, and the R1-shaped second order."""
    from gan_lab_amd import ops as raw
    gen = torch.Generator(device='cuda').manual_seed(123)
    n, r = 32, 1024

    def rn(*shape):
        return torch.randn(*shape, device='cuda', generator=gen)
    img0 = rn(n, 3, r, r)
    ws0 = [rn(16, 3, 1, 1) / np.sqrt(3), rn(16, 16, 3, 3) / 12, rn(32, 16, 3, 3) / 12, rn(32, 32, 3, 3) / 17]
    bs0 = [0.1 * rn(16), 0.1 * rn(16), 0.1 * rn(1, 32, 1, 1), 0.1 * rn(32)]
    cot = rn(n, 32, r // 2, r // 2)

    def run(fused):
        img = img0.clone().requires_grad_(True)
        ws = [w_.clone().requires_grad_(True) for w_ in ws0]
        bs = [b_.clone().requires_grad_(True) for b_ in bs0]
        h = ops.conv2d(img, ws[0], bs[0], act='lrelu')                                   # fromRGB
        h = ops.conv2d(h, ws[1], bs[1], padding=1, act='lrelu', blur=True)
        h = ops.conv2d(h, ws[2], bs[2], padding=1, act='lrelu', pool=True, defer_act_grad=fused)
        assert bool(getattr(h, ops.ACT_DEFERRED, False)) == fused
        out = ops.conv2d(h, ws[3], bs[3], padding=1, act='lrelu', in_act_slope=0.2 if fused else None)
        (out * cot).sum().backward()
        first = [img.grad.clone()] + [w_.grad.clone() for w_ in ws] + [b_.grad.clone() for b_ in bs]
        for t_ in [img] + ws + bs:
            t_.grad = None
        h = ops.conv2d(img, ws[0], bs[0], act='lrelu')
        h = ops.conv2d(h, ws[1], bs[1], padding=1, act='lrelu', blur=True)
        h = ops.conv2d(h, ws[2], bs[2], padding=1, act='lrelu', pool=True, defer_act_grad=fused)
        out = ops.conv2d(h, ws[3], bs[3], padding=1, act='lrelu', in_act_slope=0.2 if fused else None)
        g, = torch.autograd.grad(ops.sum_all(out), img, create_graph=True)
        pen = ops.sumsq_all(g)
        pen.backward()
        return first, pen.detach().clone(), [w_.grad.clone() for w_ in ws]

    fused = run(True)
    saved = raw.conv_act_bwd_fusable
    raw.conv_act_bwd_fusable = lambda g: False
    try:
        comp = run(False)
    finally:
        raw.conv_act_bwd_fusable = saved
    names = ['dimg', 'dw0', 'dw1', 'dw2', 'dw3', 'db0', 'db1', 'db2', 'db3']
    for a, b_, name in zip(fused[0], comp[0], names):
        assert (a - b_).abs().max().item() <= 1e-4 * b_.abs().max().item(), f'first order {name}'
    assert abs(fused[1].item() - comp[1].item()) <= 1e-5 * abs(comp[1].item()), 'R1 penalty'
    for a, b_, name in zip(fused[2], comp[2], names[1:5]):
        assert (a - b_).abs().max().item() <= 1e-4 * b_.abs().max().item(), f'second order {name}'


def test_full_size_resampling_ops_adjoint_identities(ops):
    """32 x 16 x 1024^2: the binomial blur is self-adjoint, the 2x2 average pool and the nearest 2x upsample are adjoint up
    to the factor 4, the pool preserves the mean and the blur preserves the sum of an interior-supported image."""
    gen = torch.Generator(device='cuda').manual_seed(17)
    x = torch.randn(32, 16, 1024, 1024, device='cuda', generator=gen)
    g = torch.randn(32, 16, 1024, 1024, device='cuda', generator=gen)

    def dot(a, b):
        return (a.double() * b.double()).sum().item()
    nrm = np.sqrt(dot(x, x) * dot(g, g))
    assert abs(dot(ops.blur(x), g) - dot(x, ops.blur(g))) <= 2e-6 * nrm
    gl = torch.randn(32, 16, 512, 512, device='cuda', generator=gen)
    nrm2 = np.sqrt(dot(x, x) * dot(gl, gl))
    assert abs(dot(ops.avg_pool2(x), gl) - 0.25 * dot(x, ops.upsample2(gl))) <= 2e-6 * nrm2
    assert abs(ops.avg_pool2(x).double().mean().item() - x.double().mean().item()) <= 1e-7
    xi = torch.zeros_like(x)
    xi[:, :, 1:-1, 1:-1] = x[:, :, 1:-1, 1:-1]          # zero border: no mass leaves through the zero padding
    sb, s0 = ops.blur(xi).double().sum().item(), xi.double().sum().item()
    assert abs(sb - s0) <= 1e-6 * xi.double().abs().sum().item()


def test_integration_snippet_forward_vs_oracle(ops):
    """INTEGRATION.md's documented binding, executed verbatim through raw ctypes (no gan_lab_amd.ops in between), on a
    stand-in for the reference's ``Conv2dEx`` (utils/custom_layers.py:147-211) - against the oracle's conv2d_ex."""
    import os
    import types
    from oracle import ops as O
    from test_host_logic import ROOT, integration_snippets
    load, fwd = integration_snippets()
    ns = {}
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        exec(compile(load, 'INTEGRATION.md#1', 'exec'), ns)
        exec(compile(fwd, 'INTEGRATION.md#2', 'exec'), ns)
    finally:
        os.chdir(cwd)
    gen = torch.Generator().manual_seed(31)
    for cin, cout, res, lrmul in ((16, 16, 64, None), (48, 32, 16, 0.5)):
        x = rnd(gen, 2, cin, res, res)
        conv = torch.nn.Conv2d(cin, cout, 3, padding=1)
        with torch.no_grad():
            conv.weight.copy_(rnd(gen, cout, cin, 3, 3))
            conv.bias.copy_(rnd(gen, cout))
        ws = O.conv_wscale(conv.weight, 2.0)
        layer = types.SimpleNamespace(conv2d=conv.cuda(), equalized_lr=True, wscale=float(ws),
                                      use_lrmul=lrmul is not None, lrmul=lrmul if lrmul is not None else 1.0)
        y = ns['conv2d_ex_forward'](layer, gpu(x))
        torch.cuda.synchronize()
        ref = O.conv2d_ex(x, conv.weight.detach().cpu(), conv.bias.detach().cpu(), ws, padding=1,
                          lrmul=lrmul if lrmul is not None else 1.0)
        assert_close(y, ref, TOL, f'INTEGRATION.md forward {cin}->{cout}@{res}')


@pytest.mark.parametrize('shape', [(2, 16, 16, 64, 16), (3, 12, 24, 128, 10), (2, 16, 8, 64, 16)],
                         ids=['16->16@64x64', '12->10 ragged 24x128', 'two steps'])
def test_deferred_instancenorm_chain_equals_composed(ops, shape, monkeypatch):
    """csrc/mod.hip: a generator block's tail with the InstanceNorm + style left to its consumers - blurred layer ->
    (deferred) -> thin plain 3x3 layer that applies the affine while staging its input and has noise + bias + LeakyReLU +
    statistics in its epilogue -> (deferred) -> toRGB with per-sample weights - against the SAME chain composed from the round-1 ops
    (layer_tail: two passes; conv2d; layer_tail; conv2d 1x1): image and every gradient (input, both conv weights, biases,
    noise weights, styles).  Reference semantics: stylegan/architectures.py:497-526."""
    n, c0, h, w, c1 = shape
    gen = torch.Generator().manual_seed(zlib.crc32(repr(shape).encode()))

    def leaf(*s, scale=1.0):
        return (rnd(gen, *s) * scale).cuda().requires_grad_(True)
    x0 = leaf(n, c0, h, w)                                   # output of the up-conv of layer A
    b0, nw0, st0 = leaf(1, c0, 1, 1, scale=0.3), leaf(1, c0, 1, 1, scale=0.3), leaf(n, 2 * c0, scale=0.5)
    w1 = leaf(c1, c0, 3, 3)
    b1, nw1, st1 = leaf(1, c1, 1, 1, scale=0.3), leaf(1, c1, 1, 1, scale=0.3), leaf(n, 2 * c1, scale=0.5)
    w2, b2 = leaf(3, c1, 1, 1), leaf(3, scale=0.3)
    nz0, nz1 = rnd(gen, n, 1, h, w).cuda(), rnd(gen, n, 1, h, w).cuda()
    s1, s2 = 1.0 / np.sqrt(c0 * 9), 1.0 / np.sqrt(c1)
    cot = rnd(gen, n, 3, h, w).cuda()
    params = [x0, b0, nw0, st0, w1, b1, nw1, st1, w2, b2]

    def composed():
        a = ops.layer_tail(x0, b0, nz0, nw0, st0, act='lrelu', slope=0.2, blur=True, eps=1e-8)
        y = ops.conv2d(a, w1, None, scale=s1, padding=1)
        a = ops.layer_tail(y, b1, nz1, nw1, st1, act='lrelu', slope=0.2, blur=False, eps=1e-8)
        return ops.conv2d(a, w2, b2, scale=s2, padding=0)

    def deferred():
        d = ops.layer_tail_deferred(x0, b0, nz0, nw0, st0, act='lrelu', slope=0.2, blur=True, eps=1e-8)
        assert ops.mod_conv_ok(d, w1, 1)
        d = ops.conv_mod_tail(d, w1, s1, b1, nz1, nw1, st1, act='lrelu', slope=0.2, eps=1e-8)
        assert isinstance(d, ops.Deferred) and ops.torgb_mod_ok(d, w2)
        return ops.torgb_mod(d, w2, b2, s2)

    outs, grads = [], []
    for fn in (composed, deferred):
        img = fn()
        g = torch.autograd.grad((img * cot).sum(), params)
        outs.append(img.detach())
        grads.append([t_.detach() for t_ in g])
    assert_close(outs[1], outs[0], 2e-5, 'image')
    names = ['x0', 'bias0', 'noise_w0', 'style0', 'w1', 'bias1', 'noise_w1', 'style1', 'w_rgb', 'b_rgb']
    for name, a, b in zip(names, grads[1], grads[0]):
        assert_close(a, b, 2e-4, 'grad ' + name)
    # toRGB as the last layer's only reader: its input gradient is recomputed inside that layer's InstanceNorm backward
    # (ops.RgbGradLink, csrc/pointwise.hip RgbSrc) instead of written and read back twice
    from gan_lab_amd import _lib
    folds = bool(_lib.lib().ganlab_instnorm_bwd_rgb_supported(n, c1, 3, h * w))
    assert folds == (h * w >= 1024)
    pw_dgrads = []
    plain_dgrad = ops.k_conv_dgrad
    monkeypatch.setattr(ops, 'k_conv_dgrad', lambda gy, w_, g, sc: (pw_dgrads.append(g.ks) if g.ks == 1 else None,
                                                                     plain_dgrad(gy, w_, g, sc))[1])
    monkeypatch.setenv('GANLAB_TORGB_FOLD', '0')
    unfolded = [t_.detach() for t_ in torch.autograd.grad((deferred() * cot).sum(), params)]
    assert len(pw_dgrads) == 1
    monkeypatch.delenv('GANLAB_TORGB_FOLD')
    folded = [t_.detach() for t_ in torch.autograd.grad((deferred() * cot).sum(), params)]
    assert len(pw_dgrads) == (1 if folds else 2), 'toRGB input gradient written although the fold is supported'
    for name, a, b in zip(names, folded, unfolded):     # (the per-plane fp64 sums are taken in another order)
        assert_close(a, b, 1e-6, 'toRGB fold: grad ' + name)
    # a deferred tensor handed to a plain consumer is materialised - same numbers, same gradients
    d = ops.layer_tail_deferred(x0, b0, nz0, nw0, st0, act='lrelu', slope=0.2, blur=True, eps=1e-8)
    ref = ops.layer_tail(x0, b0, nz0, nw0, st0, act='lrelu', slope=0.2, blur=True, eps=1e-8)
    assert torch.equal(ops.materialize(d), ref)


AFF_CASES = [
    # N, C0 (deferred tensor), H, W, C1, up   - the consumers of a deferred tensor at the wider generator layers
    (2, 32, 16, 64, 32, False),     # 32 -> 32 plain: 32-channel-tile forward, rolling-window weight gradient
    (2, 64, 32, 32, 64, False),     # 64 -> 64 plain: 64-channel-tile forward, thick 8x8-tile weight gradient
    (3, 40, 24, 48, 72, False),     # ragged: channel padding in both directions, partial tiles, two co tiles
    (2, 128, 32, 32, 96, False),    # several K-chunks
    (2, 64, 16, 32, 32, True),      # up-conv 64 -> 32 (tile kernel), 16-tap rolling weight gradient on the LOW operand
    (2, 32, 16, 32, 16, True),      # up-conv 32 -> 16: the rolling-window T kernel
    (3, 24, 12, 64, 10, True),      # the same with channel padding, two column strips
    (1, 96, 20, 32, 48, True),      # ragged rows
]


@pytest.mark.parametrize('case', AFF_CASES, ids=[str(c) for c in AFF_CASES])
def test_affine_on_load_consumers_equal_materialised(ops, case):
    """Deferred InstanceNorm at the wider layers (VERDICT r02 #2b): layer A's tail leaves its output un-normalised
    (``layer_tail_deferred``), layer B's (upsample +) conv applies a*s + t while it stages its input (AFF kernels of
    conv.hip / conv_s2.hip / conv_s2_roll.hip), the weight gradient contracts the same on-the-fly tensor (conv.hip /
    wgrad_roll.hip) - against the materialised chain ``layer_tail`` -> ``conv2d``: output and every gradient (x, bias, noise
    weight, style of A; weight of B).  Reference semantics: stylegan/architectures.py:497-526 followed by :292-334."""
    n, c0, h, w, c1, up = case
    gen = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()))

    def leaf(*s, scale=1.0):
        return (rnd(gen, *s) * scale).cuda().requires_grad_(True)
    x0 = leaf(n, c0, h, w)
    b0, nw0, st0 = leaf(1, c0, 1, 1, scale=0.3), leaf(1, c0, 1, 1, scale=0.3), leaf(n, 2 * c0, scale=0.5)
    w1 = leaf(c1, c0, 3, 3)
    nz0 = rnd(gen, n, 1, h, w).cuda()
    s1 = 1.0 / np.sqrt(c0 * 9)
    cot = rnd(gen, n, c1, 2 * h if up else h, 2 * w if up else w).cuda()
    params = [x0, b0, nw0, st0, w1]

    def composed():
        a = ops.layer_tail(x0, b0, nz0, nw0, st0, act='lrelu', slope=0.2, blur=True, eps=1e-8)
        return ops.conv2d(a, w1, None, scale=s1, padding=1, up=up)

    def deferred():
        d = ops.layer_tail_deferred(x0, b0, nz0, nw0, st0, act='lrelu', slope=0.2, blur=True, eps=1e-8)
        assert ops.conv_aff_ok(d.a.shape, w1, up, 1), 'this shape must take the affine-on-load kernels'
        return ops.conv_aff(d, w1, s1, up=up)

    outs, grads = [], []
    for fn in (composed, deferred):
        y = fn()
        g = torch.autograd.grad((y * cot).sum(), params)
        outs.append(y.detach())
        grads.append([t_.detach() for t_ in g])
    assert_close(outs[1], outs[0], 2e-5, 'output')
    for name, a, b in zip(['x0', 'bias0', 'noise_w0', 'style0', 'w1'], grads[1], grads[0]):
        assert_close(a, b, 2e-4, 'grad ' + name)


def test_batched_repack_equals_single_packs(ops):
    """csrc/pack.hip: after the fused optimiser reports the memory it rewrote, the FIRST request for a stale packed weight
    re-packs every cached entry of that range in one launch - bit-identical to packing each weight on its own, entries
    outside the range untouched (same buffers), new values picked up (Conv2dEx.forward re-scales on every call,
    custom_layers.py:202-211; here the scaled re-layout is cached between optimiser steps)."""
    from gan_lab_amd import _lib
    gen = torch.Generator().manual_seed(5)
    arena = torch.zeros(200000, device='cuda')
    other = (rnd(gen, 24, 16, 3, 3)).cuda()                       # a weight OUTSIDE the rewritten range
    shapes = [(32, 16, 3, 3), (16, 32, 3, 3), (64, 64, 3, 3), (3, 16, 1, 1), (40, 24, 3, 3)]
    ws, off = [], 0
    for sh in shapes:
        n = int(np.prod(sh))
        ws.append(arena[off:off + n].view(sh))
        ws[-1].copy_(rnd(gen, *sh))
        off += (n + 3) // 4 * 4

    def requests():
        out = []
        for w in ws:
            if w.shape[2] == 3:
                out += [ops._packed(w, _lib.PACK_FWD, 0.1), ops._packed(w, _lib.PACK_DGRAD, 0.1),
                        ops._packed(w, _lib.PACK_FWD, 0.2, s2_up=1), ops._packed(w, _lib.PACK_DGRAD, 0.2, s2_up=1),
                        ops._packed(w, _lib.PACK_FWD, 0.3, s2_up=0), ops._packed(w, _lib.PACK_DGRAD, 0.3, s2_up=0)]
                if w.shape[0] % 64 == 0 and w.shape[1] % 64 == 0:
                    out += [ops._packed_bf16(w, _lib.PACK_FWD, 0.5), ops._packed_bf16(w, _lib.PACK_DGRAD, 0.5)]
            else:
                out += [ops._packed(w, _lib.PACK_FWD, 0.1), ops._packed(w, _lib.PACK_DGRAD, 0.1)]
        return out
    ops.bump_weight_epoch()
    first = requests()
    o_other = ops._packed(other, _lib.PACK_FWD, 0.7)
    o_other_before = o_other.clone()
    # the "optimiser": rewrite the arena through a raw pointer, like FusedAdam's kernel (an in-place torch op would bump the
    # version counter the views share with their base, i.e. new cache keys), and report the range
    check = ops.check
    check(_lib.lib().ganlab_axpby_f32(ops._p(arena), None, ops._p(arena), arena.numel(), -1.5, 0.0, ops._st()), 'axpby')
    ops.bump_weight_epoch([(arena.data_ptr(), arena.data_ptr() + arena.numel() * 4)])
    again = requests()                                              # first call re-packs the whole range in one launch
    assert all(a.data_ptr() == b.data_ptr() for a, b in zip(first, again)), 're-packed in place'
    assert ops._packed(other, _lib.PACK_FWD, 0.7).data_ptr() == o_other.data_ptr() and torch.equal(o_other, o_other_before)
    batched = [t_.clone() for t_ in again]
    ops.bump_weight_epoch()                                         # drop everything: single-weight kernels
    single = requests()
    assert len(single) == len(batched) >= 20
    for a, b in zip(batched, single):
        assert a.dtype == b.dtype and torch.equal(a, b)
    ops.bump_weight_epoch()


@pytest.mark.parametrize('shape', [(2, 16, 32, 64), (3, 5, 24, 32), (1, 8, 256, 64), (2, 4, 6, 96)],
                         ids=['16ch 32x64', 'odd channels 24x32', '8-row strips', '2-row strips'])
def test_leaky_relu_mask_bits_equal_float_masks(ops, shape, monkeypatch):
    """The critic's conv -> bias -> LeakyReLU -> blur with the LeakyReLU's sign kept as BITS (written by the blur pass,
    csrc/pointwise.hip) instead of the float tensor: forward, first-order gradients (input, weight, bias) and the R1-shaped
    second order are BIT-identical to the float-mask path (progan/architectures.py:261-284; nn.LeakyReLU backward)."""
    n, c, h, w = shape
    gen = torch.Generator().manual_seed(zlib.crc32(repr(shape).encode()))
    x0 = rnd(gen, n, c, h, w)
    wt0, b0 = rnd(gen, c, c, 3, 3), rnd(gen, c)
    cot = rnd(gen, n, c, h, w).cuda()
    monkeypatch.setenv('GANLAB_ROLL_BLUR', '0')        # the composed form (the fused kernel has its own test below)

    def run(bits_on):
        monkeypatch.setattr(ops, '_MASK_BITS', [bits_on])
        x = x0.cuda().requires_grad_(True)
        wt, b = wt0.cuda().requires_grad_(True), b0.cuda().requires_grad_(True)
        y = ops.conv2d(x, wt, b, scale=0.1, padding=1, act='lrelu', blur=True)
        gx, = torch.autograd.grad((y * cot).sum(), x, create_graph=True)
        pen = (gx ** 2).sum()                                       # R1-shaped: differentiate the input gradient again
        gw2, gb2 = torch.autograd.grad(pen, (wt, b), retain_graph=True, allow_unused=True)
        gw1, gb1 = torch.autograd.grad((y * cot).sum(), (wt, b))
        return [t_.detach() for t_ in (y, gx, gw1, gb1, gw2)]
    a, b_ = run(True), run(False)
    assert ops.mask_bits_ok(x0.cuda()) or not ops._MASK_BITS[0]
    for name, u, v in zip(['y', 'gx', 'gw', 'gb', 'gw (second order)'], a, b_):
        assert torch.equal(u, v), name


@pytest.mark.parametrize('shape', [(2, 16, 16, 64, 64), (1, 16, 16, 8, 64), (3, 5, 7, 24, 128), (2, 16, 16, 256, 192),
                                   (1, 3, 16, 132, 64), (2, 16, 9, 4, 64)],
                         ids=['16ch 64x64', 'two steps', 'odd channels 24x128', 'row strips 256x192', 'ragged strips 132x64',
                              'single step'])
def test_conv_lrelu_blur_fused_kernel_equals_composed(ops, shape, monkeypatch):
    """csrc/conv_roll_blur.hip: conv3x3 -> +bias -> LeakyReLU -> blur of a thin layer in ONE rolling-window kernel (the blur
    pass folded into the convolution, sign bits written beside it) against the composed form - conv kernel, then the blur
    pass that emits the bits (progan/architectures.py:254-284).  The two sum the 144 products of an output in different
    orders: outputs and gradients agree to fp32 rounding, the sign bits wherever the pre-activation is not rounding noise,
    and the float64 CPU evaluation is as close to one as to the other."""
    n, cin, cout, h, w = shape
    gen = torch.Generator().manual_seed(zlib.crc32(repr(shape).encode()))
    x0, wt0, b0 = rnd(gen, n, cin, h, w), rnd(gen, cout, cin, 3, 3), rnd(gen, cout)
    cot = rnd(gen, n, cout, h, w).cuda()
    g = ops.Geom(n, cin, h, w, cout, 3, 1, 0)

    def run(fused):
        monkeypatch.setenv('GANLAB_ROLL_BLUR', '1' if fused else '0')
        x = x0.cuda().requires_grad_(True)
        wt, b = wt0.cuda().requires_grad_(True), b0.cuda().requires_grad_(True)
        if fused:
            direct = ops.k_conv_fwd_blur_bits(x.detach(), wt.detach(), b.detach(), g, 0.1, 1.0, 0.2)
            assert direct is not None, 'the fused kernel refused a geometry it documents'
        y = ops.conv2d(x, wt, b, scale=0.1, padding=1, act='lrelu', blur=True)
        gx, gw, gb = torch.autograd.grad((y * cot).sum(), (x, wt, b))
        return [t_.detach() for t_ in (y, gx, gw, gb)] + ([direct[1]] if fused else [])
    a, b_ = run(True), run(False)
    # float64 reference of the forward
    pre = F.conv2d(x0.double(), wt0.double() * 0.1, b0.double(), padding=1)
    act = F.leaky_relu(pre, 0.2)
    k = torch.tensor([1., 2., 1.], dtype=torch.float64)
    k2 = (k[:, None] * k[None, :] / 16).expand(cout, 1, 3, 3)
    ref = F.conv2d(act, k2, padding=1, groups=cout)
    e_f, e_c = (a[0].cpu().double() - ref).abs().max().item(), (b_[0].cpu().double() - ref).abs().max().item()
    assert e_f <= max(2 * e_c, 1e-6 * ref.abs().max().item()), (e_f, e_c)
    for name, u, v in zip(['y', 'gx', 'gw', 'gb'], a, b_):
        assert (u - v).abs().max().item() <= 2e-5 * v.abs().max().item(), name
    # the sign bits: bit e of the NCHW-linear index, wrong only where the pre-activation is rounding noise
    bits = a[4].cpu().numpy().view(np.uint32)
    got = ((bits[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).reshape(-1).astype(bool)
    want = (pre > 0).reshape(-1).numpy()
    bad = got != want
    assert (pre.reshape(-1).numpy()[bad].__abs__() < 1e-5).all() and bad.mean() < 1e-4, int(bad.sum())


@pytest.mark.parametrize('shape', [(2, 32, 32, 32, 64), (1, 48, 80, 16, 96), (2, 64, 64, 8, 32), (3, 17, 24, 24, 64)],
                         ids=['32->32 32x64', '48->80 16x96', '64->64 8x32', '17->24 24x64'])
def test_conv_aff_tail_epilogue_equals_composed(ops, shape):
    """conv.hip, TAIL epilogue of the affine-on-load tile kernels: a plain 3x3 generator layer with a deferred-InstanceNorm
    input - conv, + noise, + bias, LeakyReLU and the InstanceNorm statistics of the result (stylegan/architectures.py:497-526)
    - in ONE kernel, against the composed form (affine-on-load conv kernel, then the bias / noise / act / statistics pass):
    activation, the (s, t) of the deferred output, and every gradient of a loss on the normalised output."""
    n, cin, cout, h, w = shape
    gen = torch.Generator().manual_seed(zlib.crc32(repr(shape).encode()) + 7)
    x0, w0 = rnd(gen, n, cin, h, w), rnd(gen, cout, cin, 3, 3)
    b0, nw0, st0 = rnd(gen, 1, cout, 1, 1), rnd(gen, 1, cout, 1, 1), rnd(gen, n, 2 * cout)
    nz = rnd(gen, n, 1, h, w).cuda()
    s_in, t_in = (rnd(gen, n, cin) * 0.5 + 1).cuda(), rnd(gen, n, cin).cuda()
    cot = rnd(gen, n, cout, h, w).cuda()
    assert ops.conv_tail_shape_ok((n, cin, h, w), w0) and not ops.mod_conv_shape_ok((n, cin, h, w), w0)

    def run(fused):
        x, wt = x0.cuda().requires_grad_(True), w0.cuda().requires_grad_(True)
        b, nw, st = (t_.cuda().requires_grad_(True) for t_ in (b0, nw0, st0))
        src = ops.Deferred(x, s_in, t_in, None, None, None)
        if fused:
            d = ops.conv_mod_tail(src, wt, 0.06, b, nz, nw, st, bias_scale=1.0, act='lrelu', slope=0.2, eps=1e-8)
        else:
            c = ops.conv_aff(src, wt, 0.06, up=False)
            d = ops.layer_tail_deferred(c, b, nz, nw, st, bias_scale=1.0, act='lrelu', slope=0.2, blur=False, eps=1e-8)
        out = ops.materialize(d)
        (out * cot).sum().backward()
        return [t_.detach() for t_ in (d.a, d.s, d.t, out, x.grad, wt.grad, b.grad, nw.grad, st.grad)]
    a, b_ = run(True), run(False)
    for name, u, v in zip(['a', 's', 't', 'normalised output', 'gx', 'gw', 'gbias', 'gnoise_w', 'gstyle'], a, b_):
        assert (u - v).abs().max().item() <= 5e-5 * v.abs().max().item(), name


S2_BLUR_CASES = [(2, 32, 16, 16, 32), (1, 24, 9, 6, 64), (2, 32, 16, 128, 32), (1, 17, 16, 4, 96)]
S2_BLUR_IDS = ['32->16 16x32', '24->9 6x64', 'row strips 128x32', '17->16 two steps 4x96']


@pytest.mark.parametrize('deferred', [False, True], ids=['plain input', 'deferred input'])
@pytest.mark.parametrize('shape', S2_BLUR_CASES, ids=S2_BLUR_IDS)
def test_upconv_blur_tail_fused_kernel_equals_composed(ops, shape, deferred, monkeypatch):
    """csrc/conv_s2_roll_blur.hip, TB_TAIL: the generator layer that opens a resolution - Upsample -> conv3x3 -> blur ->
    +noise -> +bias -> LeakyReLU with the InstanceNorm statistics (stylegan/architectures.py:292-334, 497-526) - in one pass,
    against the composed form (stride-2 transposed kernel, then the fused blur + tail pass): activation, the (s, t) of the
    deferred InstanceNorm + style, and every gradient (input, weight, bias, noise weight, style) of a loss on the NORMALISED
    output."""
    n, cl, ch, hl, wl = shape
    gen = torch.Generator().manual_seed(zlib.crc32(repr((shape, deferred)).encode()))
    x0, w0 = rnd(gen, n, cl, hl, wl), rnd(gen, ch, cl, 3, 3)
    b0, nw0, st0 = rnd(gen, 1, ch, 1, 1), rnd(gen, 1, ch, 1, 1), rnd(gen, n, 2 * ch)
    nz = rnd(gen, n, 1, 2 * hl, 2 * wl).cuda()
    s_in, t_in = (rnd(gen, n, cl) * 0.5 + 1).cuda(), rnd(gen, n, cl).cuda()
    cot = rnd(gen, n, ch, 2 * hl, 2 * wl).cuda()
    assert ops.upconv_blur_tail_ok((n, cl, hl, wl), w0, deferred), 'the fused kernel refused a geometry it documents'

    def run(fused):
        x, w = x0.cuda().requires_grad_(True), w0.cuda().requires_grad_(True)
        b, nw, st = (t_.cuda().requires_grad_(True) for t_ in (b0, nw0, st0))
        src = ops.Deferred(x, s_in, t_in, None, None, None) if deferred else x
        if fused:
            d = ops.upconv_blur_tail(src, w, 0.07, b, nz, nw, st, bias_scale=1.0, act='lrelu', slope=0.2, eps=1e-8)
        else:
            monkeypatch.setenv('GANLAB_S2_ROLL_BLUR', '0')
            c = ops.conv_aff(src, w, 0.07, up=True) if deferred else ops.conv2d(x, w, None, scale=0.07, padding=1, up=True)
            d = ops.layer_tail_deferred(c, b, nz, nw, st, bias_scale=1.0, act='lrelu', slope=0.2, blur=True, eps=1e-8)
            monkeypatch.delenv('GANLAB_S2_ROLL_BLUR')
        out = ops.materialize(d)
        (out * cot).sum().backward()
        return [t_.detach() for t_ in (d.a, d.s, d.t, out, x.grad, w.grad, b.grad, nw.grad, st.grad)]
    a, b_ = run(True), run(False)
    for name, u, v in zip(['a', 's', 't', 'normalised output', 'gx', 'gw', 'gbias', 'gnoise_w', 'gstyle'], a, b_):
        assert (u - v).abs().max().item() <= 5e-5 * v.abs().max().item(), name


@pytest.mark.parametrize('shape', S2_BLUR_CASES, ids=S2_BLUR_IDS)
def test_pooled_conv_dgrad_blur_act_fused_kernel_equals_composed(ops, shape, monkeypatch):
    """csrc/conv_s2_roll_blur.hip, TB_MASK: the critic's  conv -> LeakyReLU -> blur -> conv -> AvgPool -> bias -> LeakyReLU
    (progan/architectures.py:254-284) with the blur^T + LeakyReLU' + bias-gradient pass of the first layer folded into the
    pooled conv's input-gradient kernel (ops.BlurHandoff) against the separate passes: outputs, first-order gradients of
    both layers and the R1-shaped second order."""
    n, cl, ch, hl, wl = shape           # the pooled conv maps ch (high) -> cl (low) channels
    gen = torch.Generator().manual_seed(zlib.crc32(repr(shape).encode()) + 1)
    x0 = rnd(gen, n, ch, 2 * hl, 2 * wl)
    wa0, ba0 = rnd(gen, ch, ch, 3, 3), rnd(gen, ch)
    wb0, bb0 = rnd(gen, cl, ch, 3, 3), rnd(gen, 1, cl, 1, 1)
    cot = rnd(gen, n, cl, hl, wl).cuda()

    def run(fused):
        monkeypatch.setenv('GANLAB_S2_ROLL_BLUR', '1' if fused else '0')
        x = x0.cuda().requires_grad_(True)
        wa, ba, wb, bb = (t_.cuda().requires_grad_(True) for t_ in (wa0, ba0, wb0, bb0))
        y = ops.conv2d(x, wa, ba, scale=0.08, padding=1, act='lrelu', blur=True)
        h = getattr(y, ops.BLUR_HANDOFF, None)
        assert h is not None and h.bits is not None
        z = ops.conv2d(y, wb, bb, scale=0.06, padding=1, act='lrelu', pool=True, in_blur_handoff=h)
        gx, = torch.autograd.grad((z * cot).sum(), x, create_graph=True)
        pen = (gx ** 2).sum()
        second = torch.autograd.grad(pen, (wa, wb), retain_graph=True)
        first = torch.autograd.grad((z * cot).sum(), (wa, ba, wb, bb))
        return [t_.detach() for t_ in (z, gx) + tuple(first) + tuple(second)]
    a, b_ = run(True), run(False)
    names = ['z', 'gx', 'gwA', 'gbA', 'gwB', 'gbB', 'gwA (second order)', 'gwB (second order)']
    for name, u, v in zip(names, a, b_):
        assert (u - v).abs().max().item() <= 5e-5 * v.abs().max().item(), name


@pytest.mark.parametrize('blur', [False, True], ids=['plain first conv', 'first conv + blur'])
@pytest.mark.parametrize('shape', [(2, 64, 64, 16), (1, 32, 128, 16), (3, 132, 64, 9)], ids=['64x64', '32x128', '132x64, 9 channels'])
def test_fromrgb_backward_folded_into_first_conv_input_gradient(ops, shape, blur, monkeypatch):
    """csrc/conv_roll_blur.hip, RB_RGB (ops.RgbHandoff): where nobody needs d / d image, the critic's first 3x3 conv finishes
    the backward of the fromRGB layer in front of it inside its input-gradient kernel - mask by the sign bits, weight / bias
    gradient sums against the image, straight into the arena slots - and the gradient tensor between the two layers is never
    written.  Against the separate kernels: every parameter gradient of both layers, also when a second backward of the
    same step accumulates on top (the critic sees two batches per step), and the fallback when the image wants its gradient."""
    from torch import nn
    from gan_lab_amd.optim import ParamArena
    n, h, w, c = shape
    gen = torch.Generator().manual_seed(zlib.crc32(repr((shape, blur)).encode()))
    img0, img1 = rnd(gen, n, 3, h, w), rnd(gen, n, 3, h, w)
    wr, br = nn.Parameter(rnd(gen, c, 3, 1, 1).cuda()), nn.Parameter(rnd(gen, c).cuda())
    wa, ba = nn.Parameter(rnd(gen, c, c, 3, 3).cuda()), nn.Parameter(rnd(gen, c).cuda())
    arena = ParamArena([('wr', wr), ('br', br), ('wa', wa), ('ba', ba)])
    cot = rnd(gen, n, c, h, w).cuda()

    def sweep(img):
        h0 = ops.conv2d(img, wr, br, scale=0.5, act='lrelu')
        rgb = getattr(h0, ops.RGB_HANDOFF, None)
        assert rgb is not None and rgb.bits is not None
        y = ops.conv2d(h0, wa, ba, scale=0.1, padding=1, act='lrelu', blur=blur, in_rgb_handoff=rgb)
        with ops.direct_param_grads(True):
            (y * cot).sum().backward()

    def run(fold, want_img_grad=False):
        monkeypatch.setenv('GANLAB_RGB_FOLD', '1' if fold else '0')
        arena.zero_grad()
        a = img0.cuda().requires_grad_(want_img_grad)
        sweep(a)
        first = arena.gflat.clone()
        sweep(img1.cuda())                 # second batch of the step: accumulates
        return first, arena.gflat.clone(), (a.grad.clone() if want_img_grad else None)
    launched, separate = [], []
    orig, orig_w = ops.k_conv_dgrad_rgb_sums, ops.k_conv_wgrad_act
    monkeypatch.setattr(ops, 'k_conv_dgrad_rgb_sums', lambda *a_, **k_: (launched.append(1), orig(*a_, **k_))[1])
    monkeypatch.setattr(ops, 'k_conv_wgrad_act', lambda *a_, **k_: (separate.append(1), orig_w(*a_, **k_))[1])
    f1, f2, _ = run(True)
    assert len(launched) == 2, 'the folded kernel did not run'
    assert not separate, "fromRGB's own backward ran as well (on a materialised zero gradient)"
    u1, u2, _ = run(False)
    assert len(launched) == 2 and len(separate) == 2
    for name, a_, b_ in (('first backward', f1, u1), ('accumulated', f2, u2)):
        for k, o, sz in zip(arena.names, arena.offsets, arena.sizes):
            ref = b_[o:o + sz]
            assert (a_[o:o + sz] - ref).abs().max().item() <= 2e-5 * ref.abs().max().item(), (name, k)
    g1, _, gi1 = run(True, want_img_grad=True)       # the image's gradient is wanted: the separate kernels run
    assert len(launched) == 3 and gi1 is not None     # (only the second sweep - a plain image - folds)
    g0, _, gi0 = run(False, want_img_grad=True)
    assert torch.equal(gi1, gi0) and (g1 - g0).abs().max().item() <= 2e-5 * g0.abs().max().item()


@pytest.mark.parametrize('shape', [(2, 3, 64, 64, 16), (3, 3, 64, 96, 24)], ids=['3->16 64x64', '3->24 64x96'])
def test_fromrgb_mask_bits_equal_float_masks(ops, shape, monkeypatch):
    """fromRGB (1x1 conv + LeakyReLU, progan/architectures.py:286-292) with its mask as bits: the forward writes y and
    the sign bits of y, its streaming gradient kernels read the bits - forward, first-order gradients and the R1-shaped
    second order are BIT-identical to the float-mask path."""
    n, cin, h, w, cout = shape
    gen = torch.Generator().manual_seed(zlib.crc32(repr(shape).encode()))
    x0, wt0, b0 = rnd(gen, n, cin, h, w), rnd(gen, cout, cin, 1, 1), rnd(gen, cout)
    cot = rnd(gen, n, cout, h, w).cuda()

    def run(bits_on):
        monkeypatch.setattr(ops, '_MASK_BITS', [bits_on])
        x = x0.cuda().requires_grad_(True)
        wt, b = wt0.cuda().requires_grad_(True), b0.cuda().requires_grad_(True)
        g = ops.Geom(n, cin, h, w, cout, 1, 0)
        assert ops.conv_act_bwd_fusable(g), 'this shape must take the streaming fromRGB kernels'
        y = ops.conv2d(x, wt, b, scale=0.5, padding=0, act='lrelu')
        gx, = torch.autograd.grad((y * cot).sum(), x, create_graph=True)
        gw2, = torch.autograd.grad((gx ** 2).sum(), wt, retain_graph=True)
        gw1, gb1 = torch.autograd.grad((y * cot).sum(), (wt, b))
        return [t_.detach() for t_ in (y, gx, gw1, gb1, gw2)]
    a, b_ = run(True), run(False)
    for name, u, v in zip(['y', 'gx', 'gw', 'gb', 'gw (second order)'], a, b_):
        assert torch.equal(u, v), name


@pytest.mark.parametrize('mode,scale,tmode,align', [('bilinear_up', 2, 'bilinear', False), ('bilinear_up', 2, 'bilinear', True),
                                                    ('bilinear_down', .5, 'bilinear', True),
                                                    ('bilinear_down', .5, 'bilinear', False),
                                                    ('nearest_down', .5, 'nearest', False)])
def test_resampler_variants_match_interpolate(ops, mode, scale, tmode, align):
    """nn.Upsample(mode='bilinear') / BilinearPool2d / NearestPool2d (custom_layers.py:59-75): the table-driven kernel
    against F.interpolate on the CPU - output, input gradient (adjoint gather) and the R1-shaped second order (the
    gradient of |d out / d x|^2-like functionals reaches the pooler's forward again)."""
    gen = torch.Generator().manual_seed(5)
    for shape in ((2, 3, 4, 4), (3, 5, 16, 16), (1, 2, 6, 10), (2, 16, 64, 64)):
        x0 = torch.randn(*shape, generator=gen)
        kw = {} if tmode == 'nearest' else {'align_corners': align}
        xr = x0.clone().requires_grad_(True)
        yr = F.interpolate(xr, scale_factor=scale, mode=tmode, **kw)
        cot = torch.randn(yr.shape, generator=gen)
        xg = gpu(x0).requires_grad_(True)
        yg = ops.resample(xg, mode, align)
        assert_close(yg, yr, 1e-5, f'{mode} forward {shape}')
        # second order: L = sum (d<y, cot * y>/dx)^2
        gr, = torch.autograd.grad((yr * yr * cot).sum(), xr, create_graph=True)
        gg, = torch.autograd.grad((yg * yg * gpu(cot)).sum(), xg, create_graph=True)
        assert_close(gg, gr, 1e-5, f'{mode} input gradient {shape}')
        (gr ** 2).sum().backward()
        (gg ** 2).sum().backward()
        assert_close(xg.grad, xr.grad, 1e-4, f'{mode} second order {shape}')
