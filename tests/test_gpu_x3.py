"""Split-product (3 x bf16) conv kernels (csrc/conv_x3.hip) through the C-ABI: every form against float64 on the CPU and
against the exact-fp32 MFMA kernels of the same entry points, at the accuracy bar of the fp32 kernels (TOL of test_gpu_ops).
The reference computes F.conv2d in fp32 (utils/custom_layers.py:202-211); the bar that the split products are 'fp32' is the
rms distance to float64, which must not exceed ATen's own (tools/x3_bench.py; here: <= 1.1 x the exact kernels')."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from util import assert_close

pytestmark = pytest.mark.gpu
TOL = 2e-4


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    from gan_lab_amd import ops as _ops, _lib
    _lib.lib()
    return _ops


def rms_rel(a, ref):
    a, ref = a.double().cpu(), ref.double().cpu()
    return ((a - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()


def launched(ops):
    from gan_lab_amd import _lib
    return _lib.last_launch()[0] or ''


# N, Cin, Cout, H, W: multi-tile per workgroup (more tiles than CUs), one / two / eight 64-channel double chunks, 2+ co tiles,
# non-square, the ring wrapping across tiles with an odd number of stages per tile (64 channels)
X3_CASES = [(2, 64, 64, 16, 16), (3, 128, 192, 32, 16), (2, 256, 128, 16, 48), (1, 512, 64, 16, 16), (5, 64, 128, 64, 64),
            (9, 128, 64, 32, 32)]


@pytest.mark.parametrize('case', X3_CASES)
def test_x3_forward_and_input_gradient(ops, case):
    n, ci, co, h, w = case
    g = torch.Generator().manual_seed(11 + n + ci)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g)
    b = torch.randn(co, generator=g)
    gy = torch.randn(n, co, h, w, generator=g)
    geom = ops.Geom(n, ci, h, w, co, 3, 1)
    scale = 1.0 / (3 * ci ** 0.5)
    old_min, ops._X3_MIN_TILES = ops._X3_MIN_TILES, 1
    try:
        assert ops.x3_ok(geom) and ops.x3_ok(geom, True)
        y3 = ops.k_conv_fwd(x.cuda(), wt.cuda(), b.cuda(), geom, scale, 0.5, ops.ACT_LRELU, 0.2)
        assert 'conv_x3_fwd_kernel' in launched(ops)
        gx3 = ops.k_conv_dgrad(gy.cuda(), wt.cuda(), geom, scale)
        assert 'conv_x3_fwd_kernel' in launched(ops)
        prev = ops.set_x3(False)
        try:
            y1 = ops.k_conv_fwd(x.cuda(), wt.cuda(), b.cuda(), geom, scale, 0.5, ops.ACT_LRELU, 0.2)
            assert 'conv_x3_fwd_kernel' not in launched(ops)
            gx1 = ops.k_conv_dgrad(gy.cuda(), wt.cuda(), geom, scale)
        finally:
            ops.set_x3(prev)
    finally:
        ops._X3_MIN_TILES = old_min
    yd = F.leaky_relu(F.conv2d(x.double(), wt.double() * scale, None, padding=1) + 0.5 * b.double().view(1, -1, 1, 1), 0.2)
    xd = x.double().requires_grad_(True)
    gxd, = torch.autograd.grad(F.conv2d(xd, wt.double() * scale, padding=1), xd, gy.double())
    assert_close(y3.cpu(), yd, TOL, 'x3 forward vs float64')
    assert_close(gx3.cpu(), gxd, TOL, 'x3 input gradient vs float64')
    assert_close(y3.cpu(), y1.cpu(), TOL, 'x3 forward vs exact-fp32 kernel')
    assert_close(gx3.cpu(), gx1.cpu(), TOL, 'x3 input gradient vs exact-fp32 kernel')
    # "fp32" means: no further from float64 than the exact-fp32 kernels (rms over the tensor)
    assert rms_rel(y3, yd) <= 1.1 * rms_rel(y1, yd), (rms_rel(y3, yd), rms_rel(y1, yd))
    assert rms_rel(gx3, gxd) <= 1.1 * rms_rel(gx1, gxd), (rms_rel(gx3, gxd), rms_rel(gx1, gxd))


def test_x3_masked_input_gradient(ops):
    n, ci, co, h, w = 3, 128, 64, 32, 32
    g = torch.Generator().manual_seed(5)
    x = F.leaky_relu(torch.randn(n, ci, h, w, generator=g), 0.2)       # the conv's input: a LeakyReLU output
    wt = torch.randn(co, ci, 3, 3, generator=g)
    gy = torch.randn(n, co, h, w, generator=g)
    geom = ops.Geom(n, ci, h, w, co, 3, 1)
    old_min, ops._X3_MIN_TILES = ops._X3_MIN_TILES, 1
    try:
        gx3 = ops.k_conv_dgrad_mask(gy.cuda(), wt.cuda(), x.cuda(), geom, 0.03, 0.2)
        assert 'conv_x3_fwd_kernel' in launched(ops)
    finally:
        ops._X3_MIN_TILES = old_min
    xd = torch.zeros(n, ci, h, w, dtype=torch.float64, requires_grad=True)
    gxd, = torch.autograd.grad(F.conv2d(xd, wt.double() * 0.03, padding=1), xd, gy.double())
    gxd = gxd * torch.where(x > 0, 1.0, 0.2).double()
    assert_close(gx3.cpu(), gxd, TOL, 'x3 masked input gradient vs float64')


@pytest.mark.parametrize('noise', [True, False])
def test_x3_affine_on_load_and_tail(ops, noise):
    from gan_lab_amd import _lib
    n, ci, co, h, w = 3, 128, 128, 32, 32
    g = torch.Generator().manual_seed(7)
    a = torch.randn(n, ci, h, w, generator=g)
    s_ = torch.rand(n, ci, generator=g) + 0.5
    t_ = torch.randn(n, ci, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g)
    b = torch.randn(co, generator=g)
    nz = torch.randn(n, 1, h, w, generator=g) if noise else None
    nw = torch.randn(co, generator=g) if noise else None
    geom = ops.Geom(n, ci, h, w, co, 3, 1)
    scale, eps = 0.02, 1e-8
    bd = a.double() * s_.double().view(n, ci, 1, 1) + t_.double().view(n, ci, 1, 1)
    conv = F.conv2d(bd, wt.double() * scale, None, padding=1)
    old_min, ops._X3_MIN_TILES = ops._X3_MIN_TILES, 1
    try:
        y = ops.k_conv_fwd_aff(a.cuda(), s_.cuda(), t_.cuda(), wt.cuda(), geom, scale)
        assert 'conv_x3_fwd_kernel' in launched(ops)
        assert_close(y.cpu(), conv, TOL, 'x3 affine-on-load forward vs float64')
        # the whole layer: + noise, bias, LeakyReLU, InstanceNorm statistics
        L = _lib.lib()
        chunks = L.ganlab_conv_fwd_aff_tail_x3_chunks(geom.ref())
        assert chunks == (h // 16) * (w // 16) * 4
        yt = torch.empty(n, co, h, w, device='cuda')
        mean, rstd = torch.empty(n, co, device='cuda'), torch.empty(n, co, device='cuda')
        ws = torch.empty(n * co * chunks * 2, dtype=torch.float64, device='cuda')
        wp = ops._packed_x3(wt.cuda(), ops.PACK_FWD, scale)
        p = lambda v: ctypes.c_void_p(v.data_ptr()) if v is not None else None
        ac, sc, tc, bc = a.cuda(), s_.cuda(), t_.cuda(), b.cuda()
        nzc, nwc = (nz.cuda(), nw.cuda()) if noise else (None, None)
        _lib.check(L.ganlab_conv_fwd_aff_tail_x3(p(ac), p(wp), p(sc), p(tc), p(bc), p(nzc), p(nwc), p(yt), p(mean), p(rstd),
                                                 geom.ref(), 0.7, ops.ACT_LRELU, 0.2, eps, p(ws), ws.numel() * 8, None), 'tail')
    finally:
        ops._X3_MIN_TILES = old_min
    pre = conv + 0.7 * b.double().view(1, -1, 1, 1)
    if noise:
        pre = pre + nw.double().view(1, -1, 1, 1) * nz.double()
    yd = F.leaky_relu(pre, 0.2)
    assert_close(yt.cpu(), yd, TOL, 'x3 layer tail vs float64')
    md = yd.mean(dim=(2, 3))
    rd = 1.0 / torch.sqrt(yd.var(dim=(2, 3), unbiased=False) + eps)
    assert_close(mean.cpu(), md, TOL, 'x3 layer tail: mean')
    assert_close(rstd.cpu(), rd, TOL, 'x3 layer tail: rstd')


def test_x3_padding_stays_zero_under_the_affine(ops):
    """b = a*s + t is zero-padded AFTER the affine: a constant shift t must not leak into the border taps."""
    n, ci, co, h, w = 1, 64, 64, 16, 16
    a = torch.zeros(n, ci, h, w)
    s_, t_ = torch.ones(n, ci), torch.ones(n, ci)
    wt = torch.ones(co, ci, 3, 3)
    geom = ops.Geom(n, ci, h, w, co, 3, 1)
    old_min, ops._X3_MIN_TILES = ops._X3_MIN_TILES, 1
    try:
        y = ops.k_conv_fwd_aff(a.cuda(), s_.cuda(), t_.cuda(), wt.cuda(), geom, 1.0).cpu()
    finally:
        ops._X3_MIN_TILES = old_min
    assert y[0, 0, 8, 8].item() == 9 * ci and y[0, 0, 0, 0].item() == 4 * ci and y[0, 0, 0, 5].item() == 6 * ci


def test_x3_batched_repack_equals_single_pack(ops):
    """ganlab_pack_many's X3 kind writes the bits ganlab_conv_x3_pack writes (both call gl_x3_pack_position)."""
    from gan_lab_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(3)
    ws = [torch.randn(128, 64, 3, 3, generator=g).cuda(), torch.randn(64, 192, 3, 3, generator=g).cuda()]
    singles, outs = [], []
    arr = (_lib.PackDesc * 4)()
    blocks, i = 0, 0
    for wt in ws:
        for mode in (ops.PACK_FWD, ops.PACK_DGRAD):
            co, ci = wt.shape[0], wt.shape[1]
            n = L.ganlab_conv_x3_pack(None, None, co, ci, mode, 0.37, None)
            assert n == 27 * co * ci
            one = torch.zeros(n, dtype=torch.bfloat16, device='cuda')
            assert L.ganlab_conv_x3_pack(wt.data_ptr(), one.data_ptr(), co, ci, mode, 0.37, None) == n
            singles.append(one)
            out = torch.zeros(n, dtype=torch.bfloat16, device='cuda')
            outs.append(out)
            d = arr[i]
            d.src, d.dst, d.kind, d.Cout, d.Cin, d.ks, d.mode, d.up, d.scale, d.total, d.block0 = \
                wt.data_ptr(), out.data_ptr(), 3, co, ci, 3, mode, 0, 0.37, n, blocks
            blocks += (co * ci + 255) // 256
            i += 1
    tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
    _lib.check(L.ganlab_pack_many(tab.data_ptr(), 4, blocks, None), 'pack_many')
    torch.cuda.synchronize()
    for one, out in zip(singles, outs):
        assert torch.equal(one.view(torch.int16), out.view(torch.int16))
    # the three planes sum to the scaled weight exactly: h + m + l == fl(w * scale)
    wt = ws[0]
    pk = singles[0].float().view(2, 18, 3, 4, 64, 8)     # [co tile][k-step][plane][k-group][co][8]
    total = pk.sum(dim=2)                                # exact in fp32: three bf16 values within 2^-24 of each other's ulp
    # k-step 0 of chunk 0: lane groups 0,1 = tap 0, channels 0..15; groups 2,3 = tap 1
    ref0 = (wt[:64, :16, 0, 0] * 0.37).t().reshape(2, 8, 64).permute(0, 2, 1)       # [g][co][j]
    assert torch.equal(total[0, 0, 0:2], ref0)
    ref1 = (wt[:64, :16, 0, 1] * 0.37).t().reshape(2, 8, 64).permute(0, 2, 1)
    assert torch.equal(total[0, 0, 2:4], ref1)


def test_x3_switch_restores_exact_path(ops):
    geom = ops.Geom(32, 256, 64, 64, 256, 3, 1)
    assert ops.x3_ok(geom) and ops.x3_ok(geom, True)
    prev = ops.set_x3(False)
    try:
        assert not ops.x3_ok(geom)
    finally:
        ops.set_x3(prev)
    assert not ops.x3_ok(ops.Geom(32, 32, 512, 512, 32, 3, 1))          # thin layers stay on the exact kernels
    assert not ops.x3_ok(ops.Geom(1, 512, 16, 16, 512, 3, 1))           # a launch that would leave the chip idle does too


# ---- the transposed stride-2 form: an up layer's forward (+ affine on load), a pooled layer's input gradient -------------------
# N, Cin, Cout, Hl, Wl; the last two: output channels not a multiple of 64 - the 32-channel form (both row parities per workgroup)
S2_CASES = [(2, 64, 64, 16, 16), (3, 128, 64, 32, 16), (2, 256, 128, 16, 32), (5, 64, 128, 32, 32), (2, 64, 32, 16, 16),
            (3, 128, 96, 8, 32)]


@pytest.mark.parametrize('case', S2_CASES)
def test_x3_up_layer_forward(ops, case):
    n, ci, co, hl, wl = case
    g = torch.Generator().manual_seed(21 + ci + n)
    x = torch.randn(n, ci, hl, wl, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g)
    b = torch.randn(co, generator=g)
    s_ = torch.rand(n, ci, generator=g) + 0.5
    t_ = torch.randn(n, ci, generator=g)
    geom = ops.Geom(n, ci, hl, wl, co, 3, 1, up=1)
    scale = 1.0 / (3 * ci ** 0.5)
    assert geom.s2
    old_min, ops._X3_MIN_TILES = ops._X3_MIN_TILES, 1
    try:
        assert ops.x3_s2_ok(geom)
        y3 = ops.k_conv_fwd(x.cuda(), wt.cuda(), b.cuda(), geom, scale, 0.5, ops.ACT_LRELU, 0.2)
        assert 'conv_x3_up_kernel' in launched(ops)
        ya = ops.k_conv_fwd_aff(x.cuda(), s_.cuda(), t_.cuda(), wt.cuda(), geom, scale)
        assert 'conv_x3_up_kernel' in launched(ops)
        prev = ops.set_x3(False)
        try:
            y1 = ops.k_conv_fwd(x.cuda(), wt.cuda(), b.cuda(), geom, scale, 0.5, ops.ACT_LRELU, 0.2)
            assert 'conv_x3_up_kernel' not in launched(ops)
        finally:
            ops.set_x3(prev)
    finally:
        ops._X3_MIN_TILES = old_min
    up = F.interpolate(x.double(), scale_factor=2, mode='nearest')
    yd = F.leaky_relu(F.conv2d(up, wt.double() * scale, None, padding=1) + 0.5 * b.double().view(1, -1, 1, 1), 0.2)
    bd = x.double() * s_.double().view(n, ci, 1, 1) + t_.double().view(n, ci, 1, 1)
    yad = F.conv2d(F.interpolate(bd, scale_factor=2, mode='nearest'), wt.double() * scale, None, padding=1)
    assert_close(y3.cpu(), yd, TOL, 'x3 up layer vs float64')
    assert_close(y3.cpu(), y1.cpu(), TOL, 'x3 up layer vs exact-fp32 kernel')
    assert_close(ya.cpu(), yad, TOL, 'x3 up layer, affine on load vs float64')
    assert rms_rel(y3, yd) <= 1.1 * rms_rel(y1, yd), (rms_rel(y3, yd), rms_rel(y1, yd))


@pytest.mark.parametrize('case', S2_CASES)
def test_x3_pooled_layer_input_gradient(ops, case):
    n, co_, ci_, hl, wl = case          # the pooled layer: ci_ -> co_ channels, input 2hl x 2wl
    cin, cout = ci_, co_
    g = torch.Generator().manual_seed(31 + cin + n)
    wt = torch.randn(cout, cin, 3, 3, generator=g)
    gy = torch.randn(n, cout, hl, wl, generator=g)
    geom = ops.Geom(n, cin, 2 * hl, 2 * wl, cout, 3, 1, pool=1)
    scale = 1.0 / (3 * cin ** 0.5)
    assert geom.s2
    old_min, ops._X3_MIN_TILES = ops._X3_MIN_TILES, 1
    try:
        assert ops.x3_s2_ok(geom, True)
        gx3 = ops.k_conv_dgrad(gy.cuda(), wt.cuda(), geom, scale)
        assert 'conv_x3_up_kernel' in launched(ops)
        prev = ops.set_x3(False)
        try:
            gx1 = ops.k_conv_dgrad(gy.cuda(), wt.cuda(), geom, scale)
            assert 'conv_x3_up_kernel' not in launched(ops)
        finally:
            ops.set_x3(prev)
    finally:
        ops._X3_MIN_TILES = old_min
    xd = torch.zeros(n, cin, 2 * hl, 2 * wl, dtype=torch.float64, requires_grad=True)
    gxd, = torch.autograd.grad(F.avg_pool2d(F.conv2d(xd, wt.double() * scale, padding=1), 2), xd, gy.double())
    assert_close(gx3.cpu(), gxd, TOL, 'x3 pooled layer input gradient vs float64')
    assert_close(gx3.cpu(), gx1.cpu(), TOL, 'x3 pooled layer input gradient vs exact-fp32 kernel')
    assert rms_rel(gx3, gxd) <= 1.1 * rms_rel(gx1, gxd), (rms_rel(gx3, gxd), rms_rel(gx1, gxd))


def test_x3_s2_batched_repack_equals_single_pack(ops):
    from gan_lab_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(4)
    jobs = [(128, 64, 1), (128, 64, 0), (32, 64, 1), (64, 32, 0)]       # Cout, Cin, up (the last two: the 32-channel layout)
    arr = (_lib.PackDesc * len(jobs))()
    singles, outs, keep, blocks = [], [], [], 0
    for i, (co, ci, up) in enumerate(jobs):
        wt = torch.randn(co, ci, 3, 3, generator=g).cuda()
        keep.append(wt)
        n = L.ganlab_conv_s2_x3_pack(None, None, co, ci, up, 0.41, None)
        assert n == 48 * co * ci
        one = torch.zeros(n, dtype=torch.bfloat16, device='cuda')
        assert L.ganlab_conv_s2_x3_pack(wt.data_ptr(), one.data_ptr(), co, ci, up, 0.41, None) == n
        out = torch.zeros(n, dtype=torch.bfloat16, device='cuda')
        singles.append(one)
        outs.append(out)
        d = arr[i]
        d.src, d.dst, d.kind, d.Cout, d.Cin, d.ks, d.mode, d.up, d.scale, d.total, d.block0 = \
            wt.data_ptr(), out.data_ptr(), 3, co, ci, 4, 0, up, 0.41, n, blocks
        blocks += (co * ci + 255) // 256
    tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
    _lib.check(L.ganlab_pack_many(tab.data_ptr(), len(jobs), blocks, None), 'pack_many')
    torch.cuda.synchronize()
    for one, out in zip(singles, outs):
        assert torch.equal(one.view(torch.int16), out.view(torch.int16))
        assert int((one.view(torch.int16) != 0).sum()) > one.numel() // 2       # (every position written)


# ---- weight gradient -------------------------------------------------------------------------------------------------------------
WG_CASES = [(2, 64, 64, 32, 32), (3, 128, 64, 16, 64), (2, 64, 96, 64, 32), (5, 192, 128, 8, 32), (9, 64, 64, 4, 96)]   # N, Cin, Cout, H, W


@pytest.mark.parametrize('case', WG_CASES)
@pytest.mark.parametrize('aff', [False, True])
def test_x3_weight_gradient(ops, case, aff):
    n, ci, co, h, w = case
    g = torch.Generator().manual_seed(41 + ci + n + h)
    x = torch.randn(n, ci, h, w, generator=g)
    gy = torch.randn(n, co, h, w, generator=g)
    s_ = torch.rand(n, ci, generator=g) + 0.5
    t_ = torch.randn(n, ci, generator=g)
    geom = ops.Geom(n, ci, h, w, co, 3, 1)
    assert ops.x3_wgrad_ok(geom)
    scale = 0.013
    run = (lambda: ops.k_conv_wgrad_aff(gy.cuda(), x.cuda(), s_.cuda(), t_.cuda(), geom, scale)) if aff else \
        (lambda: ops.k_conv_wgrad(gy.cuda(), x.cuda(), geom, scale))
    gw3 = run()
    assert 'x3w_reduce_kernel' in launched(ops)
    prev = ops.set_x3(False)
    try:      # (the exact-fp32 affine-on-load weight gradient does not take every geometry the split-product one does)
        gw1 = run() if (not aff or ops.conv_aff_ok((n, ci, h, w), torch.empty(co, ci, 3, 3))) else gw3
        assert 'x3w_reduce_kernel' not in launched(ops) or gw1 is gw3
    finally:
        ops.set_x3(prev)
    xin = x.double() * s_.double().view(n, ci, 1, 1) + t_.double().view(n, ci, 1, 1) if aff else x.double()
    wd = torch.zeros(co, ci, 3, 3, dtype=torch.float64, requires_grad=True)
    gwd, = torch.autograd.grad(F.conv2d(xin, wd, padding=1), wd, gy.double())
    gwd = gwd * scale
    assert_close(gw3.cpu(), gwd, TOL, 'x3 weight gradient vs float64')
    assert_close(gw3.cpu(), gw1.cpu(), TOL, 'x3 weight gradient vs exact-fp32 kernel')
    # (at these sizes - a few thousand terms per weight - both kernels sit at the rounding of the result itself; the bar is the
    # exact kernels' error or ATen's 2.2e-7 level of tools/op_error_probe.py, whichever is larger)
    assert rms_rel(gw3, gwd) <= max(1.1 * rms_rel(gw1, gwd), 2.5e-7), (rms_rel(gw3, gwd), rms_rel(gw1, gwd))


def test_x3_weight_gradient_is_deterministic(ops):
    n, ci, co, h, w = 4, 64, 64, 32, 64
    g = torch.Generator().manual_seed(9)
    x, gy = torch.randn(n, ci, h, w, generator=g).cuda(), torch.randn(n, co, h, w, generator=g).cuda()
    geom = ops.Geom(n, ci, h, w, co, 3, 1)
    a = ops.k_conv_wgrad(gy, x, geom, 1.0).clone()
    b = ops.k_conv_wgrad(gy, x, geom, 1.0)
    assert torch.equal(a, b)


# ---- the strided stride-2 form: a pooled layer's forward, an up layer's input gradient -----------------------------------------------
SD_CASES = [(2, 64, 128, 16, 16), (3, 128, 256, 8, 32), (5, 32, 128, 24, 16), (2, 256, 128, 16, 48)]     # N, Cin, Cout, Hl, Wl


@pytest.mark.parametrize('case', SD_CASES)
def test_x3_pooled_layer_forward(ops, case):
    n, ci, co, hl, wl = case
    g = torch.Generator().manual_seed(51 + ci + n)
    x = torch.randn(n, ci, 2 * hl, 2 * wl, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g)
    b = torch.randn(co, generator=g)
    geom = ops.Geom(n, ci, 2 * hl, 2 * wl, co, 3, 1, pool=1)
    scale = 1.0 / (3 * ci ** 0.5)
    old_min, ops._X3_MIN_TILES = ops._X3_MIN_TILES, 1
    try:
        assert ops.x3_s2_down_ok(geom)
        y3 = ops.k_conv_fwd(x.cuda(), wt.cuda(), b.cuda(), geom, scale, 0.5, ops.ACT_LRELU, 0.2)
        assert 'conv_x3_down_kernel' in launched(ops)
        prev = ops.set_x3(False)
        try:
            y1 = ops.k_conv_fwd(x.cuda(), wt.cuda(), b.cuda(), geom, scale, 0.5, ops.ACT_LRELU, 0.2)
            assert 'conv_x3_down_kernel' not in launched(ops)
        finally:
            ops.set_x3(prev)
    finally:
        ops._X3_MIN_TILES = old_min
    yd = F.leaky_relu(F.avg_pool2d(F.conv2d(x.double(), wt.double() * scale, None, padding=1), 2) + 0.5 * b.double().view(1, -1, 1, 1), 0.2)
    assert_close(y3.cpu(), yd, TOL, 'x3 pooled layer forward vs float64')
    assert_close(y3.cpu(), y1.cpu(), TOL, 'x3 pooled layer forward vs exact-fp32 kernel')
    assert rms_rel(y3, yd) <= 1.1 * rms_rel(y1, yd), (rms_rel(y3, yd), rms_rel(y1, yd))


@pytest.mark.parametrize('case', SD_CASES)
def test_x3_up_layer_input_gradient(ops, case):
    n, co_, ci_, hl, wl = case          # the up layer: ci_ -> co_ channels (GEMM contracts co_), input hl x wl
    cin, cout = ci_, co_
    g = torch.Generator().manual_seed(61 + cin + n)
    wt = torch.randn(cout, cin, 3, 3, generator=g)
    gy = torch.randn(n, cout, 2 * hl, 2 * wl, generator=g)
    geom = ops.Geom(n, cin, hl, wl, cout, 3, 1, up=1)
    scale = 1.0 / (3 * cin ** 0.5)
    old_min, ops._X3_MIN_TILES = ops._X3_MIN_TILES, 1
    try:
        assert ops.x3_s2_down_ok(geom, True)
        gx3 = ops.k_conv_dgrad(gy.cuda(), wt.cuda(), geom, scale)
        assert 'conv_x3_down_kernel' in launched(ops)
        prev = ops.set_x3(False)
        try:
            gx1 = ops.k_conv_dgrad(gy.cuda(), wt.cuda(), geom, scale)
        finally:
            ops.set_x3(prev)
    finally:
        ops._X3_MIN_TILES = old_min
    xd = torch.zeros(n, cin, hl, wl, dtype=torch.float64, requires_grad=True)
    gxd, = torch.autograd.grad(F.conv2d(F.interpolate(xd, scale_factor=2, mode='nearest'), wt.double() * scale, padding=1), xd, gy.double())
    assert_close(gx3.cpu(), gxd, TOL, 'x3 up layer input gradient vs float64')
    assert_close(gx3.cpu(), gx1.cpu(), TOL, 'x3 up layer input gradient vs exact-fp32 kernel')
    assert rms_rel(gx3, gxd) <= 1.1 * rms_rel(gx1, gxd), (rms_rel(gx3, gxd), rms_rel(gx1, gxd))


# ---- stride-2 weight gradient: nine taps over the 2x2 box sums of the high-resolution operand ----------------------------------------
# N, Cin, Cout, Hl, Wl, kind: several strips per image and per workgroup, odd batch sizes (ragged last split), both channel roles
SW_CASES = [(2, 32, 64, 8, 32, 'pool'), (3, 64, 128, 16, 64, 'pool'), (5, 96, 64, 4, 32, 'pool'), (2, 32, 192, 32, 96, 'pool'),
            (2, 64, 32, 8, 32, 'up'), (3, 128, 64, 16, 64, 'up'), (5, 64, 96, 4, 32, 'up'), (7, 192, 32, 8, 64, 'up'),
            (3, 64, 64, 8, 16, 'pool'), (2, 64, 64, 16, 16, 'up')]      # 16-pixel-wide low maps: one half-empty strip


@pytest.mark.parametrize('case', SW_CASES)
@pytest.mark.parametrize('aff', [False, True])
def test_x3_stride2_weight_gradient(ops, case, aff):
    n, ci, co, hl, wl, kind = case
    up = kind == 'up'
    if aff and not up:
        pytest.skip('the affine-on-load operand exists for up layers only (the discriminator has no InstanceNorm)')
    g = torch.Generator().manual_seed(71 + ci + n + hl)
    hi, wi = (hl, wl) if up else (2 * hl, 2 * wl)
    x = torch.randn(n, ci, hi, wi, generator=g)
    gy = torch.randn(n, co, 2 * hl, 2 * wl, generator=g) if up else torch.randn(n, co, hl, wl, generator=g)
    s_ = torch.rand(n, ci, generator=g) + 0.5
    t_ = torch.randn(n, ci, generator=g)
    geom = ops.Geom(n, ci, hi, wi, co, 3, 1, up=1) if up else ops.Geom(n, ci, hi, wi, co, 3, 1, pool=1)
    assert ops.x3_s2_wgrad_ok(geom)
    scale = 0.017
    run = (lambda: ops.k_conv_wgrad_aff(gy.cuda(), x.cuda(), s_.cuda(), t_.cuda(), geom, scale)) if aff else \
        (lambda: ops.k_conv_wgrad(gy.cuda(), x.cuda(), geom, scale))
    gw3 = run()
    assert 'x3sw_reduce_kernel' in launched(ops)
    from gan_lab_amd._lib import GanlabLibraryError
    prev = ops.set_x3(False)
    try:      # (the exact-fp32 affine-on-load weight gradient does not take 16-pixel-wide maps; the step never asks it to)
        try:
            gw1 = run()
            assert 'x3sw_reduce_kernel' not in launched(ops)
        except GanlabLibraryError:
            assert aff and wl == 16
            gw1 = gw3
    finally:
        ops.set_x3(prev)
    xin = x.double() * s_.double().view(n, ci, 1, 1) + t_.double().view(n, ci, 1, 1) if aff else x.double()
    wd = torch.zeros(co, ci, 3, 3, dtype=torch.float64, requires_grad=True)
    if up:
        yd = F.conv2d(F.interpolate(xin, scale_factor=2, mode='nearest'), wd, padding=1)
    else:
        yd = F.avg_pool2d(F.conv2d(xin, wd, padding=1), 2)
    gwd, = torch.autograd.grad(yd, wd, gy.double())
    gwd = gwd * scale
    assert_close(gw3.cpu(), gwd, TOL, 'x3 stride-2 weight gradient vs float64')
    assert_close(gw3.cpu(), gw1.cpu(), TOL, 'x3 stride-2 weight gradient vs exact-fp32 kernel')
    assert rms_rel(gw3, gwd) <= max(1.1 * rms_rel(gw1, gwd), 2.5e-7), (rms_rel(gw3, gwd), rms_rel(gw1, gwd))


def test_x3_stride2_weight_gradient_is_deterministic_and_closes_long_chains(ops):
    """More than 32 k-steps per workgroup: the hi*hi chains are closed into the workspace and read back (twice here)."""
    n, ci, co, hl, wl = 2, 32, 64, 64, 32
    g = torch.Generator().manual_seed(19)
    x, gy = torch.randn(n, ci, 2 * hl, 2 * wl, generator=g), torch.randn(n, co, hl, wl, generator=g)
    geom = ops.Geom(n, ci, 2 * hl, 2 * wl, co, 3, 1, pool=1)
    assert ops.x3_s2_wgrad_ok(geom)
    a = ops.k_conv_wgrad(gy.cuda(), x.cuda(), geom, 1.0).clone()
    b = ops.k_conv_wgrad(gy.cuda(), x.cuda(), geom, 1.0)
    assert torch.equal(a, b)
    wd = torch.zeros(co, ci, 3, 3, dtype=torch.float64, requires_grad=True)
    gwd, = torch.autograd.grad(F.avg_pool2d(F.conv2d(x.double(), wd, padding=1), 2), wd, gy.double())
    assert_close(a.cpu(), gwd, TOL, 'x3 stride-2 weight gradient (long chains) vs float64')
    assert rms_rel(a, gwd) <= 2.5e-7, rms_rel(a, gwd)


@pytest.mark.parametrize('kind', ['pool', 'up'])
def test_x3_stride2_weight_gradient_full_size_properties(ops, kind):
    """Batch 32 at a benchmark layer (64 <-> 128 channels, 256^2 <-> 128^2): agreement with the exact kernel, linearity in the
    output gradient, and the affine-on-load operand against the materialised one (up)."""
    up = kind == 'up'
    n, hl = 32, 128
    ci, co = (128, 64) if up else (64, 128)
    hi = hl if up else 2 * hl
    g = torch.Generator(device='cuda').manual_seed(5)
    x = torch.randn(n, ci, hi, hi, device='cuda', generator=g)
    gshape = (n, co, 2 * hl, 2 * hl) if up else (n, co, hl, hl)
    g1, g2 = torch.randn(*gshape, device='cuda', generator=g), torch.randn(*gshape, device='cuda', generator=g)
    geom = ops.Geom(n, ci, hi, hi, co, 3, 1, up=1) if up else ops.Geom(n, ci, hi, hi, co, 3, 1, pool=1)
    assert ops.x3_s2_wgrad_ok(geom)
    w1 = ops.k_conv_wgrad(g1, x, geom, 0.02).clone()
    assert 'x3sw_reduce_kernel' in launched(ops)
    w2 = ops.k_conv_wgrad(g2, x, geom, 0.02).clone()
    w12 = ops.k_conv_wgrad(g1 + g2, x, geom, 0.02).clone()
    assert_close(w12.cpu(), (w1 + w2).cpu(), 1e-5, 'linearity in the output gradient')
    prev = ops.set_x3(False)
    try:
        e1 = ops.k_conv_wgrad(g1, x, geom, 0.02).clone()
        assert 'x3sw_reduce_kernel' not in launched(ops)
    finally:
        ops.set_x3(prev)
    assert_close(w1.cpu(), e1.cpu(), 1e-5, 'split-product against exact-fp32 kernel at batch 32')
    if up:
        s_ = torch.rand(n, ci, device='cuda', generator=g) + 0.5
        t_ = torch.randn(n, ci, device='cuda', generator=g)
        wa = ops.k_conv_wgrad_aff(g1, x, s_, t_, geom, 0.02).clone()
        assert 'x3sw_reduce_kernel' in launched(ops)
        wm = ops.k_conv_wgrad(g1, x * s_.view(n, ci, 1, 1) + t_.view(n, ci, 1, 1), geom, 0.02)
        assert_close(wa.cpu(), wm.cpu(), 1e-5, 'affine on load against the materialised operand')
    print(f'SUMMARY x3 stride-2 weight gradient {kind} x32: linearity, exact-kernel agreement' + (', affine form' if up else '') + ' ok')


def test_x3_stride2_weight_gradient_several_strips_per_workgroup(ops):
    """128 channel-tile pairs leave 4 k-splits for 5 strips: workgroups walk 2, 2 and 1 strips (= images here) - the load cursor's
    strip change, the affine's reload per image and the ragged last split."""
    n, ci, co, hl, wl = 5, 512, 512, 4, 32
    g = torch.Generator(device='cuda').manual_seed(23)
    x = torch.randn(n, ci, hl, wl, device='cuda', generator=g)
    gy = torch.randn(n, co, 2 * hl, 2 * wl, device='cuda', generator=g)
    s_ = torch.rand(n, ci, device='cuda', generator=g) + 0.5
    t_ = torch.randn(n, ci, device='cuda', generator=g)
    geom = ops.Geom(n, ci, hl, wl, co, 3, 1, up=1)
    assert ops.x3_s2_wgrad_ok(geom)
    wa = ops.k_conv_wgrad_aff(gy, x, s_, t_, geom, 0.03).clone()
    assert 'x3sw_reduce_kernel' in launched(ops)
    xm = x * s_.view(n, ci, 1, 1) + t_.view(n, ci, 1, 1)
    wm = ops.k_conv_wgrad(gy, xm, geom, 0.03).clone()
    prev = ops.set_x3(False)
    try:
        we = ops.k_conv_wgrad(gy, xm, geom, 0.03).clone()
        assert 'x3sw_reduce_kernel' not in launched(ops)
    finally:
        ops.set_x3(prev)
    assert_close(wa.cpu(), wm.cpu(), 1e-5, 'affine on load against the materialised operand')
    assert_close(wm.cpu(), we.cpu(), 1e-5, 'split-product against exact-fp32 kernel')
    # and the pooled form with the same tiling
    xp = torch.randn(n, ci, 2 * hl, 2 * wl, device='cuda', generator=g)
    gp = torch.randn(n, co, hl, wl, device='cuda', generator=g)
    geom_p = ops.Geom(n, ci, 2 * hl, 2 * wl, co, 3, 1, pool=1)
    w3 = ops.k_conv_wgrad(gp, xp, geom_p, 0.03).clone()
    assert 'x3sw_reduce_kernel' in launched(ops)
    prev = ops.set_x3(False)
    try:
        w1 = ops.k_conv_wgrad(gp, xp, geom_p, 0.03).clone()
    finally:
        ops.set_x3(prev)
    assert_close(w3.cpu(), w1.cpu(), 1e-5, 'pooled form, several strips per workgroup')
