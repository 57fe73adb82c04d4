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
    gw_ref = torch.nn.grad.conv2d_weight(bf(xi), wt.shape, bf(gz32), padding=1) * scale
    assert_close(wg.grad.cpu(), gw_ref, 5e-5, 'bf16 wgrad vs bf16-operand float64')
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
