"""Layer ops of the G/D conv stack on the HIP kernels (drop-in for gan_lab/utils/custom_layers.py).

Same class names, constructor arguments and ``state_dict`` keys as the reference
(custom_layers.py:18-306), but every ``forward`` lands in ``gan_lab_amd.ops`` (hand-written gfx950
kernels): nothing here computes with ATen.  ``fused_sequential`` is the peephole executor the
architectures use so that e.g. ``Sequential(Upsample, Conv2dEx, blur)`` or
``Sequential(Conv2dEx(bias), LeakyReLU)`` run as ONE kernel while the module tree (and therefore
the checkpoint layout) stays that of the reference.
"""
import os

import torch
from torch import nn

from .. import ops
from .initializer import Initializer


class Lambda(nn.Module):
    """Converts any function into a Module (custom_layers.py:18-29)."""

    def __init__(self, func, **kwargs):
        super().__init__()
        self.func = func
        self.kwargs = kwargs if kwargs else {}

    def forward(self, x):
        return self.func(x, **self.kwargs)


# -- activations / resampling as parameter-free modules ------------------------------------------- #
class LeakyReLU(nn.Module):
    """nn.LeakyReLU stand-in (negative_slope=0 gives ReLU)."""

    def __init__(self, negative_slope=0.2):
        super().__init__()
        self.negative_slope = float(negative_slope)

    def forward(self, x):
        return ops.bias_act(x, act='lrelu', slope=self.negative_slope)

    def extra_repr(self):
        return f'negative_slope={self.negative_slope}'


class Tanh(nn.Module):
    """nn.Tanh stand-in: ``--nonlinearity tanh`` as the hidden activation (config.py:208,254, resnetgan/learner.py:180-181).
    Off the benchmark configurations: it runs as a pass of its own (no fold into the conv / normalisation kernels), first
    and second order."""

    def forward(self, x):
        return ops.tanh(x)


def own_nl(nl, default_slope=0.2):
    """The reference's nn.ReLU() / nn.LeakyReLU() / nn.Tanh() instances (or ``None``) as HIP-path modules."""
    if nl is None:
        return LeakyReLU(default_slope)
    if isinstance(nl, (LeakyReLU, Tanh)):
        return nl
    if isinstance(nl, nn.ReLU):
        return LeakyReLU(0.)
    if isinstance(nl, nn.LeakyReLU):
        return LeakyReLU(nl.negative_slope)
    if isinstance(nl, nn.Tanh):
        return Tanh()
    raise NotImplementedError(f'nonlinearity {nl!r} has no HIP kernel (ReLU / LeakyReLU / Tanh)')


class Upsample2x(nn.Module):
    """nn.Upsample(scale_factor=2, mode='nearest') stand-in (fused into the following conv)."""

    def forward(self, x):
        return ops.upsample2(x)


class AvgPool2x(nn.Module):
    """nn.AvgPool2d(kernel_size=2, stride=2) stand-in."""

    def forward(self, x):
        return ops.avg_pool2(x)


class BilinearUpsample2x(nn.Module):
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=...) stand-in (resnetgan/learner.py:147-158): a
    table-driven streaming kernel in front of a plain 3x3 conv (no fold into the stride-2 kernels)."""

    def __init__(self, align_corners=False):
        super().__init__()
        self.align_corners = bool(align_corners)

    def forward(self, x):
        return ops.resample(x, 'bilinear_up', self.align_corners)

    def extra_repr(self):
        return f'scale_factor=2.0, mode=bilinear, align_corners={self.align_corners}'


class NearestPool2x(nn.Module):
    """NearestPool2d stand-in (custom_layers.py:59-65): F.interpolate(scale_factor=.5, mode='nearest') = x[..., ::2, ::2]."""

    def forward(self, x):
        if x.dim() == 3:
            x = x.view(-1, *x.shape)
        return ops.resample(x, 'nearest_down')


class BilinearPool2x(nn.Module):
    """BilinearPool2d stand-in (custom_layers.py:67-75).  Without align_corners the 0.5x bilinear samples fall on the
    centres of the 2x2 cells - the 2x2 average - and the layer takes the average pool's kernels and folds
    (``is_avg_pool``); with align_corners it is a two-tap gather of its own."""

    def __init__(self, align_corners=False):
        super().__init__()
        self.align_corners = bool(align_corners)

    def forward(self, x):
        if x.dim() == 3:
            x = x.view(-1, *x.shape)
        return ops.resample(x, 'bilinear_down', True) if self.align_corners else ops.avg_pool2(x)

    def extra_repr(self):
        return f'align_corners={self.align_corners}'


def is_avg_pool(m):
    """Is ``m`` a 2x2 average (the stride-2 conv kernels fold it)?"""
    return isinstance(m, AvgPool2x) or (isinstance(m, BilinearPool2x) and not m.align_corners)


def make_upsampler(kind, align_corners=False):
    """config.model_upsample_type -> module (resnetgan/learner.py:147-158)."""
    kind = kind.casefold()
    if kind == 'nearest':
        return Upsample2x()
    if kind == 'bilinear':
        return BilinearUpsample2x(align_corners)
    raise ValueError("config does not support this model_upsample_type.\n"
                     "Supported Upsampling Types are: [ 'nearest', 'bilinear' ]")


def make_downsampler(kind, align_corners=False):
    """config.model_downsample_type -> module (resnetgan/learner.py:160-173)."""
    kind = kind.casefold()
    if kind in ('average', 'box',):
        return AvgPool2x()
    if kind == 'nearest':
        return NearestPool2x()
    if kind == 'bilinear':
        return BilinearPool2x(align_corners)
    raise ValueError("config does not support this model_downsample_type.\n"
                     "Supported Downsampling Types are: [ 'nearest', 'average', 'box', 'bilinear' ]")


def own_resampler(m):
    """A torch / reference resampler module -> this package's (None and own modules pass through)."""
    if m is None or isinstance(m, (Upsample2x, BilinearUpsample2x, AvgPool2x, NearestPool2x, BilinearPool2x)):
        return m
    if isinstance(m, nn.Upsample) and float(m.scale_factor) == 2.:
        if m.mode == 'nearest':
            return Upsample2x()
        if m.mode == 'bilinear':
            return BilinearUpsample2x(bool(m.align_corners))
    if isinstance(m, nn.AvgPool2d) and m.kernel_size in (2, (2, 2)) and m.stride in (2, (2, 2)):
        return AvgPool2x()
    name = type(m).__name__
    if name == 'NearestPool2d':
        return NearestPool2x()
    if name == 'BilinearPool2d':
        return BilinearPool2x(bool(getattr(m, 'align_corners', False)))
    raise NotImplementedError(f'resampler {m!r} has no HIP kernel (nearest / bilinear x2, 2x2 average, nearest / bilinear x0.5)')


class Blur2d(nn.Module):
    """Depthwise 3x3 binomial blur, zero padding (custom_layers.py:41-51)."""

    def forward(self, x):
        return ops.blur(x)


def get_blur_op(blur_type, num_channels):
    """Low-pass filter op (custom_layers.py:36-53).  Only 'binomial' is on the hot path."""
    if blur_type.casefold() == 'binomial':
        return Blur2d()
    if blur_type.casefold() == 'gaussian':
        raise NotImplementedError('Gaussian blur not yet implemented.')
    raise NotImplementedError(f"blur_type '{blur_type}' has no HIP kernel (only 'binomial').")


# -- normalisation --------------------------------------------------------------------------------- #
class PixelNorm2d(nn.Module):
    def forward(self, x, eps=1.e-8):
        return ops.pixelnorm(x, eps)


class InstanceNorm2d(nn.Module):
    """nn.InstanceNorm2d(None, eps=1e-8): biased var, no affine, no running stats."""

    def __init__(self, eps=1.e-8):
        super().__init__()
        self.eps = eps

    def forward(self, x):
        return ops.instnorm_style(x, None, self.eps)


class BatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d parameter / buffer container (same state_dict keys) whose forward runs on the HIP
    kernels (ops.batch_norm: channel sums + per-channel affines)."""

    def forward(self, x, act_slope=None):
        count = self.training and self.track_running_stats and self.num_batches_tracked is not None
        use_batch = self.training or not self.track_running_stats
        track = self.training and self.track_running_stats
        return ops.batch_norm(x, self.weight, self.bias, self.running_mean if track or not use_batch else None,
                              self.running_var if track or not use_batch else None, use_batch,
                              momentum=self.momentum, eps=self.eps, batches=self.num_batches_tracked if count else None,
                              act_slope=act_slope)


class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm([C, R, R]) parameter container whose forward (and first / second derivatives, for
    WGAN-GP) runs on the HIP kernels (ops.layer_norm)."""

    def forward(self, x, act_slope=None):
        assert tuple(x.shape[1:]) == tuple(self.normalized_shape), (x.shape, self.normalized_shape)
        return ops.layer_norm(x, self.weight, self.bias, eps=self.eps, act_slope=act_slope)


class NormalizeLayer(nn.Module):
    """All normalisation methods in one place (custom_layers.py:88-111)."""

    def __init__(self, norm_type, ni=None, res=None):
        super().__init__()
        norm_type = norm_type.lower()
        if norm_type in ('pixelnorm', 'pixel norm',):
            self.norm = PixelNorm2d()
        elif norm_type in ('instancenorm', 'instance norm',):
            self.norm = InstanceNorm2d(eps=1.e-8)
        elif norm_type in ('batchnorm', 'batch norm',):
            assert isinstance(ni, int)
            self.norm = BatchNorm2d(ni)
        elif norm_type in ('layernorm', 'layer norm',):
            assert isinstance(ni, int)
            assert isinstance(res, int)
            self.norm = LayerNorm([ni, res, res])
        else:
            raise Exception(f'`norm_type` == "{norm_type}" not supported.')

    def forward(self, x, act_slope=None):
        """``act_slope``: the LeakyReLU that follows (fused_sequential) - Batch / LayerNorm apply it in their own passes."""
        if act_slope is None:
            return self.norm(x)
        if isinstance(self.norm, (BatchNorm2d, LayerNorm)):
            return self.norm(x, act_slope=act_slope)
        return ops.bias_act(self.norm(x), act='lrelu', slope=act_slope)


# -- minibatch stddev ------------------------------------------------------------------------------ #
def concat_mbstd_layer(x, group_size=4):
    """Minibatch Standard Deviation layer (custom_layers.py:117-140).  The statistic (unbiased var
    over contiguous groups, sqrt(.+1e-8), mean over C*H*W) is a HIP kernel with explicit first and
    second derivatives; expand/cat are pure data movement."""
    b, c, h, w = x.shape
    group_size = min(b, group_size)
    if b % group_size != 0:
        group_size = b
    G = b // group_size
    if group_size > 1:
        stat = ops.mbstd_stat(x, group_size)                       # (G,)
        m = ops.group_broadcast(stat, group_size, h, w)            # (b, 1, h, w): one value per group
    else:
        m = torch.zeros(b, 1, h, w, device=x.device, dtype=x.dtype)
    return torch.cat((x, m), dim=1)


# -- eq-LR conv / linear --------------------------------------------------------------------------- #
def _init_weight(mod_weight, initializer, init, init_type, equalized_lr, lrmul, use_lrmul, stride=1):
    """Weight init + runtime scale, following custom_layers.py:171-195 (including the
    ``init_type == ('default','resnet',)`` str-vs-tuple comparison of Conv2dEx that never matches,
    handled by the callers)."""
    wscale = None
    if init_type in ('progan', 'stylegan',) and init is not None:
        bound = initializer.get_init_bound_layer(tensor=mod_weight, distribution_type='Normal', stride=stride)
        if equalized_lr:
            wscale = bound
            mod_weight.data.normal_(0., 1. / lrmul)
        else:
            mod_weight.data.normal_(0., bound / lrmul)
    elif init_type == 'standard normal' and init is None and not equalized_lr and not use_lrmul:
        mod_weight.data.normal_(0., 1.)
    return wscale


class Conv2dEx(nn.Module):
    def __init__(self, ni, nf, ks, stride=1, padding=0, groups=1, init='he', init_type='default',
                 gain_sq_base=2., equalized_lr=False, lrmul=1., include_bias=True):
        super().__init__()
        if stride != 1 or groups != 1:
            raise NotImplementedError('the HIP conv kernels implement stride 1, groups 1')
        self.ni, self.nf, self.ks, self.padding = ni, nf, ks, padding
        init = init.casefold() if init is not None else None
        init_type = init_type.casefold()
        self.equalized_lr = equalized_lr
        self.use_lrmul = True if lrmul != 1. else False
        self.lrmul = lrmul
        self.initializer = None
        if init_type != 'standard normal':
            self.initializer = Initializer(init=init, init_type=init_type, gain_sq_base=gain_sq_base,
                                           equalized_lr=equalized_lr)
        # parameter container only (keys `conv2d.weight` / `conv2d.bias`); never called
        self.conv2d = nn.Conv2d(ni, nf, kernel_size=ks, stride=stride, padding=padding, groups=groups,
                                bias=include_bias)
        # NB: for init_type 'default'/'resnet' the reference's tuple comparison never matches
        # (custom_layers.py:173), so PyTorch's default init and wscale=None are what it ends up with.
        self.wscale = _init_weight(self.conv2d.weight, self.initializer, init, init_type, equalized_lr, lrmul,
                                   self.use_lrmul, stride)
        self.bias = None
        if include_bias:
            self.conv2d.bias.data.fill_(0)

    @property
    def scale(self):
        s = self.wscale if (self.equalized_lr and self.wscale is not None) else 1.0
        return s * (self.lrmul if self.use_lrmul else 1.0)

    def forward(self, x, up=False, act=None, slope=0.2, pool=False, bias_mod=None, blur=False, defer_act_grad=False,
                in_act_slope=None, in_blur_handoff=None, in_rgb_handoff=None):
        # (conv(x*wscale) + b) * lrmul  ==  scale*conv(x) + b*lrmul   (custom_layers.py:202-211)
        # pool / bias_mod: the D down layer  conv -> AvgPool2d -> Conv2dBias -> LeakyReLU  as one kernel
        bias, bias_scale = self.conv2d.bias, (self.lrmul if self.use_lrmul else 1.0)
        if bias_mod is not None:
            assert bias is None
            bias, bias_scale = bias_mod.bias, (bias_mod.lrmul if bias_mod.use_lrmul else 1.0)
        return ops.conv2d(x, self.conv2d.weight, bias, scale=self.scale, padding=self.padding, up=up,
                          bias_scale=bias_scale, act=act, slope=slope, pool=pool, blur=blur,
                          defer_act_grad=defer_act_grad, in_act_slope=in_act_slope, in_blur_handoff=in_blur_handoff,
                          in_rgb_handoff=in_rgb_handoff)


class Conv2dBias(nn.Module):
    def __init__(self, nf, lrmul=1., device='cpu'):
        super().__init__()
        self.use_lrmul = True if lrmul != 1. else False
        self.lrmul = lrmul
        self.bias = nn.Parameter(torch.zeros(1, nf, 1, 1, device=device))

    def forward(self, x, act=None, slope=0.2, blur=False):
        return ops.bias_act(x, self.bias, bias_scale=self.lrmul if self.use_lrmul else 1.0, act=act, slope=slope,
                            blur=blur)


class LinearEx(nn.Module):
    def __init__(self, nin_feat, nout_feat, init='xavier', init_type='default', gain_sq_base=2.,
                 equalized_lr=False, lrmul=1., include_bias=True):
        super().__init__()
        self.nin_feat, self.nout_feat = int(nin_feat), int(nout_feat)
        init = init.casefold() if init is not None else None
        init_type = init_type.casefold()
        self.equalized_lr = equalized_lr
        self.use_lrmul = True if lrmul != 1. else False
        self.lrmul = lrmul
        self.initializer = None
        if init_type != 'standard normal':
            self.initializer = Initializer(init=init, init_type=init_type, gain_sq_base=gain_sq_base,
                                           equalized_lr=equalized_lr)
        self.linear = nn.Linear(self.nin_feat, self.nout_feat, bias=include_bias)
        self.wscale = None
        if init_type in ('default', 'resnet',) and init is not None:
            bound = self.initializer.get_init_bound_layer(tensor=self.linear.weight, distribution_type='Uniform')
            if equalized_lr:
                self.wscale = bound
                self.linear.weight.data.uniform_(-1. / lrmul, 1. / lrmul)
            else:
                self.linear.weight.data.uniform_(-bound / lrmul, bound / lrmul)
        else:
            self.wscale = _init_weight(self.linear.weight, self.initializer, init, init_type, equalized_lr, lrmul,
                                       self.use_lrmul)
        self.bias = None
        if include_bias:
            self.linear.bias.data.fill_(0)

    @property
    def scale(self):
        s = self.wscale if (self.equalized_lr and self.wscale is not None) else 1.0
        return s * (self.lrmul if self.use_lrmul else 1.0)

    def forward(self, x, act=None, slope=0.2):
        return ops.linear(x, self.linear.weight, self.linear.bias, scale=self.scale,
                          bias_scale=self.lrmul if self.use_lrmul else 1.0, act=act, slope=slope)


class LinearBias(nn.Module):
    def __init__(self, nout_feat, lrmul=1., device='cpu'):
        super().__init__()
        self.use_lrmul = True if lrmul != 1. else False
        self.lrmul = lrmul
        self.bias = nn.Parameter(torch.zeros(1, nout_feat, device=device))

    def forward(self, x, act=None, slope=0.2):
        return ops.bias_act(x, self.bias, bias_scale=self.lrmul if self.use_lrmul else 1.0, act=act, slope=slope)


# -- peephole executor ------------------------------------------------------------------------------ #
def _folds_act_grad(conv, after, width):
    """``conv`` (fed a ``width``-wide map, followed by ``after``) applies the previous layer's LeakyReLU derivative in
    its dgrad epilogue: a plain 3x3 'same' conv (no pooling behind it) on the fp32 kernels, rows of >= 16 pixels."""
    return isinstance(conv, Conv2dEx) and not is_avg_pool(after) and conv.conv2d.kernel_size == (3, 3) and \
        conv.padding == 1 and width >= 16 and width % 4 == 0 and ops.get_compute_dtype() == 'f32'


def fused_sequential(mods, x):
    """Run a list of modules with kernel fusion where the pattern allows:
         Upsample2x, Conv2dEx                      -> stride-2 transposed kernel (upsample folded in)
         Conv2dEx, AvgPool2x [, Conv2dBias] [, LeakyReLU] -> stride-2 kernel (pool + bias + act folded in)
         Conv2dEx / LinearEx [, LeakyReLU]         -> bias + LeakyReLU in the MFMA epilogue
         Conv2dBias / LinearBias [, LeakyReLU]     -> one bias+act pass
         Conv2dEx, LeakyReLU, Blur2d               -> blur backward fused with LeakyReLU' + bias gradient
         Blur2d, Conv2dBias [, LeakyReLU]          -> one blur+bias+act pass
         NormalizeLayer, LeakyReLU                 -> Batch / LayerNorm apply the activation (and its backward) themselves
       Everything else falls through to the module's own (HIP) forward.  nn.Sequential children are
       flattened first."""
    flat = []

    def add(m):
        if m is None:
            return
        if isinstance(m, nn.Sequential):
            for c in m:
                add(c)
        else:
            flat.append(m)

    for m in mods:
        add(m)
    i, n = 0, len(flat)
    pending_slope = None     # the previous conv left its LeakyReLU derivative to the conv that comes next
    blur_handoff = None      # the previous conv -> LeakyReLU -> blur offers its blur^T / LeakyReLU' to the pooled conv behind it
    rgb_handoff = getattr(x, ops.RGB_HANDOFF, None)     # x is a fromRGB output with this sequence as its only reader
    while i < n:
        m = flat[i]
        handoff, blur_handoff = blur_handoff, None
        rgb, rgb_handoff = rgb_handoff, None
        up = False
        up_after = None
        if isinstance(m, Upsample2x) and i + 1 < n and isinstance(flat[i + 1], Conv2dEx):
            if flat[i + 1].ks == 1 and flat[i + 1].padding == 0 and i + 2 == n and \
                    os.environ.get('GANLAB_UP_COMMUTE') != '0':
                # a pointwise conv (with its bias) commutes with the nearest upsample element for element: run it on the
                # quarter-size map and upsample its result - the residual blocks' [upsample, 1x1 conv] skip, resblocks.py:55
                # (only as the END of a sequence: nothing behind it can then pair up with the conv)
                up_after = m
            else:
                up = True
            i += 1
            m = flat[i]
        nxt = flat[i + 1] if i + 1 < n else None
        if isinstance(m, Blur2d) and isinstance(nxt, Conv2dBias):
            kw = {'blur': True}
            i += 1
            m = nxt
            nxt = flat[i + 1] if i + 1 < n else None
            if isinstance(nxt, LeakyReLU):
                kw.update(act='lrelu', slope=nxt.negative_slope)
                i += 1
            assert pending_slope is None
            x = m(x, **kw)
        elif isinstance(m, (Conv2dEx, LinearEx, Conv2dBias, LinearBias)):
            kw = {}
            if isinstance(m, Conv2dEx) and not up and is_avg_pool(nxt):
                # conv -> AvgPool2d(2) [-> Conv2dBias] [-> LeakyReLU]: one stride-2 kernel (a bias of
                # the conv itself commutes with the pooling: pool(conv + b) = pool(conv) + b)
                kw['pool'] = True
                i += 1
                nxt = flat[i + 1] if i + 1 < n else None
                if m.conv2d.bias is None and isinstance(nxt, Conv2dBias):
                    kw['bias_mod'] = nxt
                    i += 1
                    nxt = flat[i + 1] if i + 1 < n else None
            if isinstance(nxt, LeakyReLU):
                kw.update(act='lrelu', slope=nxt.negative_slope)
                i += 1
                if isinstance(m, Conv2dEx) and not up and 'pool' not in kw and i + 1 < n and \
                        isinstance(flat[i + 1], Blur2d):
                    kw['blur'] = True
                    i += 1
            if up:
                kw['up'] = True
            if isinstance(m, Conv2dEx):
                if handoff is not None and 'pool' in kw:
                    kw['in_blur_handoff'] = handoff       # x has this one reader: ops.BlurHandoff
                if rgb is not None and not up and 'pool' not in kw:
                    kw['in_rgb_handoff'] = rgb            # ops.RgbHandoff
                if pending_slope is not None:
                    kw['in_act_slope'] = pending_slope
                # conv + LeakyReLU feeding the next conv directly (D block k -> block k+1): that conv's dgrad epilogue
                # applies this LeakyReLU's derivative, saving the separate pass over this layer's output gradient
                if 'act' in kw and 'blur' not in kw and not up and i + 1 < n and \
                        _folds_act_grad(flat[i + 1], flat[i + 2] if i + 2 < n else None,
                                        x.shape[-1] // (2 if 'pool' in kw else 1)):
                    kw['defer_act_grad'] = True
            else:
                assert pending_slope is None
            x = m(x, **kw)
            pending_slope = kw['slope'] if getattr(x, ops.ACT_DEFERRED, False) else None
            blur_handoff = getattr(x, ops.BLUR_HANDOFF, None) if kw.get('blur') else None
            rgb_handoff = getattr(x, ops.RGB_HANDOFF, None)
            if up_after is not None:
                assert pending_slope is None and blur_handoff is None
                x = up_after(x)
                rgb_handoff = None
        elif isinstance(m, NormalizeLayer) and isinstance(nxt, LeakyReLU) and os.environ.get('GANLAB_NORM_ACT') != '0':
            # NormalizeLayer, LeakyReLU (every residual block, resblocks.py:48-49) -> the activation and its backward ride in
            # the normalisation's own passes
            assert pending_slope is None
            x = m(x, act_slope=nxt.negative_slope)
            i += 1
        else:
            assert pending_slope is None
            x = m(x)
        i += 1
    assert pending_slope is None
    return x
