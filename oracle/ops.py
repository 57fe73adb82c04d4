"""Layer-op oracle (CPU torch fp32).  TEST INFRASTRUCTURE - see oracle/__init__.py.

Reference paths are relative to /root/reference/gan_lab.
"""
import math

import torch
import torch.nn.functional as F


# -- utils/initializer.py:27-39,65-80 ---------------------------------------------------------- #
def he_std(fan_in, gain_sq_base=2.0):
    """Runtime eq-LR scale for init_type 'progan'/'stylegan', init 'he': fan-in only.

    initializer.py:66 ``gain_sq = gain_sq_base / 2``; :75 ``he -> gain_sq *= 2``; :78
    ``std = sqrt(gain_sq / fan)`` with fan = fan_in (:35-37, :71-72).
    """
    return math.sqrt((gain_sq_base / 2.0) * 2.0 / fan_in)


def conv_wscale(weight, gain_sq_base=2.0):
    """fan_in = Cin * kh * kw (initializer.py:52-56)."""
    return he_std(weight.shape[1] * weight.shape[2] * weight.shape[3], gain_sq_base)


def linear_wscale(weight, gain_sq_base=2.0):
    """fan_in = in_features (initializer.py:47-49)."""
    return he_std(weight.shape[1], gain_sq_base)


# -- utils/custom_layers.py:202-211 ------------------------------------------------------------- #
def conv2d_ex(x, weight, bias=None, wscale=None, padding=0, lrmul=1.0):
    """Equalised-LR conv: the *input* is scaled (custom_layers.py:204), bias unscaled, then the
    whole output (bias included) is multiplied by lrmul when lrmul != 1 (:208-209)."""
    if wscale is not None:
        x = x * wscale
    y = F.conv2d(x, weight, bias, stride=1, padding=padding)
    if lrmul != 1.0:
        y = y * lrmul
    return y


# -- utils/custom_layers.py:282-291 ------------------------------------------------------------- #
def linear_ex(x, weight, bias=None, wscale=None, lrmul=1.0):
    if wscale is not None:
        x = x * wscale
    y = F.linear(x, weight, bias)
    if lrmul != 1.0:
        y = y * lrmul
    return y


# -- utils/custom_layers.py:36-53 --------------------------------------------------------------- #
def blur_binomial(x):
    """Depthwise 3x3 [1 2 1]x[1 2 1]/16, stride 1, zero padding 1 (custom_layers.py:41-51)."""
    c = x.shape[1]
    k = torch.tensor([[1., 2., 1.], [2., 4., 2.], [1., 2., 1.]], dtype=x.dtype, device=x.device) / 16.
    return F.conv2d(x, k.expand(c, 1, 3, 3), stride=1, padding=1, groups=c)


# -- utils/custom_layers.py:81-86 --------------------------------------------------------------- #
def pixelnorm(x, eps=1e-8):
    return x * ((x ** 2).mean(dim=1, keepdim=True) + eps).rsqrt()


# -- utils/custom_layers.py:98-99 (nn.InstanceNorm2d(None, eps=1e-8)) ---------------------------- #
def instancenorm(x, eps=1e-8):
    """Biased variance over HxW per (n, c); no affine, no running stats."""
    mu = x.mean(dim=(2, 3), keepdim=True)
    var = x.var(dim=(2, 3), unbiased=False, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps)


# -- utils/custom_layers.py:117-140 ------------------------------------------------------------- #
def mbstd_concat(x, group_size=4):
    """Minibatch-stddev: contiguous groups (:129), UNBIASED variance over the group (:130),
    sqrt(var + 1e-8) (:131), mean over C*H*W (:132-133), one extra constant channel (:134-140).
    Falls back to the whole batch when B % group_size != 0 (:123-126)."""
    b, c, h, w = x.shape
    gs = min(b, group_size)
    if b % gs != 0:
        gs = b
    g = b // gs
    if gs > 1:
        m = x.view(g, gs, c, h, w)
        m = torch.var(m, dim=1)  # unbiased
        m = torch.sqrt(m + 1e-8)
        m = m.view(g, -1).mean(dim=1).view(g, 1, 1, 1, 1)
        m = m.expand(g, gs, 1, h, w).contiguous().view(b, 1, h, w)
    else:
        m = torch.zeros(b, 1, h, w, dtype=x.dtype, device=x.device)
    return torch.cat((x, m), dim=1)


# -- stylegan/architectures.py:112-119 ---------------------------------------------------------- #
def add_noise(x, noise_weight, noise):
    """x + noise_weight(1,C,1,1) * noise(B,1,H,W)."""
    return x + noise_weight * noise


# -- stylegan/architectures.py:524-526 ---------------------------------------------------------- #
def adain_affine(x, y):
    """y: (B, 2C) style vector -> x * (ys + 1) + yb."""
    b, c = x.shape[0], x.shape[1]
    y = y.view(b, 2, c, 1, 1)
    return x * (y[:, 0] + 1.0) + y[:, 1]


def lrelu(x, slope=0.2):
    return F.leaky_relu(x, slope)


def upsample2(x, mode='nearest', align_corners=False):
    """nn.Upsample(scale_factor=2, mode=...) as resnetgan/learner.py:147-158 builds it (align_corners only reaches
    the bilinear mode); the fade-in skip connections always call it with the default (progan/architectures.py:52-53)."""
    if mode == 'nearest':
        return F.interpolate(x, scale_factor=2, mode='nearest')
    return F.interpolate(x, scale_factor=2, mode=mode, align_corners=bool(align_corners))


def avgpool2(x):
    return F.avg_pool2d(x, kernel_size=2, stride=2)


def pool2(x, mode='average', align_corners=False):
    """The critic's pooler (resnetgan/learner.py:160-173): nn.AvgPool2d(2, 2) | NearestPool2d | BilinearPool2d
    (utils/custom_layers.py:59-75)."""
    if mode in ('average', 'box'):
        return avgpool2(x)
    if mode == 'nearest':
        return F.interpolate(x, scale_factor=.5, mode='nearest')
    return F.interpolate(x, scale_factor=.5, mode='bilinear', align_corners=bool(align_corners))


# -- utils/backprop_utils.py:19-49, progan/learner.py:791-812,883-896 ---------------------------- #
def loss_disc(kind, d_fake, d_real):
    if kind == 'wgan':
        return (d_fake - d_real).mean()
    # 'nonsaturating' and 'minimax' share the D loss (progan/learner.py:793-800)
    return (F.binary_cross_entropy_with_logits(d_fake, torch.zeros_like(d_fake)) +
            F.binary_cross_entropy_with_logits(d_real, torch.ones_like(d_real)))


def loss_gen(kind, d_fake):
    if kind == 'wgan':
        return -d_fake.mean()
    if kind == 'nonsaturating':
        return F.binary_cross_entropy_with_logits(d_fake, torch.ones_like(d_fake))
    if kind == 'minimax':
        return -F.binary_cross_entropy_with_logits(d_fake, torch.zeros_like(d_fake))
    raise ValueError(kind)
