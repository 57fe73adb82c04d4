"""Residual blocks of the ResNet GANs on the HIP path (drop-in for gan_lab/resnetgan/resblocks.py).

Same constructor arguments, child-module order and ``state_dict`` keys (``conv_layer_1.<i>.norm.*``,
``conv_layer_1.<i>.conv2d.*``, ``skip_connection.<i>.conv2d.*``).  The Sequentials are parameter
containers; ``forward`` hands their children to the peephole executor so that
``Upsample -> conv`` and ``conv -> AvgPool`` each run as one stride-2 MFMA kernel.
"""
from torch import nn

from .. import ops
from ..utils.custom_layers import (AvgPool2x, Conv2dEx, Lambda, LeakyReLU, NormalizeLayer, Upsample2x, own_nl, own_resampler,
                                   fused_sequential, get_blur_op)


def _own_nl(nl):
    """Accept the reference's nn.ReLU() / nn.LeakyReLU() / nn.Tanh() instances as well as the HIP-path modules."""
    return own_nl(nl, default_slope=0.)


_own_resampler = own_resampler


class ResBlock2d(nn.Module):
    """norm -> nl -> [up] conv [blur] -> norm -> nl -> conv [pool], plus a 1x1-conv skip (resblocks.py:15-64)."""

    def __init__(self, ni, nf, ks, norm_type, upsampler=None, pooler=None, init='He', nl=None, res=None,
                 flip_sampling=False, equalized_lr=False, blur_type=None):
        super().__init__()
        assert not (upsampler is not None and pooler is not None)
        upsampler, pooler, nl = _own_resampler(upsampler), _own_resampler(pooler), _own_nl(nl)
        padding = (ks - 1) // 2
        if not flip_sampling:
            self.nif = nf if (upsampler is not None and pooler is None) else ni
        else:
            self.nif = ni if (upsampler is None and pooler is not None) else nf
        self.convs = (
            Conv2dEx(ni, self.nif, ks=ks, stride=1, padding=padding, init=init, equalized_lr=equalized_lr),
            Conv2dEx(self.nif, nf, ks=ks, stride=1, padding=padding, init=init, equalized_lr=equalized_lr),
            Conv2dEx(ni, nf, ks=1, stride=1, padding=0, init='Xavier', equalized_lr=equalized_lr),
        )
        blur_op = get_blur_op(blur_type=blur_type, num_channels=self.convs[0].nf) if blur_type is not None else None
        norm_nls = ([NormalizeLayer(norm_type, ni=ni, res=res), nl],
                    [NormalizeLayer(norm_type, ni=self.convs[0].nf, res=res), nl])
        if upsampler is not None:
            op1 = [upsampler, self.convs[0], blur_op] if blur_type is not None else [upsampler, self.convs[0]]
            op2 = [upsampler, self.convs[2], blur_op] if blur_type is not None else [upsampler, self.convs[2]]
            _ops = (op1, [self.convs[1]], op2,)
        elif pooler is not None:
            op1 = [blur_op, self.convs[1], pooler] if blur_type is not None else [self.convs[1], pooler]
            op2 = [blur_op, pooler, self.convs[2]] if blur_type is not None else [pooler, self.convs[2]]
            _ops = ([self.convs[0]], op1, op2,)
        else:
            _ops = ([self.convs[0]], [self.convs[1]], [self.convs[2]],)
        self.conv_layer_1 = nn.Sequential(*(norm_nls[0] + _ops[0]))
        self.conv_layer_2 = nn.Sequential(*(norm_nls[1] + _ops[1]))
        if (upsampler is not None or pooler is not None) or ni != nf:
            self.skip_connection = nn.Sequential(*(_ops[2]))
        else:
            self.skip_connection = Lambda(lambda x: x)

    def forward(self, x):
        skip = fused_sequential([self.skip_connection], x)
        return ops.add(skip, fused_sequential([self.conv_layer_1, self.conv_layer_2], x))


class ResBlock2d32Pix(ResBlock2d):
    """resblocks.py:67-80: flip_sampling default True; the pooling skip is conv1x1 -> pool."""

    def __init__(self, ni, nf, ks, norm_type, upsampler=None, pooler=None, init='He', nl=None, res=None,
                 flip_sampling=True, equalized_lr=False, blur_type=None):
        super().__init__(ni, nf, ks, norm_type, upsampler, pooler, init, nl, res, flip_sampling, equalized_lr,
                         blur_type)
        pooler = _own_resampler(pooler)
        if upsampler is None and pooler is not None:
            self.skip_connection = nn.Sequential(self.convs[2], pooler)


class FastResBlock2dDownsample(nn.Module):
    """Downsampling block without normalisation / activation before the first conv (resblocks.py:83-124)."""

    def __init__(self, ni, nf, ks, pooler=None, init='He', nl=None, equalized_lr=False, blur_type=None):
        super().__init__()
        pooler = _own_resampler(pooler) if pooler is not None else AvgPool2x()
        nl = _own_nl(nl)
        padding = (ks - 1) // 2
        self.conv_layer_1 = nn.Sequential(
            Conv2dEx(ni, nf, ks=ks, stride=1, padding=padding, init='he', equalized_lr=equalized_lr), nl)
        self.conv_layer_2 = nn.Sequential()
        self.skip_connection = nn.Sequential()
        seq_n = 0
        if blur_type is not None:
            blur_op = get_blur_op(blur_type=blur_type, num_channels=nf)
            self.conv_layer_2.add_module(str(seq_n), blur_op)
            self.skip_connection.add_module(str(seq_n), blur_op)
            seq_n += 1
        self.conv_layer_2.add_module(str(seq_n), Conv2dEx(nf, nf, ks=ks, stride=1, padding=padding, init='he',
                                                          equalized_lr=equalized_lr))
        self.skip_connection.add_module(str(seq_n), pooler)
        seq_n += 1
        self.conv_layer_2.add_module(str(seq_n), pooler)
        self.skip_connection.add_module(str(seq_n), Conv2dEx(ni, nf, ks=1, stride=1, padding=0, init='xavier',
                                                             equalized_lr=equalized_lr))

    def forward(self, x):
        skip = fused_sequential([self.skip_connection], x)
        return ops.add(skip, fused_sequential([self.conv_layer_1, self.conv_layer_2], x))
