"""ProGAN generator and the ProGAN / StyleGAN discriminator on the HIP kernels (drop-in for
gan_lab/progan/architectures.py:31-318).  Module tree and ``state_dict`` keys follow the reference;
``forward`` runs every ``nn.Sequential`` through the peephole executor
(``utils.custom_layers.fused_sequential``), so e.g. a discriminator block

    Seq( Seq(Conv2dEx+b, lrelu), Seq(blur, Conv2dEx, AvgPool, Conv2dBias, lrelu) )        (:254-284)

is: 1 MFMA conv kernel (bias+LeakyReLU epilogue) -> blur kernel -> MFMA conv kernel -> pool kernel ->
bias+LeakyReLU kernel, all double-differentiable for the R1 / WGAN-GP penalty.
"""
import copy

from torch import nn

from .. import ops
from .._int import FMAP_SAMPLES, RES_INIT
from ..progressive import ProgressiveBase, StyleGAN
from ..utils.custom_layers import (AvgPool2x, Conv2dBias, Conv2dEx, Lambda, LeakyReLU, LinearEx, NormalizeLayer,
                                   Upsample2x, concat_mbstd_layer, fused_sequential, get_blur_op, own_resampler)
from .base import ProGAN

FMAP_G_INIT_FCTR = 1
FMAP_D_END_FCTR = 1


class ProGenerator(ProGAN):
    """Progressively Growing GAN (Karras et al. 2018) generator."""

    def __init__(self, final_res, len_latent=512, upsampler=None, blur_type=None, nl=None, num_classes=0,
                 equalized_lr=True, normalize_z=True, use_pixelnorm=True):
        super().__init__(final_res)
        self.gen_blocks = nn.ModuleList()
        self.upsampler = own_resampler(upsampler) if upsampler is not None else Upsample2x()
        self.gen_blur_type = blur_type
        self.nl = nl if nl is not None else LeakyReLU(.2)
        self.len_latent, self.num_classes = len_latent, num_classes
        self.equalized_lr = equalized_lr
        self.use_pixelnorm = use_pixelnorm
        norms = self._norms
        if normalize_z:
            self.preprocess_z = nn.Sequential(Lambda(lambda x: x.view(-1, len_latent + num_classes)),
                                              NormalizeLayer('PixelNorm'))
        else:
            self.preprocess_z = Lambda(lambda x: x.view(-1, len_latent + num_classes))
        _fmap_init = len_latent * FMAP_G_INIT_FCTR
        self.gen_blocks.append(nn.Sequential(
            LinearEx(nin_feat=len_latent + num_classes, nout_feat=_fmap_init * RES_INIT ** 2, init='He',
                     init_type='ProGAN', gain_sq_base=2. / 16, equalized_lr=equalized_lr),
            Lambda(lambda x: x.view(-1, _fmap_init, RES_INIT, RES_INIT)),
            self.nl, *norms(),
            Conv2dEx(ni=_fmap_init, nf=self.fmap, ks=3, stride=1, padding=1, init='He', init_type='ProGAN',
                     gain_sq_base=2., equalized_lr=equalized_lr),
            self.nl, *norms()))
        self.prev_torgb = None
        self._update_torgb(ni=self.fmap)

    def _norms(self):
        return [NormalizeLayer('PixelNorm')] if self.use_pixelnorm else []

    def increase_scale(self):
        if not self.scale_inc_metadata_updated:
            super().increase_scale()
        else:
            self.scale_inc_metadata_updated = False
        blur_op = get_blur_op(self.gen_blur_type, self.fmap) if self.gen_blur_type is not None else None
        dev = next(self.parameters()).device
        self.gen_blocks.append(nn.Sequential(
            self.get_conv_layer(ni=self.fmap_prev, upsample=True, blur_op=blur_op),
            self.get_conv_layer(ni=self.fmap)).to(dev))
        self.prev_torgb = copy.deepcopy(self.torgb)
        self._update_torgb(ni=self.fmap)
        self.torgb.to(dev)

    def get_conv_layer(self, ni, upsample=False, blur_op=None, append_nl=True):
        upsampler = [self.upsampler] if upsample else []
        conv = Conv2dEx(ni=ni, nf=self.fmap, ks=3, stride=1, padding=1, init='He', init_type='ProGAN',
                        gain_sq_base=2., equalized_lr=self.equalized_lr, include_bias=blur_op is None)
        blur = [blur_op] if blur_op is not None else []
        bias = [Conv2dBias(nf=self.fmap)] if blur_op is not None else []
        nl = [self.nl] if append_nl else []
        return nn.Sequential(*upsampler, conv, *(blur + bias + nl + self._norms()))

    def _update_torgb(self, ni):
        self.torgb = Conv2dEx(ni=ni, nf=FMAP_SAMPLES, ks=1, stride=1, padding=0, init='He', init_type='ProGAN',
                              gain_sq_base=1., equalized_lr=self.equalized_lr)

    def forward(self, x):
        x = self.preprocess_z(x)
        for gen_block in self.gen_blocks[:-1]:
            x = fused_sequential([gen_block], x)
        img = self.torgb(fused_sequential([self.gen_blocks[-1]], x))
        if self.fade_in_phase:
            img = ops.lerp(ops.upsample2(self.prev_torgb(x)), img, self.alpha)      # (:163-165)
        return img


class _DiscriminatorBody(object):
    """Body shared by ProDiscriminator (ProGAN family) and StyleDiscriminator (StyleGAN family) - the
    reference synthesises the latter from the former at run time (stylegan/learner.py:141-151)."""

    def _build(self, final_res, pooler=None, blur_type=None, nl=None, num_classes=0, equalized_lr=True,
               mbstd_group_size=4):
        self.init_type = 'StyleGAN' if isinstance(self, StyleGAN) else 'ProGAN'
        self.disc_blocks = nn.ModuleList()
        self.num_classes = num_classes
        self.pooler = own_resampler(pooler) if pooler is not None else AvgPool2x()
        self.disc_blur_type = blur_type
        self.nl = nl if nl is not None else LeakyReLU(.2)
        self.equalized_lr = equalized_lr
        self.mbstd_group_size = mbstd_group_size
        mbstd_layer = self.get_mbstd_layer()
        self.prev_fromrgb = None
        self._update_fromrgb(nf=self.fmap)
        _fmap_end = self.fmap * FMAP_D_END_FCTR
        self.disc_blocks.insert(0, nn.Sequential(
            *mbstd_layer,
            Conv2dEx(ni=self.fmap + (1 if mbstd_layer else 0), nf=self.fmap, ks=3, stride=1, padding=1, init='He',
                     init_type=self.init_type, gain_sq_base=2., equalized_lr=equalized_lr),
            self.nl,
            Conv2dEx(ni=self.fmap, nf=_fmap_end, ks=4, stride=1, padding=0, init='He', init_type=self.init_type,
                     gain_sq_base=2., equalized_lr=equalized_lr),
            self.nl,
            Lambda(lambda x: x.view(-1, _fmap_end)),
            LinearEx(nin_feat=_fmap_end, nout_feat=1, init='He', init_type=self.init_type, gain_sq_base=1.,
                     equalized_lr=equalized_lr)))

    def increase_scale(self):
        if not self.scale_inc_metadata_updated:
            ProgressiveBase.increase_scale(self)
        else:
            self.scale_inc_metadata_updated = False
        dev = next(self.parameters()).device
        self.prev_fromrgb = copy.deepcopy(self.fromrgb)
        self._update_fromrgb(nf=self.fmap)
        self.fromrgb.to(dev)
        blur_op = get_blur_op(self.disc_blur_type, self.fmap) if self.disc_blur_type is not None else None
        self.disc_blocks.insert(0, nn.Sequential(
            self.get_conv_layer(nf=self.fmap),
            self.get_conv_layer(nf=self.fmap_prev, downsample=True, blur_op=blur_op)).to(dev))

    def get_conv_layer(self, nf, downsample=False, blur_op=None, append_nl=True):
        blur = [blur_op] if blur_op is not None else []
        conv = Conv2dEx(ni=self.fmap, nf=nf, ks=3, stride=1, padding=1, init='He', init_type=self.init_type,
                        gain_sq_base=2., equalized_lr=self.equalized_lr, include_bias=not downsample)
        pooler = [self.pooler] if downsample else []
        bias = [Conv2dBias(nf=nf)] if downsample else []
        nl = [self.nl] if append_nl else []
        return nn.Sequential(*blur, conv, *(pooler + bias + nl))

    def _update_fromrgb(self, nf):
        self.fromrgb = nn.Sequential(
            Conv2dEx(ni=FMAP_SAMPLES + self.num_classes, nf=nf, ks=1, stride=1, padding=0, init='He',
                     init_type=self.init_type, gain_sq_base=2., equalized_lr=self.equalized_lr),
            self.nl)

    def get_mbstd_layer(self):
        if self.mbstd_group_size == -1:
            return []
        return [Lambda(lambda x, group_size: concat_mbstd_layer(x, group_size), group_size=self.mbstd_group_size)]

    def forward(self, x):
        shape = (FMAP_SAMPLES + self.num_classes, self.curr_res, self.curr_res)
        if x.dim() != 4 or tuple(x.shape[1:]) != shape:
            # (a batch that already has this shape is taken as is: a view of a leaf is a node of its own, and
            # ops.no_grad_towards / ops.RgbHandoff recognise the real batch by being that leaf)
            x = x.view(-1, *shape)
        h = fused_sequential([self.fromrgb], x)
        rest = list(self.disc_blocks)
        if self.fade_in_phase:
            h = fused_sequential([rest.pop(0)], h)
            skip = fused_sequential([self.prev_fromrgb], ops.avg_pool2(x))
            h = ops.lerp(skip, h, self.alpha)                                        # (:311-313)
        # all blocks as ONE sequence: where block k's last LeakyReLU feeds block k+1's first conv directly, that conv's
        # input-gradient kernel applies the LeakyReLU derivative (fused_sequential)
        return fused_sequential(rest, h).view(-1)


class ProDiscriminator(_DiscriminatorBody, ProGAN):
    """Progressively Growing GAN discriminator / critic."""

    def __init__(self, final_res, **kw):
        ProGAN.__init__(self, final_res)
        self._build(final_res, **kw)


class StyleDiscriminator(_DiscriminatorBody, StyleGAN):
    """The same body in the StyleGAN family (shares alpha / resolution state with StyleGenerator)."""

    def __init__(self, final_res, **kw):
        StyleGAN.__init__(self, final_res)
        self._build(final_res, **kw)
