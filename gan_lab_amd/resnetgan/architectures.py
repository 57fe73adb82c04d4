"""ResNet GAN generators / discriminators on the HIP path (drop-in for
gan_lab/resnetgan/architectures.py:29-224): same class names, constructor arguments, module tree and
``state_dict`` keys; every tensor op runs in the hand-written kernels (gan_lab_amd.ops)."""
from torch import nn

from .. import ops
from .._int import FMAP_SAMPLES, RES_INIT
from ..utils.custom_layers import Conv2dEx, Lambda, LinearEx, NormalizeLayer, Tanh, fused_sequential
from .base import GAN
from .resblocks import FastResBlock2dDownsample, ResBlock2d, ResBlock2d32Pix, _own_nl, _own_resampler

FMAP_G = 64
FMAP_D = 64
FMAP_G_INIT_32_FCTR = 1
FMAP_G_INIT_64_FCTR = 4
RES_FEATURE_SPACE = 4


def _run(seq, x):
    """Children of a Sequential through the peephole executor, ResBlocks through their own forward."""
    run = []
    for m in seq:
        if isinstance(m, (ResBlock2d, FastResBlock2dDownsample)):
            if run:
                x = fused_sequential(run, x)
                run = []
            x = m(x)
        else:
            run.append(m)
    return fused_sequential(run, x) if run else x


class _ResnetGenerator(GAN):
    def forward(self, x):
        return _run(self.generator_model, x)


class Generator32PixResnet(_ResnetGenerator):
    """32-pixel ResNet generator with optional class conditioning (architectures.py:29-59)."""

    def __init__(self, len_latent=128, fmap=FMAP_G * 2, upsampler=None, blur_type=None, nl=None, num_classes=0,
                 equalized_lr=False):
        super().__init__(32)
        from ..utils.custom_layers import Upsample2x
        upsampler = _own_resampler(upsampler) if upsampler is not None else Upsample2x()
        nl = _own_nl(nl)
        self.len_latent, self.num_classes, self.equalized_lr = len_latent, num_classes, equalized_lr
        f0 = len_latent * FMAP_G_INIT_32_FCTR
        kw = dict(ks=3, norm_type='BatchNorm', upsampler=upsampler, init='He', nl=nl, equalized_lr=equalized_lr,
                  blur_type=blur_type)
        self.generator_model = nn.Sequential(
            Lambda(lambda x: x.view(-1, len_latent + num_classes)),
            LinearEx(nin_feat=len_latent + num_classes, nout_feat=f0 * RES_INIT ** 2, init='Xavier',
                     equalized_lr=equalized_lr),
            Lambda(lambda x: x.view(-1, f0, RES_INIT, RES_INIT)),
            ResBlock2d32Pix(ni=f0, nf=fmap, **kw),
            ResBlock2d32Pix(ni=fmap, nf=fmap, **kw),
            ResBlock2d32Pix(ni=fmap, nf=fmap, **kw),
            NormalizeLayer('BatchNorm', ni=fmap),
            nl,
            Conv2dEx(ni=fmap, nf=FMAP_SAMPLES, ks=3, stride=1, padding=1, init='Xavier', equalized_lr=equalized_lr),
            Tanh(),
        )


class Generator64PixResnet(_ResnetGenerator):
    """64-pixel ResNet generator with optional class conditioning (architectures.py:62-97)."""

    def __init__(self, len_latent=128, fmap=FMAP_G, upsampler=None, blur_type=None, nl=None, num_classes=0,
                 equalized_lr=False):
        super().__init__(64)
        from ..utils.custom_layers import Upsample2x
        upsampler = _own_resampler(upsampler) if upsampler is not None else Upsample2x()
        nl = _own_nl(nl)
        self.len_latent, self.num_classes, self.equalized_lr = len_latent, num_classes, equalized_lr
        f0 = len_latent * FMAP_G_INIT_64_FCTR
        kw = dict(ks=3, norm_type='BatchNorm', upsampler=upsampler, init='He', nl=nl, equalized_lr=equalized_lr,
                  blur_type=blur_type)
        self.generator_model = nn.Sequential(
            Lambda(lambda x: x.view(-1, len_latent + num_classes)),
            LinearEx(nin_feat=len_latent + num_classes, nout_feat=f0 * RES_INIT ** 2, init='Xavier',
                     equalized_lr=equalized_lr),
            Lambda(lambda x: x.view(-1, f0, RES_INIT, RES_INIT)),
            ResBlock2d(ni=f0, nf=8 * fmap, **kw),
            ResBlock2d(ni=8 * fmap, nf=4 * fmap, **kw),
            ResBlock2d(ni=4 * fmap, nf=2 * fmap, **kw),
            ResBlock2d(ni=2 * fmap, nf=1 * fmap, **kw),
            NormalizeLayer('BatchNorm', ni=1 * fmap),
            nl,
            Conv2dEx(ni=1 * fmap, nf=FMAP_SAMPLES, ks=3, stride=1, padding=1, init='He', equalized_lr=equalized_lr),
            Tanh(),
        )


class Discriminator32PixResnet(GAN):
    """32-pixel ResNet discriminator / critic (architectures.py:103-133)."""

    def __init__(self, fmap=FMAP_D * 2, pooler=None, blur_type=None, nl=None, num_classes=0, equalized_lr=False):
        super().__init__(32)
        from ..utils.custom_layers import AvgPool2x
        pooler = _own_resampler(pooler) if pooler is not None else AvgPool2x()
        nl = _own_nl(nl)
        self.num_classes, self.equalized_lr = num_classes, equalized_lr
        self.view1 = Lambda(lambda x: x.view(-1, FMAP_SAMPLES + num_classes, self.res, self.res))
        self.conv1 = FastResBlock2dDownsample(ni=FMAP_SAMPLES + num_classes, nf=fmap, ks=3, pooler=pooler,
                                              init='Xavier', nl=nl, equalized_lr=equalized_lr, blur_type=blur_type)
        kw = dict(ks=3, norm_type='LayerNorm', init='He', nl=nl, equalized_lr=equalized_lr, blur_type=blur_type)
        self.resblocks = nn.Sequential(
            ResBlock2d32Pix(ni=fmap, nf=fmap, pooler=pooler, res=self.res // 2, **kw),
            ResBlock2d32Pix(ni=fmap, nf=fmap, res=self.res // 4, **kw),
            ResBlock2d32Pix(ni=fmap, nf=fmap, res=self.res // 4, **kw),
            nl,
            Lambda(ops.global_avg_pool),      # nn.AvgPool2d(kernel_size=res//4) on the (res//4)^2 map
            Lambda(lambda x: x.view(-1, fmap)),
        )
        self.linear1 = LinearEx(nin_feat=fmap, nout_feat=1, init='Xavier', equalized_lr=equalized_lr)

    def features(self, x):
        return _run(self.resblocks, self.conv1(self.view1(x)))

    def forward(self, x):
        return self.linear1(self.features(x)).view(-1)


class Discriminator64PixResnet(GAN):
    """64-pixel ResNet discriminator / critic (architectures.py:157-187)."""

    def __init__(self, fmap=FMAP_D, pooler=None, blur_type=None, nl=None, num_classes=0, equalized_lr=False):
        super().__init__(64)
        from ..utils.custom_layers import AvgPool2x
        pooler = _own_resampler(pooler) if pooler is not None else AvgPool2x()
        nl = _own_nl(nl)
        self.num_classes, self.equalized_lr = num_classes, equalized_lr
        self.view1 = Lambda(lambda x: x.view(-1, FMAP_SAMPLES + num_classes, self.res, self.res))
        self.conv1 = Conv2dEx(ni=FMAP_SAMPLES + num_classes, nf=1 * fmap, ks=3, stride=1, padding=1, init='Xavier',
                              equalized_lr=equalized_lr)
        kw = dict(ks=3, norm_type='LayerNorm', pooler=pooler, init='He', nl=nl, equalized_lr=equalized_lr,
                  blur_type=blur_type)
        self.resblocks = nn.Sequential(
            ResBlock2d(ni=1 * fmap, nf=2 * fmap, res=self.res // 1, **kw),
            ResBlock2d(ni=2 * fmap, nf=4 * fmap, res=self.res // 2, **kw),
            ResBlock2d(ni=4 * fmap, nf=8 * fmap, res=self.res // 4, **kw),
            ResBlock2d(ni=8 * fmap, nf=8 * fmap, res=self.res // 8, **kw),
            Lambda(lambda x: x.view(-1, RES_FEATURE_SPACE ** 2 * 8 * fmap)),
        )
        self.linear1 = LinearEx(nin_feat=RES_FEATURE_SPACE ** 2 * 8 * fmap, nout_feat=1, init='Xavier',
                                equalized_lr=equalized_lr)

    def features(self, x):
        return _run(self.resblocks, self.conv1(self.view1(x)))

    def forward(self, x):
        return self.linear1(self.features(x)).view(-1)


class DiscriminatorAC32PixResnet(Discriminator32PixResnet):
    """Auxiliary-classifier variant, discriminator class-conditioning removed (architectures.py:136-154)."""

    def __init__(self, fmap=FMAP_D * 2, pooler=None, blur_type=None, nl=None, num_classes=0, equalized_lr=False):
        super().__init__(fmap, pooler, blur_type, nl, num_classes, equalized_lr)
        self.view1 = Lambda(lambda x: x.view(-1, FMAP_SAMPLES, self.res, self.res))
        self.conv1 = FastResBlock2dDownsample(ni=FMAP_SAMPLES, nf=fmap, ks=3, pooler=pooler, init='Xavier',
                                              nl=nl, equalized_lr=equalized_lr, blur_type=blur_type)
        self.linear_aux = LinearEx(nin_feat=fmap, nout_feat=num_classes, init='Xavier', equalized_lr=equalized_lr)

    def forward(self, x):
        f = self.features(x)
        return self.linear1(f).view(-1), self.linear_aux(f)


class DiscriminatorAC64PixResnet(Discriminator64PixResnet):
    """Auxiliary-classifier variant (architectures.py:190-207)."""

    def __init__(self, fmap=FMAP_D, pooler=None, blur_type=None, nl=None, num_classes=0, equalized_lr=False):
        super().__init__(fmap, pooler, blur_type, nl, num_classes, equalized_lr)
        self.view1 = Lambda(lambda x: x.view(-1, FMAP_SAMPLES, self.res, self.res))
        self.conv1 = Conv2dEx(ni=FMAP_SAMPLES, nf=1 * fmap, ks=3, stride=1, padding=1, init='Xavier',
                              equalized_lr=equalized_lr)
        self.linear_aux = LinearEx(nin_feat=RES_FEATURE_SPACE ** 2 * 8 * fmap, nout_feat=num_classes, init='Xavier',
                                   equalized_lr=equalized_lr)

    def forward(self, x):
        f = self.features(x)
        return self.linear1(f).view(-1), self.linear_aux(f)
