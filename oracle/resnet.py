"""ResNet GAN oracle (CPU torch fp32): the non-progressive 32 / 64 pixel generators and critics and
their training iteration.  TEST INFRASTRUCTURE - see oracle/__init__.py.  Functional restatement
driven from reference-layout ``state_dict``s; reference paths relative to /root/reference/gan_lab.
Pinned by tests/golden/resnet{32,64}.npz (made by tests/golden/make_golden.py from the reference).

Only the default configuration of config #5 is restated: equalized_lr False (so no runtime weight
scale: custom_layers.py:171-195 leaves wscale None), blur_type None, no class conditioning; the hidden nonlinearity is
ReLU, or ``nl='tanh'`` / ``('leaky relu', slope)`` (config.py:208, resnetgan/learner.py:176-181; pinned by
tests/golden/resnet32_tanh.npz).
"""
import torch
import torch.nn.functional as F

from . import step as _step

EPS_NORM = 1e-5     # nn.BatchNorm2d / nn.LayerNorm default (custom_layers.py:100-107 pass no eps)


def _conv(sd, key, x):
    w = sd[key + '.conv2d.weight']
    return F.conv2d(x, w, sd.get(key + '.conv2d.bias'), stride=1, padding=(w.shape[-1] - 1) // 2)


def _bn(sd, key, x, training, buffers):
    """nn.BatchNorm2d: batch statistics in training mode, running estimates updated in `buffers`."""
    rm = buffers[key + '.norm.running_mean'] if buffers is not None else None
    rv = buffers[key + '.norm.running_var'] if buffers is not None else None
    y = F.batch_norm(x, rm, rv, sd[key + '.norm.weight'], sd[key + '.norm.bias'], training=training or rm is None,
                     momentum=0.1, eps=EPS_NORM)
    if training and buffers is not None and key + '.norm.num_batches_tracked' in buffers:
        buffers[key + '.norm.num_batches_tracked'] += 1
    return y


def _ln(sd, key, x):
    w = sd[key + '.norm.weight']
    return F.layer_norm(x, list(w.shape), w, sd[key + '.norm.bias'], eps=EPS_NORM)


def _up(x):
    return F.interpolate(x, scale_factor=2, mode='nearest')


def _pool(x):
    return F.avg_pool2d(x, kernel_size=2, stride=2)


def _nl(nl):
    if nl in (None, 'relu'):
        return F.relu
    if nl == 'tanh':
        return torch.tanh
    if isinstance(nl, (tuple, list)) and nl[0] == 'leaky relu':
        return lambda t: F.leaky_relu(t, negative_slope=float(nl[1]))
    raise ValueError(nl)


def _resblock(sd, p, x, norm, mode, pix32=False, act=F.relu):
    """ResBlock2d (resnetgan/resblocks.py:15-64; ResBlock2d32Pix :67-80).
    mode 'up':   [norm, relu, up, conv] -> [norm, relu, conv];        skip = [up, conv1x1]
    mode 'pool': [norm, relu, conv]     -> [norm, relu, conv, pool];  skip = [pool, conv1x1]
                 (32Pix: skip = [conv1x1, pool])
    mode None:   [norm, relu, conv]     -> [norm, relu, conv];        skip = conv1x1 if present else x"""
    h = act(norm(p + 'conv_layer_1.0', x))
    if mode == 'up':
        h = _conv(sd, p + 'conv_layer_1.3', _up(h))
    else:
        h = _conv(sd, p + 'conv_layer_1.2', h)
    h = act(norm(p + 'conv_layer_2.0', h))
    h = _conv(sd, p + 'conv_layer_2.2', h)
    if mode == 'pool':
        h = _pool(h)
    if mode == 'up':
        s = _conv(sd, p + 'skip_connection.1', _up(x))
    elif mode == 'pool':
        s = _pool(_conv(sd, p + 'skip_connection.0', x)) if pix32 else _conv(sd, p + 'skip_connection.1', _pool(x))
    elif p + 'skip_connection.0.conv2d.weight' in sd:
        s = _conv(sd, p + 'skip_connection.0', x)
    else:
        s = x
    return s + h


def gen_forward(sd, z, res=64, training=True, buffers=None, nl=None):
    """Generator32PixResnet / Generator64PixResnet (resnetgan/architectures.py:29-97)."""
    act = _nl(nl)
    p = 'generator_model.'
    w = sd[p + '1.linear.weight']
    h = F.linear(z.view(-1, w.shape[1]), w, sd[p + '1.linear.bias'])
    h = h.view(z.shape[0], w.shape[0] // 16, 4, 4)
    nblk = 4 if res == 64 else 3
    norm = lambda key, t: _bn(sd, key, t, training, buffers)  # noqa: E731
    for i in range(nblk):
        h = _resblock(sd, f'{p}{3 + i}.', h, norm, 'up', pix32=(res == 32), act=act)
    h = act(norm(f'{p}{3 + nblk}', h))
    return torch.tanh(_conv(sd, f'{p}{5 + nblk}', h))


def disc_forward(sd, x, res=64, nl=None):
    """Discriminator32PixResnet / Discriminator64PixResnet (resnetgan/architectures.py:103-187)."""
    act = _nl(nl)
    norm = lambda key, t: _ln(sd, key, t)  # noqa: E731
    x = x.view(-1, 3, res, res)
    if res == 64:
        h = _conv(sd, 'conv1', x)
        for i in range(4):
            h = _resblock(sd, f'resblocks.{i}.', h, norm, 'pool', act=act)
        h = h.reshape(h.shape[0], -1)
    else:
        # FastResBlock2dDownsample (resblocks.py:83-124): [conv, relu] -> [conv, pool]; skip [pool, conv1x1]
        h1 = act(_conv(sd, 'conv1.conv_layer_1.0', x))
        h = _pool(_conv(sd, 'conv1.conv_layer_2.0', h1)) + _conv(sd, 'conv1.skip_connection.1', _pool(x))
        h = _resblock(sd, 'resblocks.0.', h, norm, 'pool', pix32=True, act=act)
        h = _resblock(sd, 'resblocks.1.', h, norm, None, pix32=True, act=act)
        h = _resblock(sd, 'resblocks.2.', h, norm, None, pix32=True, act=act)
        h = act(h).mean(dim=(2, 3))
    return F.linear(h, sd['linear1.linear.weight'], sd['linear1.linear.bias']).view(-1)


def _is_buffer(k):
    return k.endswith('running_mean') or k.endswith('running_var') or k.endswith('num_batches_tracked')


class ResnetFunctionalGAN:
    """GANLearner.train's loop body (resnetgan/learner.py:538-684): `num_gen_iters` generator
    iterations with the critic frozen, then `num_disc_iters` critic iterations, Adam on both
    (backprop_utils.py:109-120), every random draw passed in.  No drift term, no EWMA."""

    def __init__(self, sd_g, sd_d, res=64, loss='wgan', gp='wgan-gp', lda=10.0, gamma=1.0, lr=1e-4, beta1=0.0,
                 beta2=0.9, adam_eps=1e-8, nl=None):
        self.nl = nl
        self.g = {k: v.detach().clone().requires_grad_(True) for k, v in sd_g.items() if not _is_buffer(k)}
        self.g_buf = {k: v.detach().clone() for k, v in sd_g.items() if _is_buffer(k)}
        self.d = {k: v.detach().clone().requires_grad_(True) for k, v in sd_d.items()}
        self.res, self.loss, self.gp, self.lda, self.gamma = res, loss, gp, lda, gamma
        self.lr, self.b1, self.b2, self.adam_eps = lr, beta1, beta2, adam_eps
        self.st_g = {k: _step.new_adam_state(v) for k, v in self.g.items()}
        self.st_d = {k: _step.new_adam_state(v) for k, v in self.d.items()}

    def gen(self, z, training=True):
        return gen_forward(self.g, z, self.res, training, self.g_buf, self.nl)

    def disc(self, x, sd=None):
        return disc_forward(self.d if sd is None else sd, x, self.res, self.nl)

    def _apply(self, params, states):
        with torch.no_grad():
            for k, p in params.items():
                if p.grad is not None:
                    _step.adam_update(p, p.grad, states[k], self.lr, self.b1, self.b2, self.adam_eps)

    def g_step(self, z):
        for p in self.g.values():
            p.grad = None
        frozen = {k: v.detach() for k, v in self.d.items()}
        out = self.disc(self.gen(z), frozen)
        if self.loss == 'wgan':
            loss = -out.mean()
        elif self.loss == 'nonsaturating':
            loss = F.binary_cross_entropy_with_logits(out, torch.ones_like(out))
        else:   # minimax (:573-578)
            loss = -F.binary_cross_entropy_with_logits(out, torch.zeros_like(out))
        loss.backward()
        self._apply(self.g, self.st_g)
        return loss.detach()

    def d_loss(self, fake, real, eps_interp=None):
        d_fake, d_real = self.disc(fake), self.disc(real)
        if self.loss == 'wgan':
            loss = (d_fake - d_real).mean()
        else:
            loss = F.binary_cross_entropy_with_logits(d_fake, torch.zeros_like(d_fake)) + \
                F.binary_cross_entropy_with_logits(d_real, torch.ones_like(d_real))
        if self.gp is not None:
            loss = loss + _step.calc_gp(self.disc, self.gp, fake, real, self.lda, self.gamma, eps_interp)
        return loss

    def d_step(self, z, real, eps_interp=None):
        for p in self.d.values():
            p.grad = None
        with torch.no_grad():
            fake = self.gen(z)          # generator in train mode: BatchNorm running stats move (:621-622)
        loss = self.d_loss(fake, real, eps_interp)
        loss.backward()
        self._apply(self.d, self.st_d)
        return loss.detach()
