"""StyleGAN generator on the HIP kernels (drop-in for gan_lab/stylegan/architectures.py).

Module tree, constructor arguments and ``state_dict`` keys are those of the reference
(stylegan/architectures.py:27-59, :105-119, :126-409) so checkpoints interchange; ``forward``
(:411-528) is re-expressed over fused kernels:

    gen_layers[n] = [ (Upsample ->) Conv2dEx (-> blur) | StyleAddNoise | (bias, lrelu, norms) | style FC ]
      -> conv kernel with the nearest-2x upsample folded into its tile load      (K1/K2)
      -> blur kernel                                                              (K3)
      -> ONE pass: + noise_w*noise + bias, LeakyReLU                              (K4+K5)
      -> InstanceNorm statistics + ONE pass: normalise and apply (ys+1, yb)       (K6+K7)
"""
import copy
import types

import numpy as np
import torch
from torch import nn

from .. import ops, rng
from .._int import FMAP_SAMPLES, RES_INIT
from ..utils.custom_layers import (Blur2d, Conv2dBias, Conv2dEx, Lambda, LeakyReLU, LinearEx, NormalizeLayer, Tanh, Upsample2x,
                                   fused_sequential, get_blur_op, own_resampler)
from ..utils.latent_utils import gen_rand_latent_vars
from .base import StyleGAN

FMAP_G_INIT_FCTR = 1
IN_EPS = 1e-8     # NormalizeLayer('InstanceNorm') epsilon (custom_layers.py:98-99)


class StyleMappingNetwork(nn.Module):
    """z -> w: PixelNorm, then num_fcs x (eq-LR FC with lrmul -> LeakyReLU); each FC+bias+LeakyReLU is
    one MFMA kernel launch."""

    def __init__(self, len_latent=512, len_dlatent=512, num_fcs=8, lrmul=.01, nl=None, equalized_lr=True,
                 normalize_z=True):
        super().__init__()
        nl = nl if nl is not None else LeakyReLU(.2)
        self.len_latent = len_latent
        if normalize_z:
            self.preprocess_z = nn.Sequential(Lambda(lambda x: x.view(-1, len_latent)), NormalizeLayer('PixelNorm'))
        else:
            self.preprocess_z = Lambda(lambda x: x.view(-1, len_latent))
        self.dims = np.linspace(len_latent, len_dlatent, num_fcs + 1).astype(np.int64)
        self.fc_mapping_model = nn.Sequential()
        for seq_n in range(num_fcs):
            self.fc_mapping_model.add_module(
                'fc_' + str(seq_n),
                LinearEx(nin_feat=self.dims[seq_n], nout_feat=self.dims[seq_n + 1], init='He', init_type='StyleGAN',
                         gain_sq_base=2., equalized_lr=equalized_lr, lrmul=lrmul))
            self.fc_mapping_model.add_module('nl_' + str(seq_n), nl)

    def forward(self, x):
        return fused_sequential([self.fc_mapping_model], self.preprocess_z(x))


class StyleAddNoise(nn.Module):
    """x + noise_weight * N(0,1)(B,1,H,W).  In training mode a user-supplied ``noise`` is ignored
    (stylegan/architectures.py:113) unless ``honour_noise_in_training`` is set (parity tests)."""
    honour_noise_in_training = False

    def __init__(self, nf):
        super().__init__()
        self.noise_weight = nn.Parameter(torch.zeros(1, nf, 1, 1))

    def draw(self, x, noise=None):
        if noise is not None and (not self.training or StyleAddNoise.honour_noise_in_training):
            if noise.shape[0] == 1 and x.shape[0] > 1:      # torch broadcasting of the reference's x + w * noise
                noise = noise.expand(x.shape[0], -1, -1, -1)
            return noise
        return rng.randn((x.shape[0], 1, x.shape[2], x.shape[3]), x.device)

    def forward(self, x, noise=None):
        return ops.bias_act(x, None, self.draw(x, noise), self.noise_weight)


class StyleGenerator(StyleGAN):
    """StyleGAN (Karras et al. 2019) generator; see the module docstring for the kernel mapping."""

    def __init__(self, final_res, latent_distribution='normal', len_latent=512, len_dlatent=512,
                 mapping_num_fcs=8, mapping_lrmul=.01, use_instancenorm=True, use_noise=True, upsampler=None,
                 blur_type=None, nl=None, num_classes=0, equalized_lr=True, normalize_z=True, use_pixelnorm=False,
                 pct_mixing_reg=.9, truncation_trick_params={'beta': .995, 'psi': .7, 'cutoff_stage': 4}):
        super().__init__(final_res)
        if num_classes:
            raise NotImplementedError('class-conditioned mapping network: SURVEY.md §8f item 4 (next)')
        self.gen_layers = nn.ModuleList()
        self.upsampler = own_resampler(upsampler) if upsampler is not None else Upsample2x()
        self.gen_blur_type = blur_type
        self.nl = nl if nl is not None else LeakyReLU(.2)
        self.equalized_lr = equalized_lr
        self.pct_mixing_reg = pct_mixing_reg
        self._use_mixing_reg = True if pct_mixing_reg else False
        self.latent_distribution = latent_distribution
        self.len_latent, self.len_dlatent = len_latent, len_dlatent
        self.num_classes = num_classes
        self.z_to_w = StyleMappingNetwork(len_latent=len_latent, len_dlatent=len_dlatent, num_fcs=mapping_num_fcs,
                                          lrmul=mapping_lrmul, nl=self.nl, equalized_lr=equalized_lr,
                                          normalize_z=normalize_z)
        _fmap_init = len_latent * FMAP_G_INIT_FCTR
        self.const_input = nn.Parameter(torch.ones(1, _fmap_init, RES_INIT, RES_INIT))
        self._use_noise = use_noise
        self._trained_with_noise = use_noise
        self.use_pixelnorm, self.use_instancenorm = use_pixelnorm, use_instancenorm

        conv = Conv2dEx(ni=_fmap_init, nf=self.fmap, ks=3, stride=1, padding=1, init='He', init_type='StyleGAN',
                        gain_sq_base=2., equalized_lr=equalized_lr, include_bias=not use_noise)
        if use_noise:
            noise = [StyleAddNoise(nf=_fmap_init), StyleAddNoise(nf=self.fmap)]
            bias = ([Conv2dBias(nf=_fmap_init)], [Conv2dBias(nf=self.fmap)],)
        else:
            noise = [None, None]
            bias = ([], [],)
        w_to_styles = tuple(
            LinearEx(nin_feat=self.z_to_w.dims[-1], nout_feat=2 * nf, init='He', init_type='StyleGAN',
                     gain_sq_base=1., equalized_lr=equalized_lr) for nf in (_fmap_init, self.fmap))
        assert 0. <= truncation_trick_params['beta'] <= 1.
        self.w_ewma_beta = truncation_trick_params['beta']
        self._w_eval_psi = truncation_trick_params['psi']
        cs = truncation_trick_params['cutoff_stage']
        assert (isinstance(cs, int) and 0 < cs <= int(np.log2(self.final_res)) - 2) or cs is None
        self._trunc_cutoff_stage = cs
        self.use_truncation_trick = True if cs else False
        self.w_ewma = None

        self.gen_layers.append(nn.ModuleList([None, noise[0], nn.Sequential(*bias[0], self.nl, *self._norms()),
                                              w_to_styles[0]]))
        self.gen_layers.append(nn.ModuleList([conv, noise[1], nn.Sequential(*bias[1], self.nl, *self._norms()),
                                              w_to_styles[1]]))
        self.prev_torgb = None
        self._update_torgb(ni=self.fmap)

    def _norms(self):
        norms = []
        if self.use_pixelnorm:
            norms.append(NormalizeLayer('PixelNorm'))
        if self.use_instancenorm:
            norms.append(NormalizeLayer('InstanceNorm'))
        return norms

    # -- growth (stylegan/architectures.py:260-290) ------------------------------------------------
    def increase_scale(self):
        if not self.scale_inc_metadata_updated:
            super().increase_scale()
        else:
            self.scale_inc_metadata_updated = False
        blur_op = get_blur_op(self.gen_blur_type, self.fmap) if self.gen_blur_type is not None else None
        dev = self.const_input.device
        self.gen_layers.append(self.get_conv_layer(ni=self.fmap_prev, upsample=True, blur_op=blur_op).to(dev))
        self.gen_layers.append(self.get_conv_layer(ni=self.fmap).to(dev))
        self.prev_torgb = copy.deepcopy(self.torgb)
        self._update_torgb(ni=self.fmap)
        self.torgb.to(dev)

    def get_conv_layer(self, ni, upsample=False, blur_op=None, append_nl=True):
        upsampler = [self.upsampler] if upsample else []
        own_bias = not (self._trained_with_noise or blur_op is not None)
        conv = Conv2dEx(ni=ni, nf=self.fmap, ks=3, stride=1, padding=1, init='He', init_type='StyleGAN',
                        gain_sq_base=2., equalized_lr=self.equalized_lr, include_bias=own_bias)
        bias = [] if own_bias else [Conv2dBias(nf=self.fmap)]
        blur = [blur_op] if blur_op is not None else []
        noise = StyleAddNoise(nf=self.fmap) if self._trained_with_noise else None
        nl = [self.nl] if append_nl else []
        w_to_style = LinearEx(nin_feat=self.z_to_w.dims[-1], nout_feat=2 * self.fmap, init='He',
                              init_type='StyleGAN', gain_sq_base=1., equalized_lr=self.equalized_lr)
        return nn.ModuleList([nn.Sequential(*upsampler, conv, *blur), noise,
                              nn.Sequential(*(bias + nl + self._norms())), w_to_style])

    def _update_torgb(self, ni):
        self.torgb = Conv2dEx(ni=ni, nf=FMAP_SAMPLES, ks=1, stride=1, padding=0, init='He', init_type='StyleGAN',
                              gain_sq_base=1., equalized_lr=self.equalized_lr)

    # -- mode switches (stylegan/architectures.py:343-409) ------------------------------------------
    def train(self, mode=True):
        super().train(mode=mode)
        self._use_noise = self._trained_with_noise
        self._use_mixing_reg = True if (self.pct_mixing_reg and mode) else False
        return self

    def eval(self):
        super().eval()
        self._use_mixing_reg = False
        return self

    def to(self, *args, **kwargs):
        super().to(*args, **kwargs)
        for arg in args:
            if arg in ('cpu', 'cuda',) or isinstance(arg, torch.device):
                if self.w_ewma is not None:
                    self.w_ewma = self.w_ewma.to(arg)
                    break
        return self

    @property
    def use_noise(self):
        return self._use_noise

    @use_noise.setter
    def use_noise(self, mode):
        if self.training:
            raise Exception('Once use_noise argument is set, it cannot be changed for training purposes. '
                            'It can, however, be changed in eval mode.')
        elif not self._trained_with_noise:
            raise Exception('Model was not trained with noise, so cannot use noise in eval mode.')
        self._use_noise = mode

    @property
    def w_eval_psi(self):
        return self._w_eval_psi

    @w_eval_psi.setter
    def w_eval_psi(self, new_w_eval_psi):
        if self.training:
            raise Exception('Can only alter psi value for truncation trick on w during evaluation mode.')
        self._w_eval_psi = new_w_eval_psi

    @property
    def trunc_cutoff_stage(self):
        return self._trunc_cutoff_stage

    @trunc_cutoff_stage.setter
    def trunc_cutoff_stage(self, new_stage):
        if self.training:
            raise Exception('Can only alter cutoff stage for truncation trick on w during evaluation mode.')
        final_stage = int(np.log2(self.final_res)) - 1
        if (isinstance(new_stage, int) and 0 < new_stage <= final_stage) or new_stage is None:
            self._trunc_cutoff_stage = new_stage
        else:
            raise ValueError(f'Input cutoff stage for truncation trick on w must be of type `int` in range '
                             f'(0,{final_stage}] or `None`.')

    # -- forward -------------------------------------------------------------------------------------
    def _layer(self, n, layer, out, w, noise, consumer=None):
        """One gen_layers entry on fused kernels.  ``out`` may be an ``ops.Deferred`` (the previous layer's output with its
        InstanceNorm + style not yet applied: this layer's convolution applies it while it stages its input, csrc/mod.hip);
        ``consumer``: what reads THIS layer's output - ('conv' | 'upconv' | 'torgb', module) or None - so that the output
        can stay deferred where that reader has an affine-on-load kernel for its shape."""
        blur = False
        mods = list(layer[2])
        bias = mods.pop(0) if mods and isinstance(mods[0], Conv2dBias) else None
        act = mods.pop(0) if mods and isinstance(mods[0], LeakyReLU) else None
        bias_t = bias.bias if bias is not None else None
        bias_scale = (bias.lrmul if bias.use_lrmul else 1.0) if bias is not None else 1.0
        if mods and isinstance(mods[0], Tanh):
            # --nonlinearity tanh (config.py:254): no fused kernels for it - the layer composed from its parts
            out = ops.materialize(out)
            if n:
                out = fused_sequential(list(layer[0]) if isinstance(layer[0], nn.Sequential) else [layer[0]], out)
            nz = layer[1].draw(out, noise[n] if noise is not None else None) if self.use_noise else None
            nw = layer[1].noise_weight if nz is not None else None
            out = ops.tanh(ops.bias_act(out, bias_t, nz, nw, bias_scale=bias_scale, act=None))
            if self.use_pixelnorm:
                out = ops.pixelnorm(out)
            y = layer[3](w)
            return ops.instnorm_style(out, y, IN_EPS) if self.use_instancenorm else ops.style_mod(out, y)
        name = 'lrelu' if act is not None else None
        slope = act.negative_slope if act is not None else 0.2
        y = layer[3](w)                                                # (B, 2C) style
        plain_in = self.use_instancenorm and not self.use_pixelnorm
        head = []
        if n:
            head = list(layer[0]) if isinstance(layer[0], nn.Sequential) else [layer[0]]
            blur = bool(head) and isinstance(head[-1], Blur2d)
            body = head[:-1] if blur else head
            up = len(body) == 2 and isinstance(body[0], Upsample2x)
            conv = body[-1] if (body and isinstance(body[-1], Conv2dEx) and len(body) == (2 if up else 1)) else None
            src = out.a if isinstance(out, ops.Deferred) else out
            if up and blur and conv is not None and conv.conv2d.bias is None and plain_in and \
                    ops.upconv_blur_tail_ok(src.shape, conv.conv2d.weight, isinstance(out, ops.Deferred), conv.padding):
                # the layer that opens a resolution in ONE pass: upsample + conv + blur + noise + bias + LeakyReLU + the
                # InstanceNorm statistics (thin transposed stride-2 kernel with the blur folded in), deferred in and out
                like = types.SimpleNamespace(shape=(src.shape[0], 1, 2 * src.shape[2], 2 * src.shape[3]), device=src.device)
                nz = layer[1].draw(like, noise[n] if noise is not None else None) if self.use_noise else None
                nw = layer[1].noise_weight if nz is not None else None
                d = ops.upconv_blur_tail(out, conv.conv2d.weight, conv.scale, bias_t, nz, nw, y, bias_scale=bias_scale,
                                         act=name, slope=slope, eps=IN_EPS)
                return d if self._defer_ok(d.a.shape, consumer) else ops.materialize(d)
            if isinstance(out, ops.Deferred) and conv is not None and conv.conv2d.bias is None:
                if not up and not blur and bias is not None and plain_in and \
                        ops.mod_conv_ok(out, conv.conv2d.weight, conv.padding):
                    # thin plain 3x3 layer, deferred in -> deferred out in ONE pass over the activations (affine on load,
                    # noise + bias + LeakyReLU + InstanceNorm statistics in the conv's epilogue)
                    nz = layer[1].draw(out.a[:, :1], noise[n] if noise is not None else None) if self.use_noise else None
                    nw = layer[1].noise_weight if nz is not None else None
                    d = ops.conv_mod_tail(out, conv.conv2d.weight, conv.scale, bias_t, nz, nw, y, bias_scale=bias_scale,
                                          act=name, slope=slope, eps=IN_EPS)
                    return d if self._defer_ok(d.a.shape, consumer) else ops.materialize(d)
                if ops.conv_aff_ok(out.a.shape, conv.conv2d.weight, up, conv.padding):
                    out = ops.conv_aff(out, conv.conv2d.weight, conv.scale, up=up)     # (up+)conv with the affine on load
                else:
                    out = fused_sequential(body, ops.materialize(out))
            else:
                out = fused_sequential(body, ops.materialize(out))    # (up+)conv MFMA kernel
        else:
            out = ops.materialize(out)
        nz = layer[1].draw(out, noise[n] if noise is not None else None) if self.use_noise else None
        nw = layer[1].noise_weight if nz is not None else None
        if plain_in:
            if ops.deferrable(out) and self._defer_ok(out.shape, consumer):
                # blur + noise + bias + LeakyReLU + statistics in one pass; the normalisation is left to the consumer
                return ops.layer_tail_deferred(out, bias_t, nz, nw, y, bias_scale=bias_scale, act=name, slope=slope,
                                               blur=blur, eps=IN_EPS)
            # blur + noise + bias + LeakyReLU (+ the InstanceNorm statistics) in one pass, IN + (ys+1, yb) in a second
            return ops.layer_tail(out, bias_t, nz, nw, y, bias_scale=bias_scale, act=name, slope=slope, blur=blur,
                                  eps=IN_EPS)
        out = ops.bias_act(out, bias_t, nz, nw, bias_scale=bias_scale, act=name, slope=slope, blur=blur)
        if self.use_pixelnorm:
            out = ops.pixelnorm(out)
        # use_instancenorm=False: the style is applied to the un-normalised activations
        return ops.instnorm_style(out, y, IN_EPS) if self.use_instancenorm else ops.style_mod(out, y)

    def _consumer(self, n, L):
        """Who reads layer n's output: ('conv' | 'upconv', Conv2dEx) for the next layer's (upsample +) 3x3 conv, ('torgb',
        Conv2dEx) behind the last layer; None where the tensor has a second reader (prev_torgb while fading in) or the next
        layer is not of that shape."""
        if self.fade_in_phase and n == L - 3:
            return None
        if n == L - 1:
            return ('torgb', self.torgb)
        nxt = self.gen_layers[n + 1][0]
        head = list(nxt) if isinstance(nxt, nn.Sequential) else [nxt]
        if head and isinstance(head[-1], Blur2d):
            head = head[:-1]
        if len(head) == 1 and isinstance(head[0], Conv2dEx):
            return ('conv', head[0])
        if len(head) == 2 and isinstance(head[0], Upsample2x) and isinstance(head[1], Conv2dEx):
            return ('upconv', head[1])
        return None

    @staticmethod
    def _defer_ok(shape, consumer):
        """Can ``consumer`` read a deferred tensor of ``shape``?"""
        if consumer is None:
            return False
        kind, mod = consumer
        n, c, h, w = (int(v) for v in shape)
        if mod.conv2d.bias is not None and kind != 'torgb':
            return False
        if kind == 'torgb':
            return tuple(mod.conv2d.weight.shape[1:]) == (c, 1, 1) and mod.conv2d.weight.shape[0] <= 4 and c <= 16 and \
                (h * w) % 4 == 0
        if kind == 'conv':
            return ops.mod_conv_shape_ok(shape, mod.conv2d.weight, mod.padding) or \
                ops.conv_aff_ok(shape, mod.conv2d.weight, False, mod.padding)
        return ops.conv_aff_ok(shape, mod.conv2d.weight, True, mod.padding)

    def _new_w(self, bs, dev):
        z2 = gen_rand_latent_vars(num_samples=bs, length=self.len_latent, distribution=self.latent_distribution,
                                  device=dev)
        return self.z_to_w(z2)

    def draw_mixing_cutoff(self):
        """The host coin + layer index of the mixing regularisation (:415-422): None = this forward is not mixed."""
        if self._use_mixing_reg and np.random.rand() < self.pct_mixing_reg:
            hi = 2 * self.scale_stage if self.alpha != 0 else 2 * self.scale_stage - 2
            return torch.randint(1, hi, (1,)).item()
        return None

    def forward(self, x, x_mixing=None, style_mixing_stage: int = None, noise=None, _mix=None):
        """``_mix=(cutoff_idx, z_mix)`` pins the mixing-regularisation draw (tests); otherwise it is
        drawn like the reference does (:415-422)."""
        cutoff_idx, z_mix = None, None
        if _mix is not None:
            cutoff_idx, z_mix = _mix
        else:
            cutoff_idx = self.draw_mixing_cutoff()
        w = self.z_to_w(x)
        bs = w.shape[0]
        if self.use_truncation_trick:
            if self.training:
                with torch.no_grad():  # running average of w for the eval-time truncation trick (:427-437)
                    wm = ops.k_channel_sum(w.detach(), None, 1.0 / bs)          # mean over the batch (HIP kernel)
                    if self.w_ewma is None:
                        self.w_ewma = wm
                    else:     # in place: the same buffer every step (a replayed step graph reads what the last one wrote)
                        ops.k_axpby(wm, self.w_ewma, 1. - self.w_ewma_beta, self.w_ewma_beta, out=self.w_ewma)
            elif self.trunc_cutoff_stage is not None:
                w = self.w_ewma.expand_as(w) + self.w_eval_psi * (w - self.w_ewma.expand_as(w))
        out = self.const_input.expand(bs, -1, -1, -1)
        L = len(self.gen_layers)
        pre_fade = None
        for n, layer in enumerate(self.gen_layers):
            if n == cutoff_idx:
                w = self.z_to_w(z_mix) if z_mix is not None else self._new_w(bs, w.device)
            if not self.fade_in_phase:
                if n == style_mixing_stage:
                    assert (style_mixing_stage and not self.training and isinstance(x_mixing, torch.Tensor))
                    w = self.z_to_w(x_mixing)
                    if self.use_truncation_trick and self.trunc_cutoff_stage is not None and \
                            n < 2 * self.trunc_cutoff_stage:
                        w = self.w_ewma.expand_as(w) + self.w_eval_psi * (w - self.w_ewma.expand_as(w))
                elif self.use_truncation_trick and not self.training and self.trunc_cutoff_stage is not None and \
                        n == 2 * self.trunc_cutoff_stage:
                    w = (w - self.w_ewma.expand_as(w)).div(self.w_eval_psi) + self.w_ewma.expand_as(w)
            out = self._layer(n, layer, out, w, noise, consumer=self._consumer(n, L))
            if self.fade_in_phase and n == L - 3:
                pre_fade = out
        if isinstance(out, ops.Deferred) and ops.torgb_mod_ok(out, self.torgb.conv2d.weight):
            t = self.torgb                    # toRGB is linear: per-sample weights w*s and bias b + w.t (csrc/mod.hip)
            img = ops.torgb_mod(out, t.conv2d.weight, t.conv2d.bias, t.scale, t.lrmul if t.use_lrmul else 1.0)
        else:
            img = self.torgb(ops.materialize(out))
        if self.fade_in_phase:
            prev = ops.upsample2(self.prev_torgb(pre_fade))
            img = ops.lerp(prev, img, self.alpha)                      # (:481-494)
        return img
