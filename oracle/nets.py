"""Generator / discriminator graph oracle (CPU torch fp32), driven from a reference-layout
``state_dict``.  TEST INFRASTRUCTURE - see oracle/__init__.py.

Reference paths are relative to /root/reference/gan_lab.  The functions take the *same key layout*
the reference modules produce, so a ``state_dict`` of either the reference or the
product modules can be fed in unchanged.
"""
import re

import torch

from . import ops

DEFAULT_CFG = dict(
    blur=True,              # config.blur_type == 'binomial' (config.py default)
    use_noise=True,         # config.use_noise
    use_instancenorm=True,  # config.use_instancenorm
    use_pixelnorm=False,    # StyleGAN default False, ProGAN default True
    normalize_z=True,
    mapping_lrmul=0.01,
    leak=0.2,
    equalized_lr=True,
    mbstd_group_size=4,
    upsample='nearest',     # config.model_upsample_type: 'nearest' | 'bilinear'
    downsample='average',   # config.model_downsample_type: 'average' | 'box' | 'nearest' | 'bilinear'
    align_corners=False,    # config.align_corners (bilinear only)
)


def make_cfg(**kw):
    cfg = dict(DEFAULT_CFG)
    cfg.update(kw)
    return cfg


def _idx_keys(sd, prefix, suffix):
    """Keys ``prefix + <path> + suffix`` sorted by the integers in <path>."""
    out = []
    for k in sd:
        if k.startswith(prefix) and k.endswith(suffix):
            mid = k[len(prefix):len(k) - len(suffix)]
            out.append((tuple(int(t) for t in re.findall(r'\d+', mid)), k))
    out.sort()
    return [k for _, k in out]


def _ws(w, gain, cfg, conv=True):
    if not cfg['equalized_lr']:
        return None
    return ops.conv_wscale(w, gain) if conv else ops.linear_wscale(w, gain)


# ============================================================================================== #
# StyleGAN generator - stylegan/architectures.py:27-59 (mapping), :411-528 (forward)
# ============================================================================================== #
def style_mapping(sd, z, cfg, prefix='z_to_w.'):
    """PixelNorm(z) -> num_fcs x (LinearEx(lrmul) -> LeakyReLU) (stylegan/architectures.py:39-59)."""
    wkeys = _idx_keys(sd, prefix + 'fc_mapping_model.fc_', '.linear.weight')
    x = z.view(-1, sd[wkeys[0]].shape[1])
    if cfg['normalize_z']:
        x = ops.pixelnorm(x)
    for wk in wkeys:
        w = sd[wk]
        b = sd[wk[:-len('weight')] + 'bias']
        x = ops.linear_ex(x, w, b, _ws(w, 2.0, cfg, conv=False), lrmul=cfg['mapping_lrmul'])
        x = ops.lrelu(x, cfg['leak'])
    return x


def _style_layer(sd, n, out, w, noise_n, cfg, first_conv_upsamples):
    """One ``gen_layers[n]`` entry (stylegan/architectures.py:497-526):
    [up -> conv3x3(no bias) -> blur] -> +noise -> +bias -> lrelu -> [PN] -> [IN] -> AdaIN."""
    p = f'gen_layers.{n}.'
    if n > 0:
        ck = _idx_keys(sd, p + '0.', 'conv2d.weight')[0]
        cw = sd[ck]
        cb = sd.get(ck[:-len('weight')] + 'bias')
        up = first_conv_upsamples
        if up:
            out = ops.upsample2(out, cfg['upsample'], cfg['align_corners'])
        out = ops.conv2d_ex(out, cw, cb, _ws(cw, 2.0, cfg), padding=1)
        if up and cfg['blur']:
            out = ops.blur_binomial(out)
    if cfg['use_noise']:
        out = ops.add_noise(out, sd[p + '1.noise_weight'], noise_n)
    bk = p + '2.0.bias'
    if bk in sd:
        out = out + sd[bk]
    out = ops.lrelu(out, cfg['leak'])
    if cfg['use_pixelnorm']:
        out = ops.pixelnorm(out)
    if cfg['use_instancenorm']:
        out = ops.instancenorm(out)
    sw, sb = sd[p + '3.linear.weight'], sd[p + '3.linear.bias']
    y = ops.linear_ex(w, sw, sb, _ws(sw, 1.0, cfg, conv=False))
    return ops.adain_affine(out, y)


def stylegen_num_layers(sd):
    n = 0
    while f'gen_layers.{n}.3.linear.weight' in sd:
        n += 1
    return n


def stylegen_forward(sd, z, noise, cfg, alpha=1.0, fade_in=False, cutoff_idx=None, z_mix=None,
                     return_w=False):
    """StyleGenerator.forward with every random draw made explicit.

    ``noise``: list of (B,1,H,W) tensors, one per gen_layers entry (train mode draws these with
    randn, stylegan/architectures.py:113-116).  ``cutoff_idx``/``z_mix``: the mixing-regularisation
    draw (:415-422, :507-512): layers n >= cutoff_idx... precisely, the style of every layer from
    the one *at* which ``n == cutoff_idx`` fires onward uses w(z_mix).
    """
    L = stylegen_num_layers(sd)
    w = style_mapping(sd, z, cfg)
    w_cur = w
    b = w.shape[0]
    out = sd['const_input'].expand(b, -1, -1, -1)
    pre_fade = None
    for n in range(L):
        if cutoff_idx is not None and n == cutoff_idx:
            w_cur = style_mapping(sd, z_mix, cfg)
        up = (n >= 2 and n % 2 == 0)
        out = _style_layer(sd, n, out, w_cur, None if noise is None else noise[n], cfg, up)
        if fade_in and n == L - 3:
            pre_fade = out
    tw, tb = sd['torgb.conv2d.weight'], sd['torgb.conv2d.bias']
    img = ops.conv2d_ex(out, tw, tb, _ws(tw, 1.0, cfg))
    if fade_in:
        pw, pb = sd['prev_torgb.conv2d.weight'], sd['prev_torgb.conv2d.bias']
        prev = ops.upsample2(ops.conv2d_ex(pre_fade, pw, pb, _ws(pw, 1.0, cfg)))
        img = prev * (1.0 - alpha) + img * alpha  # stylegan/architectures.py:481-487
    return (img, w) if return_w else img


# NOTE on cutoff ordering (stylegan/architectures.py:497-512): inside the loop the reference runs
# conv/noise/bias/norm of layer n FIRST and then, ``if n == cutoff_idx``, swaps w before computing
# the style of that same layer n.  Only the style (layer[3]) consumes w, so swapping before the
# layer body (as above) is equivalent.


# ============================================================================================== #
# ProGAN generator - progan/architectures.py:31-167
# ============================================================================================== #
def progen_num_blocks(sd):
    n = 0
    while any(k.startswith(f'gen_blocks.{n}.') for k in sd):
        n += 1
    return n


def _pn(x, cfg):
    return ops.pixelnorm(x) if cfg['use_pixelnorm'] else x


def progen_forward(sd, z, cfg, alpha=1.0, fade_in=False):
    nb = progen_num_blocks(sd)
    lw, lb = sd['gen_blocks.0.0.linear.weight'], sd['gen_blocks.0.0.linear.bias']
    x = z.view(-1, lw.shape[1])
    if cfg['normalize_z']:
        x = ops.pixelnorm(x)                                               # :69-73
    x = ops.linear_ex(x, lw, lb, _ws(lw, 2.0 / 16, cfg, conv=False))      # :80-83
    x = x.view(x.shape[0], -1, 4, 4)
    x = _pn(ops.lrelu(x, cfg['leak']), cfg)
    ck = _idx_keys(sd, 'gen_blocks.0.', 'conv2d.weight')[0]
    cw, cb = sd[ck], sd[ck[:-len('weight')] + 'bias']
    x = _pn(ops.lrelu(ops.conv2d_ex(x, cw, cb, _ws(cw, 2.0, cfg), padding=1), cfg['leak']), cfg)
    pre_fade = x
    for i in range(1, nb):
        pre_fade = x
        p0, p1 = f'gen_blocks.{i}.0.', f'gen_blocks.{i}.1.'
        ck = _idx_keys(sd, p0, 'conv2d.weight')[0]
        cw, cb = sd[ck], sd.get(ck[:-len('weight')] + 'bias')
        x = ops.upsample2(x, cfg['upsample'], cfg['align_corners'])
        x = ops.conv2d_ex(x, cw, cb, _ws(cw, 2.0, cfg), padding=1)
        if cfg['blur']:
            x = ops.blur_binomial(x)
            bk = [k for k in _idx_keys(sd, p0, '.bias') if 'conv2d' not in k]
            x = x + sd[bk[0]]
        x = _pn(ops.lrelu(x, cfg['leak']), cfg)
        ck = _idx_keys(sd, p1, 'conv2d.weight')[0]
        cw, cb = sd[ck], sd[ck[:-len('weight')] + 'bias']
        x = _pn(ops.lrelu(ops.conv2d_ex(x, cw, cb, _ws(cw, 2.0, cfg), padding=1), cfg['leak']), cfg)
    tw, tb = sd['torgb.conv2d.weight'], sd['torgb.conv2d.bias']
    img = ops.conv2d_ex(x, tw, tb, _ws(tw, 1.0, cfg))
    if fade_in:
        pw, pb = sd['prev_torgb.conv2d.weight'], sd['prev_torgb.conv2d.bias']
        prev = ops.upsample2(ops.conv2d_ex(pre_fade, pw, pb, _ws(pw, 1.0, cfg)))
        img = prev * (1.0 - alpha) + img * alpha                           # :163-165
    return img


# ============================================================================================== #
# ProGAN / StyleGAN discriminator - progan/architectures.py:175-318
# ============================================================================================== #
def disc_num_blocks(sd):
    n = 0
    while any(k.startswith(f'disc_blocks.{n}.') for k in sd):
        n += 1
    return n


def _fromrgb(sd, x, cfg, prefix):
    w, b = sd[prefix + '0.conv2d.weight'], sd[prefix + '0.conv2d.bias']
    return ops.lrelu(ops.conv2d_ex(x, w, b, _ws(w, 2.0, cfg)), cfg['leak'])   # :286-292


def _disc_block(sd, i, x, cfg):
    """conv3x3+b, lrelu ; [blur], conv3x3 (no bias), avgpool2, +bias, lrelu  (:254-284)."""
    p0, p1 = f'disc_blocks.{i}.0.', f'disc_blocks.{i}.1.'
    ck = _idx_keys(sd, p0, 'conv2d.weight')[0]
    cw, cb = sd[ck], sd[ck[:-len('weight')] + 'bias']
    x = ops.lrelu(ops.conv2d_ex(x, cw, cb, _ws(cw, 2.0, cfg), padding=1), cfg['leak'])
    if cfg['blur']:
        x = ops.blur_binomial(x)
    ck = _idx_keys(sd, p1, 'conv2d.weight')[0]
    cw = sd[ck]
    x = ops.conv2d_ex(x, cw, None, _ws(cw, 2.0, cfg), padding=1)
    x = ops.pool2(x, cfg['downsample'], cfg['align_corners'])
    bk = [k for k in _idx_keys(sd, p1, '.bias') if 'conv2d' not in k]
    x = x + sd[bk[0]]
    return ops.lrelu(x, cfg['leak'])


def _disc_final(sd, i, x, cfg):
    """mbstd -> conv3x3+b -> lrelu -> conv4x4 valid +b -> lrelu -> flatten -> linear (:219-233)."""
    p = f'disc_blocks.{i}.'
    cks = _idx_keys(sd, p, 'conv2d.weight')
    if cfg['mbstd_group_size'] != -1:
        x = ops.mbstd_concat(x, cfg['mbstd_group_size'])
    w3, b3 = sd[cks[0]], sd[cks[0][:-len('weight')] + 'bias']
    x = ops.lrelu(ops.conv2d_ex(x, w3, b3, _ws(w3, 2.0, cfg), padding=1), cfg['leak'])
    w4, b4 = sd[cks[1]], sd[cks[1][:-len('weight')] + 'bias']
    x = ops.lrelu(ops.conv2d_ex(x, w4, b4, _ws(w4, 2.0, cfg), padding=0), cfg['leak'])
    x = x.view(x.shape[0], -1)
    lk = _idx_keys(sd, p, 'linear.weight')[0]
    lw, lb = sd[lk], sd[lk[:-len('weight')] + 'bias']
    return ops.linear_ex(x, lw, lb, _ws(lw, 1.0, cfg, conv=False))


def disc_forward(sd, x, cfg, alpha=1.0, fade_in=False):
    nb = disc_num_blocks(sd)
    res = 4 * 2 ** (nb - 1)
    x = x.view(-1, 3, res, res)

    def block(i, t):
        return _disc_final(sd, i, t, cfg) if i == nb - 1 else _disc_block(sd, i, t, cfg)

    h = block(0, _fromrgb(sd, x, cfg, 'fromrgb.'))
    if fade_in:
        skip = _fromrgb(sd, ops.avgpool2(x), cfg, 'prev_fromrgb.')
        h = skip * (1.0 - alpha) + h * alpha                                # :311-313
    for i in range(1, nb):
        h = block(i, h)
    return h.view(-1)
