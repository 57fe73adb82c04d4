"""Losses, gradient penalties and the Adam factory on the HIP path (drop-in for
gan_lab/utils/backprop_utils.py; penalty semantics follow the method the train loops actually
call, GANLearner.calc_gp at gan_lab/resnetgan/learner.py:780-827)."""
from functools import partial

import torch

from .. import ops
from .._int import FMAP_SAMPLES


# -- loss functions (backprop_utils.py:19-49) ------------------------------------------------------ #
def wasserstein_distance_gen(outb):
    return -ops.sum_all(outb, 1.0 / outb.numel())


def nonsaturating_loss_gen(outb):
    return ops.bce_logits_mean(outb, 1.0)


def minimax_loss_gen(outb):
    return -ops.bce_logits_mean(outb, 0.0)


def wasserstein_distance_disc(outb, yb):
    """Assumes a pair of real & fake each time: mean(outb - yb)."""
    n = outb.numel()
    return ops.sum_all(outb, 1.0 / n) - ops.sum_all(yb, 1.0 / n)


def minimax_loss_disc(outb, yb):
    return ops.bce_logits_mean(outb, 0.0) + ops.bce_logits_mean(yb, 1.0)


def loss_disc(kind, d_fake, d_real):
    """D adversarial loss as assembled in the train loop (progan/learner.py:791-800)."""
    kind = kind.casefold()
    if kind == 'wgan':
        return wasserstein_distance_disc(d_fake, d_real)
    if kind in ('nonsaturating', 'minimax'):
        return minimax_loss_disc(d_fake, d_real)
    raise ValueError("config does not support this loss.\nCurrently supported Loss Functions are: "
                     "[ 'wgan', 'nonsaturating', 'minimax' ]")


def loss_gen(kind, d_fake):
    """G loss (progan/learner.py:883-896)."""
    kind = kind.casefold()
    if kind == 'wgan':
        return wasserstein_distance_gen(d_fake)
    if kind == 'nonsaturating':
        return nonsaturating_loss_gen(d_fake)
    if kind == 'minimax':
        return minimax_loss_gen(d_fake)
    raise ValueError("config does not support this loss.\nCurrently supported Loss Functions are: "
                     "[ 'wgan', 'nonsaturating', 'minimax' ]")


def drift_loss(d_real, eps_drift):
    """mean(D(real)^2) * eps_drift (progan/learner.py:811-812)."""
    return ops.sumsq_all(d_real, eps_drift / d_real.numel())


# -- gradient regularisers -------------------------------------------------------------------------- #
def calc_gp(nn_disc, gp_type, gen_data, real_data, lda=10., gamma=1., eps_interp=None):
    """R1 / R2 / WGAN-GP penalty with the double backward running through the HIP kernels.

    resnetgan/learner.py:780-827: the gradient norm is taken over the CHANNEL dim only and averaged
    over B*H*W; R1/R2 = mean(|g|_c^2) * lda/2; WGAN-GP (gamma == 1) = mean((|g|_c - 1)^2) * lda/2;
    gamma != 1: mean((|g|_c - gamma)^2 / gamma^2) * lda.  ``eps_interp`` (B,1,1,1) pins the uniform
    draw of :794 (tests); by default it is drawn on the device."""
    gp_type = gp_type.casefold()
    if gp_type in ('wgan-gp', 'r1',):
        real_data = real_data.view(-1, FMAP_SAMPLES, real_data.shape[2], real_data.shape[3])
    if gp_type in ('wgan-gp', 'r2',):
        gen_data = gen_data.view(-1, FMAP_SAMPLES, gen_data.shape[2], gen_data.shape[3])
    if gp_type == 'wgan-gp':
        b = gen_data.shape[0]
        if eps_interp is None:
            eps_interp = torch.rand(b, device=gen_data.device)
        xb = ops.lerp_rows(gen_data.detach(), real_data.detach(), eps_interp.reshape(b).contiguous())
    elif gp_type == 'r1':
        xb = real_data.detach().clone()
    elif gp_type == 'r2':
        xb = gen_data.detach().clone()
    else:
        raise ValueError(f"unsupported gradient penalty '{gp_type}'")
    xb.requires_grad_(True)
    outb = nn_disc(xb)
    return gp_from_output(outb, xb, gp_type, lda, gamma)


def gp_from_output(outb, xb, gp_type, lda=10., gamma=1.):
    """Penalty from an existing D(xb) graph: g = d outb / d xb with create_graph=True, then the
    channel-norm reduction of resnetgan/learner.py:811-825.  Lets the D step evaluate D(real) ONCE and
    use it both for the adversarial/drift terms and for R1 (the reference evaluates the identical
    forward twice, progan/learner.py:789 and resnetgan/learner.py:809)."""
    gp_type = gp_type.casefold()
    ones = torch.ones(outb.shape[0], device=outb.device)
    with ops.input_grad_only():      # only d/dx is wanted here: skip every weight / bias gradient kernel
        outb_grads = torch.autograd.grad(outb, xb, grad_outputs=ones, create_graph=True, retain_graph=True,
                                         only_inputs=True)[0]
    n_pix = outb_grads.numel() // outb_grads.shape[1]          # B*H*W
    if gp_type == 'wgan-gp':
        if gamma != 1.:
            return ops.chnorm_penalty(outb_grads, gamma, lda / (gamma ** 2) / n_pix)
        return ops.chnorm_penalty(outb_grads, 1.0, lda / 2. / n_pix)
    return ops.sumsq_all(outb_grads, lda / 2. / n_pix)


# -- optimiser factory (backprop_utils.py:109-120) -------------------------------------------------- #
def configure_adam_for_gan(lr_base, betas: tuple, eps=1.e-8, wd=0):
    assert isinstance(betas, tuple)
    from ..optim import FusedAdam
    return partial(FusedAdam, lr=lr_base, betas=betas, eps=eps, weight_decay=wd)
