"""Training-step oracle (CPU torch fp32): gradient penalties, D-step, G-step, Adam, EWMA.
TEST INFRASTRUCTURE - see oracle/__init__.py.  Reference paths relative to /root/reference/gan_lab.
"""
import math

import torch

from . import nets, ops


# -- resnetgan/learner.py:780-827 --------------------------------------------------------------- #
def calc_gp(disc_fn, kind, fake, real, lda=10.0, gamma=1.0, eps_interp=None):
    """``GANLearner.calc_gp`` - the method every train loop calls (NOT backprop_utils.calc_gp).

    R1/R2: ``(g.norm(2, dim=1)**2).mean() * lda / 2`` - the norm is over the CHANNEL dim only and
    the mean runs over B*H*W (:825).  WGAN-GP with gamma == 1: ``((|g|_c - 1)**2).mean()*lda/2``
    (:823, note the /2); gamma != 1: ``((|g|_c - gamma)**2 / gamma**2).mean() * lda`` (:820).
    ``eps_interp``: the (B,1,1,1) uniform draw of :794 made explicit.
    """
    kind = kind.casefold()
    if kind == 'wgan-gp':
        xb = eps_interp * fake.detach() + (1 - eps_interp) * real.detach()
    elif kind == 'r1':
        xb = real.detach().clone()
    elif kind == 'r2':
        xb = fake.detach().clone()
    else:
        raise ValueError(kind)
    xb.requires_grad_(True)
    outb = disc_fn(xb)
    g = torch.autograd.grad(outb, xb, grad_outputs=torch.ones_like(outb),
                            create_graph=True, retain_graph=True, only_inputs=True)[0]
    if kind == 'wgan-gp':
        if gamma != 1.0:
            return ((g.norm(2, dim=1) - gamma) ** 2 / gamma ** 2).mean() * lda
        return ((g.norm(2, dim=1) - 1.0) ** 2).mean() * lda / 2.0
    return (g.norm(2, dim=1) ** 2).mean() * lda / 2.0


# -- progan/learner.py:734-816 ------------------------------------------------------------------ #
def d_loss(sd_d, cfg, fake, real, loss='nonsaturating', gp='r1', lda=10.0, gamma=1.0,
           eps_drift=0.001, alpha=1.0, fade_in=False, eps_interp=None, return_parts=False):
    """D-step loss: adversarial + gradient penalty (:808-809) + drift (:811-812)."""
    def D(x):
        return nets.disc_forward(sd_d, x, cfg, alpha=alpha, fade_in=fade_in)
    d_fake, d_real = D(fake), D(real)
    adv = ops.loss_disc(loss, d_fake, d_real)
    total = adv
    gpv = None
    if gp is not None:
        gpv = calc_gp(D, gp, fake, real, lda, gamma, eps_interp)
        total = total + gpv
    if eps_drift > 0:
        total = total + (d_real ** 2).mean() * eps_drift
    if return_parts:
        return total, dict(adv=adv, gp=gpv, d_fake=d_fake, d_real=d_real)
    return total


# -- torch.optim.Adam as configured by backprop_utils.py:109-120 --------------------------------- #
def adam_update(p, g, state, lr, beta1=0.0, beta2=0.99, eps=1e-8, wd=0.0):
    """Single-tensor torch.optim.Adam step (no amsgrad, L2 weight decay folded into the grad).
    ``state``: dict(step, exp_avg, exp_avg_sq); updated in place, ``p`` updated in place."""
    if wd != 0.0:
        g = g + wd * p
    state['step'] += 1
    t = state['step']
    state['exp_avg'].mul_(beta1).add_(g, alpha=1 - beta1)
    state['exp_avg_sq'].mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** t
    bc2 = 1 - beta2 ** t
    denom = (state['exp_avg_sq'].sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(state['exp_avg'], denom, value=-lr / bc1)


def new_adam_state(p):
    return dict(step=0, exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p))


# -- progan/learner.py:909-916, :1124-1127 ------------------------------------------------------ #
def ewma_beta(batch_size, gen_bs_mult=1, half_life=10.0):
    return 0.5 ** ((batch_size * gen_bs_mult) / (half_life * 1000.0)) if half_life > 0 else 0.0


def ewma_update(lagged, p, beta):
    """lagged = p*(1-beta) + lagged*beta."""
    return p * (1.0 - beta) + lagged * beta


# -- progan/learner.py:653 ---------------------------------------------------------------------- #
def delta_alpha(batch_size, nimg_transition, num_disc_iters=1):
    return batch_size / ((nimg_transition / num_disc_iters) - batch_size)


def round_nimg_transition(nimg_transition, batch_size):
    """progan/learner.py:451-455, :646-649."""
    if nimg_transition % batch_size != 0:
        return batch_size * (int(nimg_transition / batch_size) + 1)
    return nimg_transition


class FunctionalGAN:
    """A whole G+D training step (1 D-iter + 1 G-iter, progan/learner.py:734-943) on leaf-tensor
    state dicts, with every random draw passed in.  Used as the parity checker for the product
    learner step and as ``bench.py``'s ``cpu_baseline`` (kind "port")."""

    def __init__(self, sd_g, sd_d, cfg, model='stylegan', loss='nonsaturating', gp='r1', lda=10.0,
                 gamma=1.0, eps_drift=0.001, lr=1e-3, beta1=0.0, beta2=0.99, adam_eps=1e-8,
                 excluded=('prev_torgb.conv2d.weight', 'prev_torgb.conv2d.bias',
                           'prev_fromrgb.0.conv2d.weight', 'prev_fromrgb.0.conv2d.bias')):
        self.g = {k: v.detach().clone().requires_grad_(True) for k, v in sd_g.items()}
        self.d = {k: v.detach().clone().requires_grad_(True) for k, v in sd_d.items()}
        self.cfg, self.model = cfg, model
        self.loss, self.gp, self.lda, self.gamma, self.eps_drift = loss, gp, lda, gamma, eps_drift
        self.lr, self.b1, self.b2, self.adam_eps = lr, beta1, beta2, adam_eps
        self.excluded = set(excluded)
        self.st_g = {k: new_adam_state(v) for k, v in self.g.items()}
        self.st_d = {k: new_adam_state(v) for k, v in self.d.items()}
        self.lagged = {k: v.detach().clone() for k, v in self.g.items()}

    def gen(self, z, noise=None, alpha=1.0, fade_in=False, cutoff_idx=None, z_mix=None):
        if self.model == 'stylegan':
            return nets.stylegen_forward(self.g, z, noise, self.cfg, alpha, fade_in, cutoff_idx, z_mix)
        return nets.progen_forward(self.g, z, self.cfg, alpha, fade_in)

    def _apply(self, params, states, fade_in, lr_factor):
        with torch.no_grad():
            for k, p in params.items():
                if p.grad is None or (not fade_in and k in self.excluded):
                    continue
                adam_update(p, p.grad, states[k], self.lr * lr_factor, self.b1, self.b2, self.adam_eps)

    def d_step(self, z, real, noise=None, alpha=1.0, fade_in=False, lr_factor=1.0, eps_interp=None,
               cutoff_idx=None, z_mix=None):
        for p in self.d.values():
            p.grad = None
        with torch.no_grad():
            fake = self.gen(z, noise, alpha, fade_in, cutoff_idx, z_mix)
            if fade_in:   # reals follow the fade-in: up(down(x))*(1-a) + x*a  (progan/learner.py:771-779)
                real = ops.upsample2(ops.avgpool2(real)) * (1.0 - alpha) + real * alpha
        total, parts = d_loss(self.d, self.cfg, fake, real, self.loss, self.gp, self.lda, self.gamma,
                              self.eps_drift, alpha, fade_in, eps_interp, return_parts=True)
        total.backward()
        self._apply(self.d, self.st_d, fade_in, lr_factor)
        return total.detach(), parts

    def g_step(self, z, noise=None, alpha=1.0, fade_in=False, lr_factor=1.0, beta=None,
               cutoff_idx=None, z_mix=None):
        for p in self.g.values():
            p.grad = None
        d_frozen = {k: v.detach() for k, v in self.d.items()}
        fake = self.gen(z, noise, alpha, fade_in, cutoff_idx, z_mix)
        out = nets.disc_forward(d_frozen, fake, self.cfg, alpha=alpha, fade_in=fade_in)
        loss = ops.loss_gen(self.loss, out)
        loss.backward()
        self._apply(self.g, self.st_g, fade_in, lr_factor)
        if beta is not None:
            with torch.no_grad():
                for k, p in self.g.items():
                    self.lagged[k] = ewma_update(self.lagged[k], p.detach(), beta)
        return loss.detach()
