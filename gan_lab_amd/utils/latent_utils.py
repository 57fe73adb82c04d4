"""Latent sampling on the device (reference: gan_lab/utils/latent_utils.py:12-19).

Normal latents come from the counter-based Philox generator of the HIP library (ops.randn) so that
every rank / every step draws from an explicit (seed, offset) stream."""
import torch

from .. import ops, rng


def gen_rand_latent_vars(num_samples, length, distribution='normal', device='cuda'):
    if distribution == 'normal':
        return rng.randn((num_samples, length), device)
    elif distribution == 'uniform':
        return torch.rand(num_samples, length, dtype=torch.float32, device=device)
    raise ValueError(distribution)
