"""Process-wide counter-based RNG stream for latents and per-layer noise (Philox4x32-10 in HIP)."""
import torch

from . import ops

_STATE = {'seed': 0x5EED, 'offset': 0}


def manual_seed(seed, rank=0):
    _STATE['seed'] = (int(seed) * 0x9E3779B97F4A7C15 + int(rank) * 0xD1B54A32D192ED03) & (2 ** 64 - 1)
    _STATE['offset'] = 0


def randn(shape, device='cuda'):
    n = 1
    for s in shape:
        n *= int(s)
    out = ops.randn(tuple(shape), _STATE['seed'], _STATE['offset'], device)
    _STATE['offset'] += (n + 3) // 4
    return out
