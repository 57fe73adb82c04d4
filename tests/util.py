"""Shared helpers for the test-suite (CPU side)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: z[k] for k in z.files}


def sub(d, prefix, as_torch=True):
    out = {}
    for k, v in d.items():
        if k.startswith(prefix):
            out[k[len(prefix):]] = torch.from_numpy(np.asarray(v)).clone() if as_torch else v
    return out


def t(a):
    return torch.from_numpy(np.asarray(a)).clone()


def rel_err(a, b):
    """max |a-b| / max(|b|max, tiny): the 'relative fp32' measure used for every parity assert."""
    a = torch.as_tensor(a, dtype=torch.float64).cpu()
    b = torch.as_tensor(b, dtype=torch.float64).cpu()
    denom = max(b.abs().max().item(), 1e-30)
    return (a - b).abs().max().item() / denom


def assert_close(a, b, tol, what=''):
    a_ = torch.as_tensor(a)
    b_ = torch.as_tensor(b)
    assert tuple(a_.shape) == tuple(b_.shape), f'{what}: shape {tuple(a_.shape)} vs {tuple(b_.shape)}'
    e = rel_err(a_, b_)
    assert e <= tol, f'{what}: rel err {e:.3e} > {tol:.1e}'


def resnet_zero_grad_key(k):
    """Generator residual-block conv biases: a per-channel shift in front of a BatchNorm."""
    return k.startswith('generator_model.') and k.endswith('conv2d.bias') and \
        ('conv_layer' in k or 'skip_connection' in k)
