"""Process-wide counter-based RNG stream for latents and per-layer noise (Philox4x32-10 in HIP)."""
import torch

from . import ops

_STATE = {'seed': 0x5EED, 'offset': 0}
# While a step graph is being captured (graphs.GraphedStep) the stream position cannot be a launch argument - it would be
# frozen into the graph: draws then read a device-resident base (rewritten before every replay) and pass only their
# distance from the position the capture started at.
_DEVICE_BASE = {'block': None, 'start': 0}


def begin_device_offsets(block):
    _DEVICE_BASE['block'], _DEVICE_BASE['start'] = block, _STATE['offset']


def end_device_offsets():
    """Back to by-value offsets; returns how far the stream advanced since ``begin_device_offsets``."""
    n = _STATE['offset'] - _DEVICE_BASE['start']
    _DEVICE_BASE['block'] = None
    return n


def manual_seed(seed, rank=0):
    _STATE['seed'] = (int(seed) * 0x9E3779B97F4A7C15 + int(rank) * 0xD1B54A32D192ED03) & (2 ** 64 - 1)
    _STATE['offset'] = 0


def seed_from_config(random_seed):
    """Seed the device stream the way config.py:337-376 seeds numpy / torch: ``random_seed`` in [0, 2**32) is used as
    is, -1 draws a fresh one.  Called by the learners at construction (the process group, if any, exists by then):
    rank 0's seed is shared, then every rank mixes its own rank in, so that the replicas of a data-parallel run draw
    DIFFERENT latents and per-layer noise (an effective fake batch of B * world_size) yet a fixed ``random_seed``
    reproduces the run."""
    import os
    from . import parallel
    seed = int(random_seed) if random_seed is not None else -1
    if seed < 0:
        seed = int.from_bytes(os.urandom(7), 'little')
    if parallel.is_dist():
        import torch.distributed as dist
        box = [seed]
        dist.broadcast_object_list(box, src=0)
        seed = int(box[0])
    manual_seed(seed, parallel.rank())
    return seed


def randn(shape, device='cuda'):
    n = 1
    for s in shape:
        n *= int(s)
    if _DEVICE_BASE['block'] is not None:
        out = ops.randn_dev(tuple(shape), _STATE['seed'], _DEVICE_BASE['block'], _STATE['offset'] - _DEVICE_BASE['start'],
                            device)
    else:
        out = ops.randn(tuple(shape), _STATE['seed'], _STATE['offset'], device)
    _STATE['offset'] += (n + 3) // 4
    return out
