"""Data parallelism over RCCL / xGMI (new work: the reference is single-device, README.md:136).

One process per GPU; each rank runs the identical phase machine on its own micro-batch.  The only
exchange step is the MEAN all-reduce of the D gradients after the D backward (R1 contribution
included) and of the G gradients after the G backward.  Because every network's
gradients live in ONE flat arena (optim.ParamArena) the exchange is a few large collectives over
contiguous memory - sized for xGMI's point-to-point links (7 x ~153 GB/s per GPU), not one small
all-reduce per tensor.  ``start()`` launches them asynchronously on RCCL's own stream so the next
forward pass (which does not depend on the gradients) overlaps the transfer; ``finish()`` is called
right before the optimiser step.
"""
import os

import torch
import torch.distributed as dist


def is_dist():
    """True when gradients must be exchanged.  GANLAB_DIST_WORLD1=1 keeps the RCCL code path on for a single rank
    (a 1-GPU box can then exercise init / broadcast / async all-reduce / AVG exactly as an 8-GPU node runs them)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get('GANLAB_DIST_WORLD1') == '1'


def world_size():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


class GradReducer(object):
    def __init__(self, bucket_mb=32, group=None):
        self.group = group
        self.bucket_elems = int(bucket_mb * (1 << 20) // 4)
        self._pending = []
        self._flat = None
        self._scale = None

    def start(self, flat_grad, n=None):
        """Asynchronously mean-reduce ``flat_grad[:n]`` across ranks (no-op for a single rank)."""
        if not is_dist():
            return
        assert not self._pending, 'finish() the previous reduction first'
        n = flat_grad.numel() if n is None else n
        backend = dist.get_backend(self.group)
        use_avg = backend == 'nccl'
        op = dist.ReduceOp.AVG if use_avg else dist.ReduceOp.SUM
        for lo in range(0, n, self.bucket_elems):
            chunk = flat_grad[lo:min(lo + self.bucket_elems, n)]
            self._pending.append(dist.all_reduce(chunk, op=op, group=self.group, async_op=True))
        self._flat = flat_grad[:n]
        self._scale = None if use_avg else 1.0 / dist.get_world_size(self.group)

    def finish(self):
        if not self._pending:
            return
        for h in self._pending:
            h.wait()
        self._pending = []
        if self._scale is not None:
            if self._flat.is_cuda:
                from . import ops
                ops.check(ops._lib.lib().ganlab_axpby_f32(ops._p(self._flat), None, ops._p(self._flat),
                                                          self._flat.numel(), self._scale, 0.0, ops._st()), 'axpby')
            else:
                self._flat.mul_(self._scale)   # gloo / CPU tensors (tests)
        self._flat = None

    def allreduce(self, flat_grad, n=None):
        self.start(flat_grad, n)
        self.finish()


def barrier(group=None):
    """All ranks wait here (no-op for a single process)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.barrier(group=group)


def broadcast_params(flat, src=0, group=None):
    """Make every replica start from rank `src`'s parameters."""
    if is_dist():
        dist.broadcast(flat, src=src, group=group)


def shard_of_global_batch(global_batch, r=None, w=None):
    """Rank r takes images [r*B, (r+1)*B) of each global batch (rank-major layout: minibatch-stddev
    groups and InstanceNorm statistics stay rank-local)."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    per = global_batch.shape[0] // w
    return global_batch[r * per:(r + 1) * per]
