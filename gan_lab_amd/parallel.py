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


class _Watch(object):
    """Bucket plan of one flat gradient arena + the state of the backward that is filling it."""
    __slots__ = ('arena', 'bounds', 'bucket_of', 'need', 'count', 'order', 'observed', 'launched', 'early', 'hooks',
                 'agreed')

    def __init__(self, arena, bucket_elems):
        self.arena = arena
        # buckets: maximal runs of whole parameters, closed once they reach ``bucket_elems`` floats (a parameter is
        # never split: its gradient arrives in one piece)
        self.bounds, self.bucket_of = [], []
        lo = 0
        for i, (o, n) in enumerate(zip(arena.offsets, arena.sizes)):
            self.bucket_of.append(len(self.bounds))
            end = arena.offsets[i + 1] if i + 1 < len(arena.offsets) else arena.total
            if end - lo >= bucket_elems or i + 1 == len(arena.offsets):
                self.bounds.append((lo, end))
                lo = end
        self.need = [0] * len(self.bounds)
        for b in self.bucket_of:
            self.need[b] += 1
        self.count = [0] * len(self.bounds)
        self.order = None          # launch order agreed between the ranks (None until one backward was observed)
        self.observed = []         # completion order seen in the running backward
        self.launched = set()
        self.early = 0             # buckets the hooks launched from inside the last backward
        self.hooks = []
        self.agreed = False


class GradReducer(object):
    """MEAN all-reduce of a flat gradient arena in ~``bucket_mb`` pieces.

    Two ways in:
      * ``start(flat)`` / ``finish()``: everything is launched (asynchronously, on RCCL's stream) when ``start`` is
        called - the round-1 behaviour, still used for plain tensors.
      * ``watch(arena)`` once per arena, then ``arm(arena)`` in front of a ``backward()`` and ``start(arena.gflat)``
        behind it: every parameter carries a post-accumulate hook, and a bucket's all-reduce is launched from inside
        the backward sweep as soon as its last gradient has been accumulated into the arena - the exchange of the deep
        layers' gradients overlaps the input- / weight-gradient kernels of the shallow ones; ``start`` launches what is
        left (buckets holding a parameter that received no gradient), ``finish`` waits.
    Collectives on one communicator must be issued in the same order by every rank.  The FIRST armed backward of an
    arena therefore only records the order in which its buckets complete and launches nothing early; the ranks then
    compare what they saw (one MIN / MAX all-reduce of the order vector) and use it as the fixed launch order from the
    next step on - a bucket is launched when it is complete AND every bucket in front of it in that order has been
    launched.  Ranks that disagree (they should not: same graph, deterministic engine) keep the launch-at-``start``
    behaviour.
    """

    def __init__(self, bucket_mb=32, group=None):
        self.group = group
        self.bucket_elems = max(int(bucket_mb * (1 << 20) // 4), 1)
        self._pending = []
        self._flat = None
        self._scale = None
        self._watch = {}
        self._armed = None

    # ---- plain flat tensors --------------------------------------------------------------------------------------
    def _op(self):
        use_avg = dist.get_backend(self.group) == 'nccl'
        return (dist.ReduceOp.AVG if use_avg else dist.ReduceOp.SUM), \
            (None if use_avg else 1.0 / dist.get_world_size(self.group))

    def start(self, flat_grad, n=None):
        """Asynchronously mean-reduce ``flat_grad[:n]`` across ranks (no-op for a single rank).  For the gradient
        buffer of an armed arena: launch the buckets the backward hooks have not launched yet."""
        if not is_dist():
            self._armed = None
            return
        w = self._armed
        if w is not None and flat_grad.data_ptr() == w.arena.gflat.data_ptr():
            self._start_armed(w)
            return
        assert not self._pending, 'finish() the previous reduction first'
        n = flat_grad.numel() if n is None else n
        op, self._scale = self._op()
        for lo in range(0, n, self.bucket_elems):
            chunk = flat_grad[lo:min(lo + self.bucket_elems, n)]
            self._pending.append(dist.all_reduce(chunk, op=op, group=self.group, async_op=True))
        self._flat = flat_grad[:n]

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

    # ---- arenas: bucket launches from inside the backward sweep --------------------------------------------------
    def watch(self, arena):
        """Plan the buckets of ``arena`` and hook its parameters (once per arena; a growth event builds new arenas)."""
        old = self._watch.get(id(arena))
        if old is not None:
            self.unwatch(old.arena)
        w = _Watch(arena, self.bucket_elems)
        for i, p in enumerate(arena.params):
            w.hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(w, w.bucket_of[i])))
        self._watch[id(arena)] = w
        return w

    def unwatch(self, arena):
        w = self._watch.pop(id(arena), None)
        if w is not None:
            for h in w.hooks:
                h.remove()
            if self._armed is w:
                self._armed = None

    def unwatch_all(self):
        """Drop every plan and hook (the learner rebuilds its arenas at a growth event: the surviving ``nn.Parameter``
        objects move into new arenas and must not keep the hooks of the old ones)."""
        for w in list(self._watch.values()):
            self.unwatch(w.arena)

    def abandon(self):
        """Forget every plan, hook and pending handle WITHOUT waiting on anything: the learner rebuilds its arenas at a
        growth event (the surviving ``nn.Parameter`` objects move into new arenas and must not keep the old hooks), and
        the interrupt path must not touch a collective that the other ranks may never have launched."""
        self.unwatch_all()
        self._pending, self._flat, self._scale, self._armed = [], None, None, None

    def _make_hook(self, w, b):
        def hook(_param):
            if self._armed is not w:
                return
            w.count[b] += 1
            if w.count[b] == w.need[b]:
                w.observed.append(b)
                if w.order is not None:
                    self._launch_ready(w)
        return hook

    def arm(self, arena):
        """Call right before the ``backward()`` that fills ``arena.gflat`` (after its zero_grad)."""
        self._armed = None
        if not is_dist():
            return
        w = self._watch.get(id(arena))
        if w is None or w.arena is not arena:
            w = self.watch(arena)
        assert not self._pending, 'finish() the previous reduction first'
        w.count = [0] * len(w.bounds)
        w.observed, w.launched, w.early = [], set(), 0
        self._armed = w

    def _launch(self, w, b):
        lo, hi = w.bounds[b]
        op, self._scale = self._op()
        self._pending.append(dist.all_reduce(w.arena.gflat[lo:hi], op=op, group=self.group, async_op=True))
        w.launched.add(b)

    def _launch_ready(self, w):
        for b in w.order:
            if b in w.launched:
                continue
            if w.count[b] < w.need[b]:
                break                       # in-order launch: nothing behind an incomplete bucket goes out
            self._launch(w, b)

    def _start_armed(self, w):
        self._armed = None
        first = w.order is None
        w.early = len(w.launched)
        rest = [b for b in (w.order or []) if b not in w.launched]
        rest += [b for b in range(len(w.bounds)) if b not in w.launched and b not in rest]
        for b in rest:
            self._launch(w, b)
        self._flat = w.arena.gflat
        if first:
            self._agree(w)

    def _agree(self, w):
        """Adopt the observed completion order as the launch order if every rank saw the same one."""
        seen = list(w.observed) + [b for b in range(len(w.bounds)) if b not in w.observed]
        t = torch.tensor(seen + [len(w.observed)], dtype=torch.int64, device=w.arena.gflat.device)
        lo, hi = t.clone(), t.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        w.agreed = bool(torch.equal(lo, hi))
        # (a rank-local disagreement is impossible to act on consistently: all ranks see lo != hi or none does)
        w.order = seen if w.agreed else []

    def bucket_report(self, arena):
        """(bucket bounds, agreed launch order, buckets launched from inside the last backward) - for tests / logs."""
        w = self._watch.get(id(arena))
        if w is None:
            return None
        return {'bounds': list(w.bounds), 'order': None if w.order is None else list(w.order), 'agreed': w.agreed,
                'observed': list(w.observed), 'launched_in_backward': w.early}


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
