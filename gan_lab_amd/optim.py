"""Flat parameter arenas + fused Adam / EWMA (K15) for the G and D parameter sets.

MI355X-first layout: all parameters of a network live in ONE contiguous fp32 buffer (and all
gradients in another), so that
  * the optimiser step is one kernel launch per contiguous run instead of ~100 small ones
    (reference: torch.optim.Adam(betas=(0,.99)) built by backprop_utils.py:109-120 and re-created at
    every phase boundary, progan/learner.py:1064-1095),
  * the EWMA generator update (progan/learner.py:909-916) is one launch,
  * the data-parallel gradient exchange is a handful of large RCCL all-reduces over the flat
    gradient buffer (parallel.py) rather than one per tensor.
Parameters stay ordinary ``nn.Parameter`` objects (views into the arena), so ``state_dict()``,
``named_parameters()`` and checkpoints are unchanged.
"""
from collections import OrderedDict

import torch

from . import ops

_ALIGN = 4  # floats (16 B): keeps every parameter view 16-byte aligned for float4 kernels


def _round_up(n, m):
    return (n + m - 1) // m * m


class ParamArena(object):
    """Re-homes the given named parameters into one flat buffer (+ one flat grad buffer)."""

    def __init__(self, named_params, device=None):
        self.names, self.params, self.offsets, self.sizes = [], [], [], []
        named_params = [(k, p) for k, p in named_params]
        if not named_params:
            raise ValueError('empty parameter list')
        device = device if device is not None else named_params[0][1].device
        off = 0
        for k, p in named_params:
            self.names.append(k)
            self.params.append(p)
            self.offsets.append(off)
            self.sizes.append(p.numel())
            off += _round_up(p.numel(), _ALIGN)
        self.total = off
        self.serial = 0          # bumped by zero_grad(): ops.direct_param_grads writes each slot once per step
        self.flat = torch.zeros(off, dtype=torch.float32, device=device)
        self.gflat = torch.zeros(off, dtype=torch.float32, device=device)
        with torch.no_grad():
            for p, o, n in zip(self.params, self.offsets, self.sizes):
                self.flat[o:o + n].copy_(p.detach().reshape(-1))
        self.attach()

    def attach(self):
        """(Re-)point every parameter's data and grad at its arena slot."""
        for p, o, n in zip(self.params, self.offsets, self.sizes):
            p.data = self.flat[o:o + n].view(p.shape)
            p.grad = self.gflat[o:o + n].view(p.shape)
            p._ganlab_arena = self

    def is_attached(self):
        base, gbase = self.flat.data_ptr(), self.gflat.data_ptr()
        for p, o in zip(self.params, self.offsets):
            if p.data_ptr() != base + 4 * o or p.grad is None or p.grad.data_ptr() != gbase + 4 * o:
                return False
        return True

    def reabsorb(self):
        """Parameters were re-allocated behind our back (e.g. ``model.to('cpu')`` and back for a
        checkpoint, progan/learner.py:962-985): copy their values in and re-attach."""
        with torch.no_grad():
            for p, o, n in zip(self.params, self.offsets, self.sizes):
                if p.data_ptr() != self.flat.data_ptr() + 4 * o:
                    self.flat[o:o + n].copy_(p.detach().reshape(-1).to(self.flat.device))
        self.attach()

    def zero_grad(self):
        self.gflat.zero_()
        self.serial += 1
        if not self.is_attached():
            self.reabsorb()

    def slot(self, name):
        i = self.names.index(name)
        return self.offsets[i], self.sizes[i]

    def views_of(self, flat):
        return OrderedDict((k, flat[o:o + n].view(p.shape))
                           for k, p, o, n in zip(self.names, self.params, self.offsets, self.sizes))


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (no amsgrad; L2 weight decay) with the update done by the HIP
    kernel ``ganlab_adam_f32`` over maximal runs of parameters that are adjacent in memory - one
    launch for an arena-backed network.  Parameters whose ``grad`` is None are skipped, like torch."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._runs = None
        self._sig = None
        # step-graph capture (graphs.GraphedStep): device address of this optimiser's (lr, 1-beta1^t, 1-beta2^t); while
        # set, step() launches the kernel that reads them there and leaves the step count to host_scalars()
        self.dev_scalars = None

    def host_scalars(self, advance=True):
        """(lr, 1 - beta1^t, 1 - beta2^t) of the NEXT step of the (single) parameter group, advancing its step count:
        what ``step()`` computes on the host, for a replayed step."""
        group = self.param_groups[0]
        cached = self.state.get('_fused', {}).get(id(group))
        if cached is None:
            raise RuntimeError('host_scalars() before the first eager step()')
        if advance:
            cached['step'] += 1
        t = cached['step']
        b1, b2 = group['betas']
        return float(group['lr']), 1 - b1 ** t, 1 - b2 ** t

    def _build_runs(self, params):
        items = sorted(((p.data_ptr(), p) for p in params), key=lambda t: t[0])
        runs, cur = [], None
        for ptr, p in items:
            n = p.numel()
            gptr = p.grad.data_ptr()
            if cur is not None:
                gap = (ptr - cur['end']) // 4
                ggap = (gptr - cur['gend']) // 4
                if 0 <= gap < _ALIGN and gap == ggap and (ptr - cur['end']) % 4 == 0:
                    cur['params'].append(p)
                    cur['end'] = ptr + 4 * n
                    cur['gend'] = gptr + 4 * n
                    continue
            cur = dict(params=[p], start=ptr, end=ptr + 4 * n, gstart=gptr, gend=gptr + 4 * n)
            runs.append(cur)
        for r in runs:
            r['n'] = (r['end'] - r['start']) // 4
            r['offs'] = {id(p): (p.data_ptr() - r['start']) // 4 for p in r['params']}
        return runs

    @torch.no_grad()
    def step(self, closure=None):
        touched = []
        if self.dev_scalars is not None and len(self.param_groups) != 1:
            # host_scalars() describes ONE (lr, step count): a replayed multi-group optimiser would train every group with
            # group 0's.  graphs.GraphedStep only captures the learners' single-group Adam.
            raise RuntimeError('FusedAdam.dev_scalars (step-graph capture) supports exactly one parameter group')
        for group in self.param_groups:
            params = [p for p in group['params'] if p.grad is not None]
            if not params:
                continue
            sig = tuple((p.data_ptr(), p.grad.data_ptr()) for p in params)
            st = self.state.setdefault('_fused', {})
            key = id(group)
            cached = st.get(key)
            if cached is None or cached['sig'] != sig:
                runs = self._build_runs(params)
                old = cached
                cached = dict(sig=sig, runs=runs, step=old['step'] if old else 0)
                for r in runs:
                    dev = r['params'][0].device
                    r['m'] = torch.zeros(r['n'], dtype=torch.float32, device=dev)
                    r['v'] = torch.zeros(r['n'], dtype=torch.float32, device=dev)
                    if old is not None:  # carry moments across a re-layout
                        self._carry(old, r)
                st[key] = cached
            b1, b2 = group['betas']
            if self.dev_scalars is None:
                cached['step'] += 1
                t = cached['step']
                bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
            for r in cached['runs']:
                p0 = r['params'][0]
                # raw views over the whole run (padding floats have zero grad -> stay zero)
                pv = torch.as_strided(p0.data.reshape(-1), (r['n'],), (1,))
                gv = torch.as_strided(p0.grad.reshape(-1), (r['n'],), (1,))
                if self.dev_scalars is not None:
                    ops.adam_step_dev(pv, gv, r['m'], r['v'], self.dev_scalars, b1, b2, group['eps'], group['weight_decay'])
                else:
                    ops.adam_step(pv, gv, r['m'], r['v'], group['lr'], b1, b2, group['eps'], group['weight_decay'],
                                  bc1, bc2)
                touched.append((r['start'], r['end']))
        # packed conv weights inside the rewritten memory are stale: they are re-packed (all of them, one launch) at
        # their next use; the other network's stay valid
        if touched:
            ops.bump_weight_epoch(touched)

    @staticmethod
    def _carry(old, run):
        """Copy the Adam moments of parameters that survive a re-layout (keyed by parameter object)."""
        for p in run['params']:
            for r in old['runs']:
                o = r['offs'].get(id(p))
                if o is not None:
                    n, no = p.numel(), run['offs'][id(p)]
                    run['m'][no:no + n].copy_(r['m'][o:o + n])
                    run['v'][no:no + n].copy_(r['v'][o:o + n])

    def export_moments(self, named_params):
        """Plain-data Adam state for checkpoints: {'step', 'exp_avg': {name: tensor}, 'exp_avg_sq': {...}}."""
        out = dict(step=0, exp_avg={}, exp_avg_sq={})
        for group in self.param_groups:
            cached = self.state.get('_fused', {}).get(id(group))
            if cached is None:
                continue
            out['step'] = cached['step']
            for name, p in named_params:
                for r in cached['runs']:
                    o = r['offs'].get(id(p))
                    if o is not None:
                        out['exp_avg'][name] = r['m'][o:o + p.numel()].view(p.shape).detach().cpu().clone()
                        out['exp_avg_sq'][name] = r['v'][o:o + p.numel()].view(p.shape).detach().cpu().clone()
        return out

    @torch.no_grad()
    def import_moments(self, named_params, saved):
        """Inverse of export_moments (parameters must already have their arena-backed grads)."""
        named_params = list(named_params)
        for group in self.param_groups:
            params = [p for p in group['params'] if p.grad is not None]
            runs = self._build_runs(params)
            for r in runs:
                dev = r['params'][0].device
                r['m'] = torch.zeros(r['n'], dtype=torch.float32, device=dev)
                r['v'] = torch.zeros(r['n'], dtype=torch.float32, device=dev)
            for name, p in named_params:
                if name not in saved['exp_avg']:
                    continue
                for r in runs:
                    o = r['offs'].get(id(p))
                    if o is not None:
                        r['m'][o:o + p.numel()].copy_(saved['exp_avg'][name].reshape(-1))
                        r['v'][o:o + p.numel()].copy_(saved['exp_avg_sq'][name].reshape(-1))
            self.state.setdefault('_fused', {})[id(group)] = dict(
                sig=tuple((p.data_ptr(), p.grad.data_ptr()) for p in params), runs=runs, step=int(saved['step']))

    def zero_grad(self, set_to_none=False):
        # keep the arena-backed .grad views alive: zero in place
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is not None:
                    p.grad.zero_()


class EwmaTracker(object):
    """EWMA shadow of the generator parameters (progan/learner.py:462-472, :909-916, :662-684):
    ``lagged = p*(1-beta) + lagged*beta`` over the whole arena in one launch.  ``lagged_params`` is
    the name -> tensor view dict the reference keeps (same key order as ``named_parameters()``)."""

    def __init__(self, arena):
        self.rebuild(arena, None)

    def rebuild(self, arena, old_lagged, rename=None):
        """New arena after a growth step: carry over the values of surviving parameters (after the
        reference's ``torgb -> prev_torgb`` re-keying), start new parameters from their current value."""
        self.arena = arena
        self.flat = arena.flat.detach().clone()
        self.lagged_params = arena.views_of(self.flat)
        if old_lagged is not None:
            rename = rename or {}
            with torch.no_grad():
                for k_old, v in old_lagged.items():
                    if rename and k_old in rename.values():
                        continue          # e.g. the previous prev_torgb.* is overwritten by the renamed torgb.*
                    k = rename.get(k_old, k_old)
                    if k in self.lagged_params and self.lagged_params[k].shape == v.shape:
                        self.lagged_params[k].copy_(v)

    def update(self, beta):
        if not self.arena.is_attached():
            self.arena.reabsorb()
        ops.ewma_step(self.flat, self.arena.flat, float(beta) if beta else 0.0)
