"""HIP-graph capture of the eval-mode generator forward (sampling / serving path).

Training steps at the benchmark sizes are GPU-bound (the kernels' own time equals the step time), but drawing
samples at batch 1..16 is launch-bound: a StyleGAN-1024 forward is ~110 small launches issued through ctypes.  The
forward has static shapes, so it is captured ONCE into a hipGraph (``torch.cuda.CUDAGraph``: the kernels of
``gan_lab_amd.ops`` are launched on torch's current stream, which is the capturing stream inside the context) and
replayed with new latents / noise copied into the graph's static input buffers.

The per-layer noise of StyleGAN is an explicit input of the captured forward (the device RNG's (seed, offset) are
by-value kernel arguments and would be frozen inside a graph): it is redrawn into the static buffers before every
replay, or pinned by the caller.
"""
import torch

from . import ops, rng


class GraphedGenerator(object):
    """``g = GraphedGenerator(gen_model, batch); img = g(z)`` - same result as ``gen_model.eval()(z, noise=...)``.

    The weight-packing kernels are captured too, so the parameters are read at replay time: weights updated in
    place (the fused optimiser writes through the same storage) are picked up; growing the network (new resolution)
    or changing eval-time switches (truncation psi, noise on / off) needs a new capture."""

    def __init__(self, gen, batch, len_z=None, warmup=2, follow_weight_updates=True):
        if gen.training:
            raise ValueError('GraphedGenerator captures the eval-mode forward: call gen.eval() first')
        self.gen, self.batch = gen, int(batch)
        dev = next(gen.parameters()).device
        len_z = len_z if len_z is not None else gen.len_latent + getattr(gen, 'num_classes', 0)
        self.z = torch.zeros(self.batch, len_z, device=dev)
        self.noise = None
        layers = getattr(gen, 'gen_layers', None)
        if layers is not None and getattr(gen, 'use_noise', False):      # StyleGAN: one (B,1,H,W) map per layer
            self.noise = [torch.zeros(self.batch, 1, 4 * 2 ** (n // 2), 4 * 2 ** (n // 2), device=dev)
                          for n in range(len(layers))]
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):            # warm-up off the capture: allocator pools, packed-weight cache
            for _ in range(warmup):
                self._forward()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        # follow_weight_updates: the packed-weight cache must MISS during the capture so that the pack kernels are
        # part of the graph and every replay re-packs from the parameters' current values (costs one pass over the
        # weights per replay: ~0.4 ms for the 26 M parameters of StyleGAN-1024).  A serving process with frozen
        # weights passes False: the graph then reads the packed copies made during the warm-up.
        if follow_weight_updates:
            ops.bump_weight_epoch()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self._forward()
        if follow_weight_updates:
            # The capture inserted packed-weight tensors into the cache that live in the graph's private pool and that
            # no kernel has written yet (capture records, it does not execute): an eager forward of the same module
            # before the first replay would convolve with uninitialised weights, and later ones would alias buffers
            # every replay rewrites.  Drop them - eager calls re-pack into ordinary allocations.
            ops.bump_weight_epoch()

    def _forward(self):
        with torch.no_grad():
            return self.gen(self.z, noise=self.noise) if self.noise is not None else self.gen(self.z)

    def redraw_noise(self):
        if self.noise is not None:
            for buf in self.noise:
                buf.copy_(rng.randn(buf.shape, buf.device))

    def __call__(self, z, noise=None, redraw_noise=True):
        """z: (batch, len_z).  ``noise``: list of per-layer maps to pin; otherwise fresh noise is drawn unless
        ``redraw_noise`` is False (then the previous maps are reused).  Returns the graph's output buffer - clone it
        to keep it across calls."""
        if tuple(z.shape) != tuple(self.z.shape):
            raise ValueError(f'latents must be {tuple(self.z.shape)}, got {tuple(z.shape)}')
        self.z.copy_(z)
        if self.noise is not None:
            if noise is not None:
                for buf, nz in zip(self.noise, noise):
                    buf.copy_(nz.expand_as(buf))
            elif redraw_noise:
                self.redraw_noise()
        self.graph.replay()
        return self.out


class _StepGraphs(object):
    """Capture / replay machinery shared by the learners' step graphs: a 64-byte device block of per-replay scalars, one
    memory pool, graphs keyed by (half, variant)."""

    def __init__(self, learner, warmup=2):
        self.L = learner
        self.warmup = int(warmup)
        self._reset()

    def _reset(self):
        self.graphs, self.pool, self.sig, self.calls = {}, None, None, 0
        self.block = self.real = None
        self.losses = {}
        # what the captured launches read through raw pointers and nothing else owns: the packed-weight buffers and the
        # descriptor tables of the batched re-pack that existed when a graph was captured.  The pack cache may drop them
        # (ops.bump_weight_epoch() without ranges: cache overflow, GraphedGenerator, growth) - the graphs must not lose them
        self.keepalive = []
        self.failed = None          # the exception of a capture that did not go through: this phase then steps eagerly

    def eligible(self):
        from . import parallel
        return parallel.world_size() == 1 and torch.cuda.is_available() and self.failed is None

    def _common_signature(self, real):
        # ``_graph_gen``: bumped by the learner whenever it rebuilds arenas or optimisers (an ``id()`` can be recycled by a
        # checkpoint load at the same resolution).  ``ops.pack_generation()``: bumped when the pack cache is flushed or a
        # descriptor table is rebuilt - the graphs own what they captured (``keepalive``), but a re-capture then packs
        # into the live cache again instead of carrying a private copy of every packed weight forever.
        # The batch's leading size is NOT part of it: a short last batch steps eagerly and the graphs stay (__call__)
        L = self.L
        return (getattr(L, '_graph_gen', 0), ops.pack_generation(), L.batch_size, tuple(real.shape[1:]), real.device,
                ops.get_compute_dtype(), bool(L.gen_model.training), bool(L.disc_model.training))

    def _check(self, sig):
        if sig != self.sig:
            self._reset()
            self.sig = sig

    def _capture(self, key, fn):
        """Record ``fn()`` (a learner half returning its loss) as the graph ``key``."""
        L = self.L
        dev = self.real.device
        if self.block is None:
            self.block = torch.zeros(16, dtype=torch.int32, device=dev)
            self.pool = torch.cuda.graph_pool_handle()
        if key[0] not in self.losses:
            self.losses[key[0]] = torch.zeros((), device=dev)
        base = self.block.data_ptr()
        # the graph packs the weights it uses itself: every cached pack is stale, and the steady state's batched re-pack
        # (one launch per parameter range, descriptor table already on the device) is what gets captured
        ops.mark_packs_stale()
        known = set(ops._PACK_CACHE.keys())
        self._own_pack_state()
        start = rng._STATE['offset']
        rng.begin_device_offsets(self.block)
        L.opt_disc.dev_scalars, L.opt_gen.dev_scalars = base + 16, base + 28
        graph = torch.cuda.CUDAGraph()
        import warnings
        try:
            with warnings.catch_warnings():
                # (a capture that an exception - or Ctrl-C - cuts short ends empty: torch says so in a warning of its own; the
                # graph object is dropped below, nothing is left behind)
                warnings.filterwarnings('ignore', message='The CUDA Graph is empty')
                with torch.cuda.graph(graph, pool=self.pool):
                    self.losses[key[0]].copy_(fn())
        except Exception as exc:      # an op that cannot be captured (a host sync, an allocation outside the pool): no retry
            self.failed = exc
            warnings.warn(f'step-graph capture of {key} failed ({type(exc).__name__}: {exc}); stepping eagerly', RuntimeWarning)
            raise
        finally:
            L.opt_disc.dev_scalars = L.opt_gen.dev_scalars = None
            draws = rng.end_device_offsets()
            rng._STATE['offset'] = start              # a capture records, it does not run: nothing was drawn yet
            # packed weights first made during the capture live in the graph's pool: not for eager use
            for k in [k for k in ops._PACK_CACHE if k not in known]:
                del ops._PACK_CACHE[k]
            ops.mark_packs_stale()
            self._own_pack_state()          # tables uploaded / entries re-packed while this capture ran
        self.graphs[key] = (graph, draws)

    def _own_pack_state(self):
        seen = {id(t) for t in self.keepalive}
        for e in ops._PACK_CACHE.values():
            for t in (e.out, e.w):
                if id(t) not in seen:
                    seen.add(id(t))
                    self.keepalive.append(t)
        for tab in ops._PACK_TABLES.values():
            if id(tab[1]) not in seen:
                seen.add(id(tab[1]))
                self.keepalive.append(tab[1])

    def _replay(self, key, opts):
        """``opts``: which optimisers step inside this graph (their step counts advance on the host here)."""
        L = self.L
        graph, draws = self.graphs[key]
        scal = [0.0] * 6
        if 'd' in opts:
            scal[0:3] = L.opt_disc.host_scalars()
        if 'g' in opts:
            scal[3:6] = L.opt_gen.host_scalars()
        ops.set_step_scalars(self.block, rng._STATE['offset'], scal)
        graph.replay()
        rng._STATE['offset'] += draws
        ops.mark_packs_stale()          # the replay rewrote parameters (and the cached packs' buffers) on its own
        return self.losses[key[0]]


class GraphedStep(_StepGraphs):
    """One main iteration of a stabilised phase - ``d_step(real, defer_update=True)`` then
    ``g_step(d_update_pending=True)`` (progan/learner.py:734-943: critic loss + gradient penalty + backward, generator
    loss + backward, both Adam updates, the EWMA generator) - replayed as HIP graphs.

    The launch-bound configurations (StyleGAN-128 at batch 8: ~1300 launches of a few microseconds each per step; the
    4^2 / 8^2 phases of a growth schedule) spend their step on the HOST: ctypes + autograd issue a launch
    every ~12 us while the kernels need less.  A replay submits the same launches from the runtime's graph executor.
    What a captured launch cannot carry as an argument lives in a 64-byte device block that ONE ordinary launch rewrites
    before each replay (``ganlab_set_step_scalars``):
      * the Philox stream position - the latent / per-layer noise draws pass their distance from it
        (``rng.begin_device_offsets``), so replay k draws exactly what eager step k would have drawn;
      * (lr, 1 - beta1^t, 1 - beta2^t) of both optimisers (``FusedAdam.dev_scalars``) - the LR schedule keeps working.
    The style-mixing cut (a host coin and a host integer per generator forward, stylegan/architectures.py:415-422) selects
    among captured VARIANTS of each half: ('d', cut) and ('g', cut), at most 2 * (layers + 1) graphs sharing one memory
    pool, all captured at the first replayed iteration (``precapture``).  Each graph re-packs the conv weights it uses
    (every cached pack is stale when the capture starts), so it depends on nothing but the parameter / optimiser / EWMA
    buffers, the static real batch and the scalar block.  Results are bit-identical to the eager steps
    (tests/test_gpu_learner.py::test_graphed_step_equals_eager).

    Single process only (the bucketed all-reduce of ``parallel.GradReducer`` is host-driven), stabilised phases only (alpha
    is a launch argument), one critic and one generator iteration per main iteration; anything else - and the first
    ``warmup`` calls, which let allocator pools and lazily built tables settle - runs eagerly.  A change of network,
    optimiser, batch size or phase drops the graphs."""

    def __init__(self, learner, warmup=2, precapture=True):
        self.precapture = bool(precapture)      # capture every mixing variant at the first replayed iteration
        super().__init__(learner, warmup)

    def eligible(self):
        return super().eligible() and not self.L.gen_model.fade_in_phase

    def _mix_kwargs(self):
        """The generator's own host draw of the mixing cut, made here so that it can select the graph."""
        g = self.L.gen_model
        if not hasattr(g, 'draw_mixing_cutoff'):
            return None, None
        cut = g.draw_mixing_cutoff()
        return cut, {'_mix': (cut, None)}

    def _all_cuts(self):
        g = self.L.gen_model
        if not hasattr(g, 'draw_mixing_cutoff') or not g._use_mixing_reg or g.pct_mixing_reg <= 0:
            return [None]
        hi = 2 * g.scale_stage if g.alpha != 0 else 2 * g.scale_stage - 2
        return [None] + list(range(1, hi))

    # -- the eager form (also what gets captured) ----------------------------------------------------------------------
    def _d_half(self, real, kw):
        self.L.set_requires_grad_disc(True)
        return self.L.d_step(real, defer_update=True, gen_kwargs=kw)

    def _g_half(self, kw):
        self.L.set_requires_grad_disc(False)
        return self.L.g_step(d_update_pending=True, gen_kwargs=kw)

    def _capture_half(self, kind, cut, kw):
        self._capture((kind, cut), (lambda: self._d_half(self.real, kw)) if kind == 'd' else (lambda: self._g_half(kw)))

    def __call__(self, real):
        """-> (loss_d, loss_g) device scalars (valid until the next call)."""
        L = self.L
        g = L.gen_model
        self._check(self._common_signature(real) + (g.curr_res, bool(g.fade_in_phase)))
        self.calls += 1
        off_size = real.shape[0] != L.batch_size or (self.real is not None and real.shape != self.real.shape)
        if not self.eligible() or self.calls <= self.warmup or off_size:
            cut_d, kw_d = self._mix_kwargs()
            ld = self._d_half(real, kw_d)
            cut_g, kw_g = self._mix_kwargs()
            return ld, self._g_half(kw_g)
        if self.real is None:
            self.real = torch.empty_like(real)
            self.real.copy_(real)
            if self.precapture:
                try:
                    for cut in self._all_cuts():
                        kw = {'_mix': (cut, None)} if hasattr(g, 'draw_mixing_cutoff') else None
                        for kind in ('d', 'g'):
                            self._capture_half(kind, cut, kw)
                except Exception:
                    cut_d, kw_d = self._mix_kwargs()
                    ld = self._d_half(real, kw_d)
                    cut_g, kw_g = self._mix_kwargs()
                    return ld, self._g_half(kw_g)
        self.real.copy_(real)
        out = []
        for kind in ('d', 'g'):
            cut, kw = self._mix_kwargs()
            if (kind, cut) not in self.graphs:
                try:
                    self._capture_half(kind, cut, kw)
                except Exception:
                    # nothing of this half has run (a capture records): finish the iteration eagerly, stay eager afterwards
                    out.append(self._d_half(real, kw) if kind == 'd' else self._g_half(kw))
                    if kind == 'd':
                        cut_g, kw_g = self._mix_kwargs()
                        out.append(self._g_half(kw_g))
                    return tuple(out)
            # the deferred critic update and the generator update both sit in the 'g' half
            out.append(self._replay((kind, cut), ('d', 'g') if kind == 'g' else ()))
        return tuple(out)
