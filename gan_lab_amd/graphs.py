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
