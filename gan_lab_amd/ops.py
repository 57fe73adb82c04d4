"""Autograd ops over the HIP C-ABI (include/ganlab_hip.h).

Layering:
  k_*        thin launchers: torch tensors in (device memory + current stream), kernel call, tensor out.
  _Fn        torch.autograd.Function subclasses.  Every backward is itself written with other
             Functions of this module, so `torch.autograd.grad(..., create_graph=True)` (the R1 /
             WGAN-GP penalty, reference resnetgan/learner.py:811-815) differentiates straight
             through the hand-written kernels:
                conv is bilinear -> {fwd, dgrad, wgrad} is closed under differentiation;
                LeakyReLU'' = 0 -> act_bwd is linear in the cotangent, None towards the activation;
                blur is self-adjoint; pool2/up2 are an adjoint pair; mbstd has an explicit bwd-of-bwd.
  functional wrappers (conv2d, linear, bias_act, ...) used by custom_layers.py / the architectures.

All ops require contiguous fp32 CUDA(HIP) tensors and raise otherwise - there is no CPU path.
"""
import ctypes
import os

import numpy as np

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib
from ._lib import ACT_LRELU, ACT_NONE, PACK_DGRAD, PACK_FWD, ConvGeom, check


# ---------------------------------------------------------------------------------------------- #
# plumbing
# ---------------------------------------------------------------------------------------------- #
def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


# torch.cuda.current_stream() builds a Python Stream object through several layers of device-index resolution (~10 us;
# a quarter of the host time of a launch-bound step): ask the C binding for the raw handle of this process's device.
_RAW_STREAM = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_GET_DEVICE = getattr(torch._C, '_cuda_getDevice', None)


def _st():
    if _RAW_STREAM is None or _GET_DEVICE is None:
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    return ctypes.c_void_p(_RAW_STREAM(_GET_DEVICE()))      # two C calls (~0.3 us): follows torch.cuda.set_device


def _c(t, what='tensor'):
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.float32:
        raise TypeError(f'gan_lab_amd.ops: {what} must be a float32 tensor on the GPU (got '
                        f'{type(t).__name__}, {getattr(t, "device", None)}, {getattr(t, "dtype", None)}); '
                        f'the HIP path has no CPU fallback')
    return t if t.is_contiguous() else t.contiguous()


def _new(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


# ---- compute dtype of the dense 3x3 convolutions ---------------------------------------------------
# 'f32' (default): exact fp32 MFMA, the reference's arithmetic.  'bf16': BASELINE config #2 "bf16 compute /
# fp32 master" - eligible 3x3 layers multiply bf16-rounded operands on v_mfma_f32_16x16x32_bf16 with fp32
# accumulation (csrc/conv_bf16.hip); tensors in HBM, parameters, optimiser state and every other kernel stay fp32.
_COMPUTE = ['f32']


def set_compute_dtype(name):
    if name not in ('f32', 'bf16'):
        raise ValueError("compute dtype must be 'f32' or 'bf16'")
    _COMPUTE[0] = name


def get_compute_dtype():
    return _COMPUTE[0]


class compute_dtype(object):
    """with ops.compute_dtype('bf16'): ... - graphs BUILT inside keep their kernels for the backward."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self._prev = _COMPUTE[0]
        set_compute_dtype(self.name)

    def __exit__(self, *exc):
        _COMPUTE[0] = self._prev
        return False


class Geom:
    """Static description of one convolution (mirrors ganlab_conv_geom) + derived output size.
    ``up``: nearest 2x upsample folded in front; ``pool``: 2x2 average pool folded behind (stride-2
    fused kernels, csrc/conv_s2.hip).  ``s2`` tells whether the stride-2 fast path applies."""
    __slots__ = ('N', 'Cin', 'Hin', 'Win', 'Cout', 'ks', 'pad', 'up', 'pool', 'Ho', 'Wo', 's2', '_c', 'bf', 'bf_fused')

    _CACHE = {}

    def __new__(cls, N, Cin, Hin, Win, Cout, ks, pad, up=0, pool=0):
        # a training step builds the same few dozen geometries again and again; each costs two ctypes structures and
        # two library queries, so they are interned (instances are immutable after construction).  An instance enters
        # the cache only once its construction has SUCCEEDED: a rejected geometry raises on every attempt.
        key = (int(N), int(Cin), int(Hin), int(Win), int(Cout), int(ks), int(pad), int(bool(up)), int(bool(pool)),
               _COMPUTE[0])
        g = cls._CACHE.get(key)
        if g is None:
            g = object.__new__(cls)
            g._c = None
            g._build(*key[:9])
            if len(cls._CACHE) > 4096:
                cls._CACHE.clear()
            cls._CACHE[key] = g
        return g

    def __init__(self, N, Cin, Hin, Win, Cout, ks, pad, up=0, pool=0):
        pass            # built (and validated) in __new__

    def _build(self, N, Cin, Hin, Win, Cout, ks, pad, up, pool):
        if ks not in (1, 3):
            raise ValueError('conv kernels support ks in {1, 3}; a 4x4 valid conv runs as a linear')
        self.N, self.Cin, self.Hin, self.Win, self.Cout, self.ks, self.pad, self.up, self.pool = \
            int(N), int(Cin), int(Hin), int(Win), int(Cout), int(ks), int(pad), int(bool(up)), int(bool(pool))
        hv, wv = (2 * Hin, 2 * Win) if up else (Hin, Win)
        self.Ho, self.Wo = hv + 2 * pad - ks + 1, wv + 2 * pad - ks + 1
        if self.pool:
            self.Ho, self.Wo = self.Ho // 2, self.Wo // 2
        cg = ConvGeom(self.N, self.Cin, self.Hin, self.Win, self.Cout, self.ks, self.pad, self.up, self.pool)
        # bf16 compute mode (decided HERE, so a backward that runs outside the `compute_dtype` block still uses the
        # kernels its forward used).  ``bf``: the plain geometry at the resolution of the taps (what the weight-gradient
        # kernels take, on a materialised upsample of x or of the pooled layer's gy); ``bf_fused``: this geometry with
        # its up / pool flag - the forward and input-gradient kernels fold the resampling in
        self.bf = self.bf_fused = None
        if _COMPUTE[0] == 'bf16':
            v = ConvGeom(self.N, self.Cin, hv, wv, self.Cout, self.ks, self.pad, 0, 0)
            if _lib.lib().ganlab_conv_bf16_supported(ctypes.byref(v)):
                self.bf = v
                if (self.up or self.pool) and _lib.lib().ganlab_conv_bf16_supported(ctypes.byref(cg)):
                    self.bf_fused = cg
                elif self.pool:
                    self.bf = None
        self.s2 = bool(self.bf is None and (self.up or self.pool) and
                       _lib.lib().ganlab_conv_s2_supported(ctypes.byref(cg)))
        if self.pool and not self.s2 and self.bf_fused is None:
            raise ValueError('pool=1 geometry is not supported by the stride-2 kernels; compose conv + pool')
        self._c = cg

    def ref(self):
        return ctypes.byref(self._c)

    @property
    def in_shape(self):
        return (self.N, self.Cin, self.Hin, self.Win)

    @property
    def out_shape(self):
        return (self.N, self.Cout, self.Ho, self.Wo)


def pool_fusable(N, Cin, Hin, Win, Cout, ks, pad):
    """Whether conv(ks, pad) followed by AvgPool2d(2) can run as ONE stride-2 kernel."""
    if ks != 3 or pad != 1:
        return False
    if _COMPUTE[0] == 'bf16':
        v = ConvGeom(int(N), int(Cin), int(Hin), int(Win), int(Cout), 3, 1, 0, 1)
        if _lib.lib().ganlab_conv_bf16_supported(ctypes.byref(v)):
            return True         # the bf16 kernel sums the 2 x 2 outputs in its epilogue
    c = ConvGeom(int(N), int(Cin), int(Hin), int(Win), int(Cout), 3, 1, 0, 1)
    return bool(_lib.lib().ganlab_conv_s2_supported(ctypes.byref(c)))


# ---- packed-weight cache ------------------------------------------------------------------------
# Packing (OIHW -> [tap][ci][co], eq-LR scale folded in) costs one tiny kernel; D weights are used by 3 forward + several
# dgrad passes per step, so the result is cached until the weights change.  The entry keeps an alias of the weight alive so
# its address cannot be recycled under the key.
# Who changes weights: (i) in-place torch ops - seen through ``w._version`` in the key; (ii) the fused optimiser, which
# rewrites a whole parameter arena through raw pointers - it reports the byte ranges it touched
# (``bump_weight_epoch(ranges)``), every cached entry inside such a range is stale, and the FIRST request for one of them
# re-packs ALL of them in ONE launch from a device-resident descriptor table (csrc/pack.hip: 60-190 pack launches per
# training step before).  Entries outside the range (the other network) stay valid.  ``bump_weight_epoch()`` without
# ranges drops everything (growth events, graph capture).
class _PackEntry(object):
    __slots__ = ('w', 'out', 'serial', 'ptr', 'desc')


_PACK_CACHE = {}
_PACK_SERIAL = [0]
_PACK_GEN = [0]          # bumped when buffers a captured graph may point at are dropped: full flush, table rebuild
_PACK_RANGES = {}        # (lo, hi) byte range -> serial of its last rewrite
_PACK_TABLES = {}        # (lo, hi) -> (entry keys, device descriptor table, total blocks)
_KIND_PLAIN, _KIND_S2, _KIND_BF16, _KIND_X3 = 0, 1, 2, 3


def bump_weight_epoch(ranges=None):
    """Called by the fused optimiser after it rewrites parameters through raw pointers (``ranges``: [(lo, hi)] device
    address ranges), or without arguments to drop every packed weight."""
    _PACK_SERIAL[0] += 1
    if ranges is None:
        if _PACK_CACHE or _PACK_TABLES:      # (a flush that drops nothing leaves the generation - a step graph's signature - alone)
            _PACK_GEN[0] += 1
        _PACK_CACHE.clear()
        _PACK_TABLES.clear()
        _PACK_RANGES.clear()
        return
    for r in ranges:
        _PACK_RANGES[(int(r[0]), int(r[1]))] = _PACK_SERIAL[0]


def pack_generation():
    """Changes whenever the cache dropped buffers that an earlier reader may still hold raw pointers to (graphs.py)."""
    return _PACK_GEN[0]


_ALL_ADDRESSES = (0, 1 << 62)


def mark_packs_stale():
    """Every cached packed weight is re-packed at its next use, but the cache entries, their buffers and the descriptor
    tables of the batched re-pack stay (``bump_weight_epoch()`` without arguments drops them).  graphs.GraphedStep: a step
    graph must pack what it uses itself - with the ONE launch per parameter range of the steady state, whose table is
    already on the device - and a replay rewrites the parameters behind the host's back."""
    if _PACK_RANGES:
        bump_weight_epoch(list(_PACK_RANGES.keys()))
    elif _PACK_CACHE:
        # no range recorded yet (no optimiser step since the last flush): one catch-all range - nothing is dropped, so the
        # generation, which is part of a step graph's signature, stays: a capture or replay must not invalidate its own graphs
        bump_weight_epoch([_ALL_ADDRESSES])


def _pack_one(e):
    """Re-pack one cache entry with its own kernel (the batched launch's table cannot be uploaded while a capture runs)."""
    L = _lib.lib()
    kind, cout, cin, ks, mode, up, scale, total = e.desc
    if kind == _KIND_S2:
        rc = L.ganlab_conv_s2_pack_f32(_p(e.w), _p(e.out), cout, cin, up, mode, scale, _st())
    elif kind == _KIND_BF16:
        rc = L.ganlab_conv_pack_bf16(_p(e.w), e.out.data_ptr(), cout, cin, mode, scale, _st())
    elif kind == _KIND_X3 and ks == 4:
        rc = L.ganlab_conv_s2_x3_pack(_p(e.w), e.out.data_ptr(), cout, cin, up, scale, _st())
    elif kind == _KIND_X3 and ks == 5:
        rc = L.ganlab_conv_s2_down_x3_pack(_p(e.w), e.out.data_ptr(), cout, cin, up, scale, _st())
    elif kind == _KIND_X3:
        rc = L.ganlab_conv_x3_pack(_p(e.w), e.out.data_ptr(), cout, cin, mode, scale, _st())
    else:
        rc = L.ganlab_conv_pack_f32(_p(e.w), _p(e.out), cout, cin, ks, mode, scale, _st())
    if rc != total:
        raise _lib.GanlabLibraryError(f're-pack failed ({rc} != {total})')


def _stale_range(e):
    for r, ser in _PACK_RANGES.items():
        if ser > e.serial and r[0] <= e.ptr < r[1]:
            return r
    return None


def _repack_range(r):
    """Re-pack every cached entry whose weight lives in the rewritten range ``r`` with ONE launch."""
    L = _lib.lib()
    ser = _PACK_RANGES[r]
    items = [(k, e) for k, e in _PACK_CACHE.items() if r[0] <= e.ptr < r[1] and e.serial < ser]
    keys = tuple(k for k, _ in items)
    tab = _PACK_TABLES.get(r)
    if (tab is None or tab[0] != keys) and torch.cuda.is_current_stream_capturing():
        for _, e in items:          # no host-to-device copy inside a capture: one kernel per entry
            _pack_one(e)
            e.serial = _PACK_SERIAL[0]
        return
    if tab is None or tab[0] != keys:
        arr = (_lib.PackDesc * len(items))()
        blocks = 0
        for i, (_, e) in enumerate(items):
            kind, cout, cin, ks, mode, up, scale, total = e.desc
            d = arr[i]
            d.src, d.dst = e.w.data_ptr(), e.out.data_ptr()
            d.kind, d.Cout, d.Cin, d.ks, d.mode, d.up, d.scale, d.total, d.block0 = kind, cout, cin, ks, mode, up, scale, \
                total, blocks
            # one thread per weight position, all of its taps (csrc/pack.hip pack_many_kernel)
            blocks += (total // (16 if kind == _KIND_S2 else (9 if kind == _KIND_BF16 else ((48 if ks >= 4 else 27) if kind == _KIND_X3 else ks * ks))) + 255) // 256
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        if tab is not None:
            _PACK_GEN[0] += 1       # the old table's memory goes back to the allocator
        tab = (keys, host.to(items[0][1].w.device), blocks)
        _PACK_TABLES[r] = tab
    check(L.ganlab_pack_many(tab[1].data_ptr(), len(items), tab[2], _st()), 'pack_many')
    now = _PACK_SERIAL[0]
    for _, e in items:
        e.serial = now


def _pack_lookup(key):
    e = _PACK_CACHE.get(key)
    if e is None:
        return None
    r = _stale_range(e)
    if r is not None:
        _repack_range(r)
    return e.out


def _pack_store(key, w, out, desc):
    if len(_PACK_CACHE) > 1024:
        bump_weight_epoch()
    e = _PackEntry()
    e.w, e.out, e.serial, e.ptr, e.desc = w.detach(), out, _PACK_SERIAL[0], w.data_ptr(), desc
    _PACK_CACHE[key] = e
    # a new entry changes the entry set of its range: the table is rebuilt at the next re-pack (keys differ)
    return out


def _packed(w, mode, scale, s2_up=None):
    """mode: PACK_FWD / PACK_DGRAD.  s2_up: None -> plain [tap][ci][co] packing; 0 / 1 -> K4 packing of the
    stride-2 kernels for a down (pool) / up layer."""
    key = (w.data_ptr(), w._version, tuple(w.shape), mode, float(scale), s2_up)
    hit = _pack_lookup(key)
    if hit is not None:
        return hit
    cout, cin, ks = w.shape[0], w.shape[1], w.shape[2]
    L = _lib.lib()
    if s2_up is not None:
        tr = 1 if mode == PACK_DGRAD else 0
        n = L.ganlab_conv_s2_pack_f32(None, None, cout, cin, int(s2_up), tr, scale, None)
        if n <= 0:
            raise _lib.GanlabLibraryError(f'conv_s2_pack size query failed ({n}) for weight {tuple(w.shape)}')
        out = _new((n,), w)
        rc = L.ganlab_conv_s2_pack_f32(_p(w), _p(out), cout, cin, int(s2_up), tr, scale, _st())
        if rc != n:
            raise _lib.GanlabLibraryError(f'conv_s2_pack failed ({rc})')
        return _pack_store(key, w, out, (_KIND_S2, cout, cin, 3, tr, int(s2_up), float(scale), int(n)))
    n = L.ganlab_conv_pack_f32(None, None, cout, cin, ks, mode, scale, None)
    if n <= 0:
        raise _lib.GanlabLibraryError(f'conv_pack size query failed ({n}) for weight {tuple(w.shape)}')
    out = _new((n,), w)
    rc = L.ganlab_conv_pack_f32(_p(w), _p(out), cout, cin, ks, mode, scale, _st())
    if rc != n:
        raise _lib.GanlabLibraryError(f'conv_pack failed ({rc})')
    return _pack_store(key, w, out, (_KIND_PLAIN, cout, cin, ks, mode, 0, float(scale), int(n)))


def _packed_bf16(w, mode, scale):
    key = (w.data_ptr(), w._version, tuple(w.shape), mode, float(scale), 'bf16')
    hit = _pack_lookup(key)
    if hit is not None:
        return hit
    cout, cin = w.shape[0], w.shape[1]
    L = _lib.lib()
    n = L.ganlab_conv_pack_bf16(None, None, cout, cin, mode, scale, None)
    if n <= 0:
        raise _lib.GanlabLibraryError(f'conv_pack_bf16 size query failed ({n}) for weight {tuple(w.shape)}')
    out = torch.empty((n,), dtype=torch.bfloat16, device=w.device)
    rc = L.ganlab_conv_pack_bf16(_p(w), out.data_ptr(), cout, cin, mode, scale, _st())
    if rc != n:
        raise _lib.GanlabLibraryError(f'conv_pack_bf16 failed ({rc})')
    return _pack_store(key, w, out, (_KIND_BF16, cout, cin, 3, mode, 0, float(scale), int(n)))


# ---- fp32 convolutions as split products on the bf16 matrix cores (csrc/conv_x3.hip) --------------------------------
# Three bf16 planes per operand, six products per fp32 product, fp32 chains of one 32-channel chunk: as close to float64 as
# the exact-fp32 MFMA kernels (tools/x3_bench.py: 0.45-0.51 of ATen's rms error against 0.82-1.00) at 1.6-1.9x their speed.
# On by default for the thick plain 3x3 layers; GANLAB_X3=0 (or set_x3(False)) restores the exact-fp32 kernels everywhere.
_X3 = [os.environ.get('GANLAB_X3', '1') != '0']
_X3_MIN_TILES = 128       # (16 x 16 pixels x 64 channels) tiles below which the launch leaves most of the 256 CUs idle


def set_x3(enabled):
    """Switch the split-product kernels on / off (returns the previous setting)."""
    old = _X3[0]
    _X3[0] = bool(enabled)
    return old


def x3_enabled():
    return _X3[0]


def x3_ok(g, dgrad=False):
    """Does this conv (forward, or its input gradient) run on the split-product kernels?"""
    if not _X3[0] or g.bf is not None or g.s2 or g.up or g.pool or g.ks != 3 or g.pad != 1:
        return False
    co = g.Cin if dgrad else g.Cout
    if g.N * (g.Hin // 16) * (g.Win // 16) * (co // 64) < _X3_MIN_TILES:
        return False
    return bool(_lib.lib().ganlab_conv_x3_supported(g.ref(), 1 if dgrad else 0))


def x3_s2_ok(g, dgrad=False):
    """Does this stride-2 fused layer run on the split-product kernel's transposed form?  (An up layer's forward, a pooled
    layer's input gradient; the strided form - pooled forward, up layer's input gradient - stays on the exact kernels.)"""
    if not _X3[0] or not g.s2:
        return False
    if (g.pool if dgrad else g.up) != 1:
        return False
    co = g.Cin if dgrad else g.Cout
    hl, wl = (g.Hin // 2, g.Win // 2) if dgrad else (g.Hin, g.Win)
    if g.N * (hl // 8) * (wl // 16) * (co // 32) < _X3_MIN_TILES:       # (32 output channels: both row parities in one workgroup)
        return False
    return bool(_lib.lib().ganlab_conv_s2_x3_supported(g.ref(), 1 if dgrad else 0))


def x3_wgrad_ok(g):
    """Does this plain 3x3 layer's weight gradient run on the split-product kernel?"""
    if not _X3[0] or g.bf is not None or g.s2 or g.up or g.pool or g.ks != 3 or g.pad != 1 or g.Cin < 64 or g.Cout < 64:
        return False
    return bool(_lib.lib().ganlab_conv_wgrad_x3_supported(g.ref()))


def _wgrad_x3(gy, x, s_, t_, gw, g, scale):
    L = _lib.lib()
    ws = torch.empty((L.ganlab_conv_wgrad_x3_workspace(g.ref()) + 3) // 4, dtype=torch.float32, device=x.device)
    check(L.ganlab_conv_wgrad_x3(_p(gy), _p(x), _p(s_), _p(t_), _p(gw), g.ref(), scale, _p(ws), ws.numel() * 4, _st()),
          'conv_wgrad_x3')
    return gw


def x3_s2_wgrad_ok(g):
    """Does this stride-2 layer's weight gradient run on the split-product box-sum kernel?"""
    if not _X3[0] or g.bf is not None or not g.s2 or g.ks != 3 or g.pad != 1:
        return False
    return bool(_lib.lib().ganlab_conv_s2_wgrad_x3_supported(g.ref()))


def _wgrad_x3_s2(gy, x, s_, t_, gw, g, scale):
    L = _lib.lib()
    ws = torch.empty((L.ganlab_conv_s2_wgrad_x3_workspace(g.ref()) + 3) // 4, dtype=torch.float32, device=x.device)
    check(L.ganlab_conv_s2_wgrad_x3(_p(gy), _p(x), _p(s_), _p(t_), _p(gw), g.ref(), scale, _p(ws), ws.numel() * 4, _st()),
          'conv_s2_wgrad_x3')
    return gw


def x3_s2_down_ok(g, dgrad=False):
    """The strided stride-2 form on the split-product kernel: a pooled layer's forward, an up layer's input gradient."""
    if not _X3[0] or not g.s2:
        return False
    if (g.up if dgrad else g.pool) != 1:
        return False
    co = g.Cin if dgrad else g.Cout
    hl, wl = (g.Hin, g.Win) if dgrad else (g.Hin // 2, g.Win // 2)
    if g.N * (hl // 8) * (wl // 16) * (co // 128) < _X3_MIN_TILES:
        return False
    return bool(_lib.lib().ganlab_conv_s2_down_x3_supported(g.ref(), 1 if dgrad else 0))


def _packed_x3_s2_down(w, up, scale):
    """``up``: 0 = a pooled layer's forward weights, 1 = an up layer's input-gradient weights."""
    key = (w.data_ptr(), w._version, tuple(w.shape), int(up), float(scale), 'x3s2d')
    hit = _pack_lookup(key)
    if hit is not None:
        return hit
    cout, cin = w.shape[0], w.shape[1]
    L = _lib.lib()
    n = L.ganlab_conv_s2_down_x3_pack(None, None, cout, cin, int(up), scale, None)
    if n <= 0:
        raise _lib.GanlabLibraryError(f'conv_s2_down_x3_pack size query failed ({n}) for weight {tuple(w.shape)}')
    out = torch.empty((n,), dtype=torch.bfloat16, device=w.device)
    rc = L.ganlab_conv_s2_down_x3_pack(_p(w), out.data_ptr(), cout, cin, int(up), scale, _st())
    if rc != n:
        raise _lib.GanlabLibraryError(f'conv_s2_down_x3_pack failed ({rc})')
    return _pack_store(key, w, out, (_KIND_X3, cout, cin, 5, 0, int(up), float(scale), int(n)))


def _packed_x3_s2(w, up, scale):
    """``up``: 1 = an up layer's forward weights, 0 = a pooled layer's input-gradient weights."""
    key = (w.data_ptr(), w._version, tuple(w.shape), int(up), float(scale), 'x3s2')
    hit = _pack_lookup(key)
    if hit is not None:
        return hit
    cout, cin = w.shape[0], w.shape[1]
    L = _lib.lib()
    n = L.ganlab_conv_s2_x3_pack(None, None, cout, cin, int(up), scale, None)
    if n <= 0:
        raise _lib.GanlabLibraryError(f'conv_s2_x3_pack size query failed ({n}) for weight {tuple(w.shape)}')
    out = torch.empty((n,), dtype=torch.bfloat16, device=w.device)
    rc = L.ganlab_conv_s2_x3_pack(_p(w), out.data_ptr(), cout, cin, int(up), scale, _st())
    if rc != n:
        raise _lib.GanlabLibraryError(f'conv_s2_x3_pack failed ({rc})')
    return _pack_store(key, w, out, (_KIND_X3, cout, cin, 4, 0, int(up), float(scale), int(n)))


def _packed_x3(w, mode, scale):
    key = (w.data_ptr(), w._version, tuple(w.shape), mode, float(scale), 'x3')
    hit = _pack_lookup(key)
    if hit is not None:
        return hit
    cout, cin = w.shape[0], w.shape[1]
    L = _lib.lib()
    n = L.ganlab_conv_x3_pack(None, None, cout, cin, mode, scale, None)
    if n <= 0:
        raise _lib.GanlabLibraryError(f'conv_x3_pack size query failed ({n}) for weight {tuple(w.shape)}')
    out = torch.empty((n,), dtype=torch.bfloat16, device=w.device)
    rc = L.ganlab_conv_x3_pack(_p(w), out.data_ptr(), cout, cin, mode, scale, _st())
    if rc != n:
        raise _lib.GanlabLibraryError(f'conv_x3_pack failed ({rc})')
    return _pack_store(key, w, out, (_KIND_X3, cout, cin, 3, mode, 0, float(scale), int(n)))


# ---- "input gradient only" mode ---------------------------------------------------------------------
# torch.autograd.grad(D(x), x, create_graph=True) (the gradient-penalty pass) only needs d/dx, but a
# Python autograd.Function cannot see which of its outputs the engine will keep: without this flag
# every conv would also run its weight-gradient kernel there and throw the result away.  The flag is
# a module global (the autograd engine runs backward on its own device thread).
_INPUT_GRAD_ONLY = [False]


class input_grad_only(object):
    def __enter__(self):
        self._prev = _INPUT_GRAD_ONLY[0]
        _INPUT_GRAD_ONLY[0] = True

    def __exit__(self, *exc):
        _INPUT_GRAD_ONLY[0] = self._prev
        return False


def _want_param_grads():
    return not _INPUT_GRAD_ONLY[0]


# ---- "no gradient towards this leaf" ----------------------------------------------------------------
# The mirror case: ``loss_d.backward()`` of the D step.  The real batch is a leaf with requires_grad (R1 needs
# d D/d x with create_graph=True), so the sweep would also produce d loss/d x - one input-gradient kernel of the
# first layer over the widest tensor of the network - which nobody reads.  ``with ops.no_grad_towards(x):`` makes the
# first layer's backward skip it (a Python autograd.Function cannot see that the engine will discard an output).
_NO_GRAD_TOWARDS = [None]


class no_grad_towards(object):
    def __init__(self, leaf):
        self.ptr = leaf.data_ptr() if leaf is not None else None

    def __enter__(self):
        self._prev = _NO_GRAD_TOWARDS[0]
        _NO_GRAD_TOWARDS[0] = self.ptr

    def __exit__(self, *exc):
        _NO_GRAD_TOWARDS[0] = self._prev
        return False


def _wants_input_grad(ctx, x):
    return ctx.needs_input_grad[0] and not (_NO_GRAD_TOWARDS[0] is not None and x.grad_fn is None and
                                            x.data_ptr() == _NO_GRAD_TOWARDS[0])


# ---- parameter gradients written straight into the flat arena ---------------------------------------
# ``loss.backward()`` hands every parameter gradient to an AccumulateGrad node: one ``grad += g`` launch per parameter and
# use (~250 per StyleGAN step - a fifth of the launches of the launch-bound configurations).  Inside
# ``with direct_param_grads():`` (the learners' backward sweeps) the gradient kernels of a parameter
# that lives in a ``ParamArena`` write their result INTO its (zeroed) arena slot when it is the first contribution of this
# step, and the Function returns None for it; later contributions of the same step (the critic is applied to two batches)
# take the ordinary accumulating path.  Never active while a differentiable backward is being recorded.
_DIRECT = [False]
_SINK, _TAKEN = {}, {}


def direct_grads_enabled():
    """A/B knob GANLAB_DIRECT_GRADS=0.  The mechanism is the same with and without data parallelism: the reducer's
    post-accumulate hooks fire when a parameter's AccumulateGrad node runs - after EVERY contributing Function, direct or not -
    and a parameter whose Functions never ran (RgbHandoff) simply leaves its bucket to ``GradReducer.start``; the data-parallel
    code path on one rank (GANLAB_DIST_WORLD1=1) ends bit-identical to the plain one (tools/dist1_probe.py)."""
    import os
    return os.environ.get('GANLAB_DIRECT_GRADS') != '0'


class direct_param_grads(object):
    def __init__(self, enabled=True):
        self.enabled = bool(enabled)

    def __enter__(self):
        self._prev = _DIRECT[0]
        _DIRECT[0] = self.enabled
        _SINK.clear()
        _TAKEN.clear()

    def __exit__(self, *exc):
        _DIRECT[0] = self._prev
        _SINK.clear()
        _TAKEN.clear()
        return False


def _sink(name, p):
    """Offer parameter ``p``'s arena slot to the next launcher that allocates the output called ``name``."""
    if not _DIRECT[0] or p is None or torch.is_grad_enabled():
        return
    base = p._base if p._base is not None else p
    arena = getattr(base, '_ganlab_arena', None)
    if arena is None or not base.is_leaf or base.grad is None or base.numel() != p.numel() or \
            getattr(base, '_ganlab_written', -1) == arena.serial:
        return
    _SINK[name] = base


def _take(name, shape, like):
    """Output buffer ``name`` of a launcher: the offered arena slot (marked written for this step) or a fresh tensor."""
    if _SINK:
        base = _SINK.pop(name, None)
        if base is not None:
            n = 1
            for v in shape:
                n *= int(v)
            if base.grad.numel() == n and base.grad.device == like.device:
                base._ganlab_written = base._ganlab_arena.serial
                _TAKEN[name] = True
                return base.grad.view(shape)
    return _new(shape, like)


def _sunk(name):
    """After the launcher: did it write the parameter's arena slot (-> the Function returns None for that gradient)?"""
    if _SINK:
        _SINK.pop(name, None)
    return _TAKEN.pop(name, False) if _TAKEN else False


# ---- launch observer (measurement only) -------------------------------------------------------------
# bench.py / tools/step_layers.py count the convolution FLOPs the step actually EXECUTES (stride-2 fused layers run
# 16 low-resolution taps instead of 4 x 9, the shared D(real) forward and the skipped weight gradients never launch)
# by registering a callable here; every conv launcher reports (kind, Geom) to it.  None = no overhead.
_OBSERVER = [None]


def set_launch_observer(fn):
    prev = _OBSERVER[0]
    _OBSERVER[0] = fn
    return prev


def conv_flops(g):
    """FLOPs one pass (forward, input gradient or weight gradient) over geometry ``g`` executes in this library."""
    if g.s2:
        lo_h, lo_w = (g.Hin, g.Win) if g.up else (g.Ho, g.Wo)
        return 2.0 * 16 * g.Cin * g.Cout * lo_h * lo_w * g.N
    if g.bf is not None:       # bf16 layers run all 9 taps at the taps' resolution, also with a folded upsample / pool
        return 2.0 * g.ks * g.ks * g.Cin * g.Cout * g.bf.Hin * g.bf.Win * g.N
    return 2.0 * g.ks * g.ks * g.Cin * g.Cout * g.Ho * g.Wo * g.N


def _note(kind, g):
    if _OBSERVER[0] is not None:
        _OBSERVER[0](kind, g)


# ---------------------------------------------------------------------------------------------- #
# kernel launchers
# ---------------------------------------------------------------------------------------------- #
def k_conv_fwd(x, w, bias, g, scale, bias_scale=1.0, act=ACT_NONE, slope=0.2):
    x, w = _c(x, 'conv input'), _c(w, 'conv weight')
    assert tuple(x.shape) == g.in_shape, (tuple(x.shape), g.in_shape)
    assert tuple(w.shape) == (g.Cout, g.Cin, g.ks, g.ks), (tuple(w.shape), g.Cout, g.Cin, g.ks)
    _note('fwd', g)
    if bias is not None:
        bias = _c(bias, 'bias')
        assert bias.numel() == g.Cout
    y = _new(g.out_shape, x)
    if g.bf is not None:   # bf16-compute layer; a nearest upsample in front / average pool behind folds into the kernel
        geom = g.bf_fused if g.bf_fused is not None else g.bf
        xin = k_up2(x, 1.0) if (g.up and g.bf_fused is None) else x
        wp = _packed_bf16(w, PACK_FWD, scale)
        L = _lib.lib()
        split = L.ganlab_conv_bf16_splitk_plan(ctypes.byref(geom), 0)
        if split >= 2:     # few output tiles, long contraction (512-channel 16x16 layers at small batch)
            m = 2 if geom.up else 1
            ws = torch.empty((split * geom.N * geom.Cout * geom.Hin * m * geom.Win * m,), dtype=torch.float32, device=x.device)
            check(L.ganlab_conv_fwd_bf16_splitk(_p(xin), wp.data_ptr(), _p(bias), _p(y), ctypes.byref(geom), bias_scale, act,
                                                slope, _p(ws), ws.numel() * 4, _st()), 'conv_fwd_bf16_splitk')
            return y
        check(L.ganlab_conv_fwd_bf16(_p(xin), wp.data_ptr(), _p(bias), _p(y), ctypes.byref(geom), bias_scale,
                                     act, slope, _st()), 'conv_fwd_bf16')
        return y
    if x3_s2_down_ok(g):
        check(_lib.lib().ganlab_conv_s2_down_fwd_x3(_p(x), _packed_x3_s2_down(w, 0, scale).data_ptr(), _p(bias), _p(y), g.ref(),
                                                    bias_scale, act, slope, _st()), 'conv_s2_down_fwd_x3')
        return y
    if x3_s2_ok(g):
        check(_lib.lib().ganlab_conv_s2_fwd_x3(_p(x), _packed_x3_s2(w, 1, scale).data_ptr(), _p(bias), _p(y), g.ref(), bias_scale,
                                               act, slope, _st()), 'conv_s2_fwd_x3')
        return y
    if g.s2:   # stride-2 fused layer: conv+avgpool (S kernel) or upsample+conv (T kernel)
        wp = _packed(w, PACK_FWD, scale, s2_up=g.up)
        check(_lib.lib().ganlab_conv_s2_fwd_f32(_p(x), _p(wp), _p(bias), _p(y), g.ref(), bias_scale, act, slope,
                                                _st()), 'conv_s2_fwd')
        return y
    if g.ks == 1 and g.Hin == 1 and g.Win == 1 and g.N <= 64 and g.Cin >= 2048 and g.Cin % 128 == 0:
        # A linear layer with a long contraction and few rows (the critic's 4x4 "valid" conv: 8192 -> 512 on `batch`
        # rows): the forward kernel would walk all of K in 32 workgroups.  y[n][co] = sum_k x[n][k] w[co][k] is the
        # weight-gradient contraction with k as the pixel axis, x as a (1, N, k) "output gradient" and w as a
        # (1, Cout, k) "input" - no copies - and that kernel splits the pixel axis over ~1024 workgroups.
        g2 = Geom(1, g.Cout, g.Cin // 128, 128, g.N, 1, 0)
        obs = set_launch_observer(None)          # already reported as this forward
        try:
            y = k_conv_wgrad(x.view(g2.out_shape), w.view(g2.in_shape), g2, scale).view(g.out_shape)
        finally:
            set_launch_observer(obs)
        if bias is not None or act != ACT_NONE:
            y = k_bias_act(y, bias, None, None, bias_scale, act, slope)
        return y
    L = _lib.lib()
    if x3_ok(g):
        check(L.ganlab_conv_fwd_x3(_p(x), _packed_x3(w, PACK_FWD, scale).data_ptr(), _p(bias), _p(y), g.ref(), bias_scale, act,
                                   slope, _st()), 'conv_fwd_x3')
        return y
    wp = _packed(w, PACK_FWD, scale)
    split = 0 if g.up else L.ganlab_conv_splitk_plan(g.ref(), 0)
    if split >= 2:     # few output tiles, long contraction (512-channel 4x4 / 8x8 maps, small-batch linears)
        ws = torch.empty((split,) + tuple(y.shape), dtype=torch.float32, device=x.device)
        check(L.ganlab_conv_fwd_splitk_f32(_p(x), _p(wp), _p(bias), _p(y), g.ref(), bias_scale, act, slope, _p(ws),
                                           ws.numel() * 4, _st()), 'conv_fwd_splitk')
        return y
    check(L.ganlab_conv_fwd_f32(_p(x), _p(wp), _p(bias), _p(y), g.ref(), bias_scale, act, slope, _st()), 'conv_fwd')
    return y


def k_conv_dgrad(gy, w, g, scale):
    gy, w = _c(gy, 'conv grad_out'), _c(w, 'conv weight')
    assert tuple(gy.shape) == g.out_shape, (tuple(gy.shape), g.out_shape)
    _note('dgrad', g)
    if g.bf is not None:
        wp = _packed_bf16(w, PACK_DGRAD, scale)
        L = _lib.lib()

        def run(geom, out):
            split = L.ganlab_conv_bf16_splitk_plan(ctypes.byref(geom), 1)
            if split >= 2:
                m = 2 if geom.up else 1
                ws = torch.empty((split * geom.N * geom.Cin * geom.Hin * m * geom.Win * m,), dtype=torch.float32,
                                 device=gy.device)
                check(L.ganlab_conv_dgrad_bf16_splitk(_p(gy), wp.data_ptr(), _p(out), ctypes.byref(geom), _p(ws),
                                                      ws.numel() * 4, _st()), 'conv_dgrad_bf16_splitk')
            else:
                check(L.ganlab_conv_dgrad_bf16(_p(gy), wp.data_ptr(), _p(out), ctypes.byref(geom), _st()), 'conv_dgrad_bf16')
            return out
        if g.bf_fused is not None:     # the adjoint of the upsample (2 x 2 sum) / of the pool (upsample / 4) is in the kernel
            return run(g.bf_fused, _new(g.in_shape, gy))
        gxv = run(g.bf, _new((g.N, g.Cin, g.bf.Hin, g.bf.Win), gy))
        return k_pool2(gxv, 1.0) if g.up else gxv
    if x3_s2_down_ok(g, True):
        gx = _new(g.in_shape, gy)
        check(_lib.lib().ganlab_conv_s2_down_dgrad_x3(_p(gy), _packed_x3_s2_down(w, 1, scale).data_ptr(), _p(gx), g.ref(), _st()),
              'conv_s2_down_dgrad_x3')
        return gx
    if x3_s2_ok(g, True):
        gx = _new(g.in_shape, gy)
        check(_lib.lib().ganlab_conv_s2_dgrad_x3(_p(gy), _packed_x3_s2(w, 0, scale).data_ptr(), _p(gx), g.ref(), _st()),
              'conv_s2_dgrad_x3')
        return gx
    if g.s2:
        wp = _packed(w, PACK_DGRAD, scale, s2_up=g.up)
        gx = _new(g.in_shape, gy)
        check(_lib.lib().ganlab_conv_s2_dgrad_f32(_p(gy), _p(wp), _p(gx), g.ref(), _st()), 'conv_s2_dgrad')
        return gx
    if x3_ok(g, True):
        gx = _new(g.in_shape, gy)
        check(_lib.lib().ganlab_conv_dgrad_x3(_p(gy), _packed_x3(w, PACK_DGRAD, scale).data_ptr(), _p(gx), g.ref(), _st()),
              'conv_dgrad_x3')
        return gx
    wp = _packed(w, PACK_DGRAD, scale)
    hv, wv = (2 * g.Hin, 2 * g.Win) if g.up else (g.Hin, g.Win)
    gxv = _new((g.N, g.Cin, hv, wv), gy)
    split = 0 if g.up else _lib.lib().ganlab_conv_splitk_plan(g.ref(), 1)
    if split >= 2:
        ws = torch.empty((split,) + tuple(gxv.shape), dtype=torch.float32, device=gy.device)
        check(_lib.lib().ganlab_conv_dgrad_splitk_f32(_p(gy), _p(wp), _p(gxv), g.ref(), _p(ws), ws.numel() * 4, _st()),
              'conv_dgrad_splitk')
        return gxv
    check(_lib.lib().ganlab_conv_dgrad_f32(_p(gy), _p(wp), _p(gxv), g.ref(), _st()), 'conv_dgrad')
    if g.up:
        return k_pool2(gxv, 1.0)   # adjoint of the nearest upsample
    return gxv


def conv_dgrad_mask_ok(g):
    """The conv's input gradient can come out multiplied by lrelu'(x) (x: its input, a LeakyReLU output)."""
    return g.bf is None and not g.s2 and not g.up and bool(_lib.lib().ganlab_conv_dgrad_mask_supported(g.ref()))


def k_conv_dgrad_mask(gy, w, x, g, scale, slope):
    gy, w, x = _c(gy, 'conv grad_out'), _c(w, 'conv weight'), _c(x, 'conv input')
    assert tuple(gy.shape) == g.out_shape and tuple(x.shape) == g.in_shape
    _note('dgrad', g)
    gx = torch.empty_like(x)
    if x3_ok(g, True):
        check(_lib.lib().ganlab_conv_dgrad_mask_x3(_p(gy), _packed_x3(w, PACK_DGRAD, scale).data_ptr(), _p(x), _p(gx), g.ref(),
                                                   slope, _st()), 'conv_dgrad_mask_x3')
        return gx
    check(_lib.lib().ganlab_conv_dgrad_mask_f32(_p(gy), _p(_packed(w, PACK_DGRAD, scale)), _p(x), _p(gx), g.ref(), slope,
                                                _st()), 'conv_dgrad_mask')
    return gx


def conv_act_bwd_fusable(g):
    """The conv's LeakyReLU backward folds into its own gradient kernels (fromRGB at >= 64x64, fp32)."""
    return g.bf is None and not g.s2 and bool(_lib.lib().ganlab_conv_act_bwd_fused_supported(g.ref()))


def k_conv_dgrad_act(gy, y, w, g, scale, slope):
    """``y``: the conv's activated output, or its sign bits (``k_conv_fwd_bits``)."""
    gy, w = _c(gy, 'conv grad_out'), _c(w, 'conv weight')
    assert tuple(gy.shape) == g.out_shape
    _note('dgrad', g)
    gx = _new(g.in_shape, gy)
    if _is_bits(y):
        check(_lib.lib().ganlab_conv_dgrad_act_bits_f32(_p(gy), y.data_ptr(), _p(_packed(w, PACK_DGRAD, scale)), _p(gx),
                                                        g.ref(), slope, _st()), 'conv_dgrad_act_bits')
        return gx
    y = _c(y, 'conv output')
    assert tuple(y.shape) == g.out_shape
    check(_lib.lib().ganlab_conv_dgrad_act_f32(_p(gy), _p(y), _p(_packed(w, PACK_DGRAD, scale)), _p(gx), g.ref(), slope,
                                               _st()), 'conv_dgrad_act')
    return gx


def k_conv_fwd_mask(x, w, y, g, scale, slope):
    x, w = _c(x, 'conv input'), _c(w, 'conv weight')
    assert tuple(x.shape) == g.in_shape
    _note('fwd', g)
    out = _new(g.out_shape, x)
    if _is_bits(y):
        check(_lib.lib().ganlab_conv_fwd_mask_bits_f32(_p(x), _p(_packed(w, PACK_FWD, scale)), y.data_ptr(), _p(out), g.ref(),
                                                       slope, _st()), 'conv_fwd_mask_bits')
        return out
    y = _c(y, 'conv output')
    assert tuple(y.shape) == g.out_shape
    check(_lib.lib().ganlab_conv_fwd_mask_f32(_p(x), _p(_packed(w, PACK_FWD, scale)), _p(y), _p(out), g.ref(), slope,
                                              _st()), 'conv_fwd_mask')
    return out


def k_conv_wgrad_act(gy, y, x, g, scale, slope, bias_scale, want_gb):
    gy, x = _c(gy, 'conv grad_out'), _c(x, 'conv input')
    assert tuple(gy.shape) == g.out_shape and tuple(x.shape) == g.in_shape
    _note('wgrad', g)
    L = _lib.lib()
    gw = _take('gw', (g.Cout, g.Cin, g.ks, g.ks), x)
    gb = _take('gb', (g.Cout,), x) if want_gb else None
    ws = torch.empty((max(L.ganlab_conv_wgrad_workspace(g.ref()), 4) + 3) // 4, dtype=torch.float32, device=x.device)
    if _is_bits(y):
        check(L.ganlab_conv_wgrad_act_bits_f32(_p(gy), y.data_ptr(), _p(x), _p(gw), _p(gb), g.ref(), scale, bias_scale, slope,
                                               _p(ws), ws.numel() * 4, _st()), 'conv_wgrad_act_bits')
        return gw, gb
    y = _c(y, 'conv output')
    assert tuple(y.shape) == g.out_shape
    check(L.ganlab_conv_wgrad_act_f32(_p(gy), _p(y), _p(x), _p(gw), _p(gb), g.ref(), scale, bias_scale, slope, _p(ws),
                                      ws.numel() * 4, _st()), 'conv_wgrad_act')
    return gw, gb


def k_conv_fwd_bits(x, w, bias, g, scale, bias_scale, act, slope):
    """fromRGB forward that also writes the sign bits of its activated output (the mask of its own backward)."""
    x, w = _c(x, 'conv input'), _c(w, 'conv weight')
    assert tuple(x.shape) == g.in_shape
    _note('fwd', g)
    y = _new(g.out_shape, x)
    bits = torch.empty((y.numel() // 32,), dtype=torch.int32, device=x.device)
    check(_lib.lib().ganlab_conv_fwd_bits_f32(_p(x), _p(_packed(w, PACK_FWD, scale)), _p(_c(bias, 'bias')) if bias is not None
                                              else None, _p(y), bits.data_ptr(), g.ref(), bias_scale, act, slope, _st()),
          'conv_fwd_bits')
    return y, bits


def k_conv_fwd_blur_bits(x, w, bias, g, scale, bias_scale, slope):
    """blur(lrelu(conv3x3(x, w) * scale + bias)) and the sign bits of the un-blurred activation in ONE rolling-window kernel
    (csrc/conv_roll_blur.hip) - or None where the geometry is not the thin one it takes."""
    if g.ks != 3 or g.up or g.pool or g.bf is not None or get_compute_dtype() != 'f32' or not mask_bits_ok_plane():
        return None
    if os.environ.get('GANLAB_ROLL_BLUR') == '0':      # A/B switch: conv kernel + blur pass
        return None
    x, w = _c(x, 'conv input'), _c(w, 'conv weight')
    L = _lib.lib()
    y = _new(g.out_shape, x)
    if not L.ganlab_conv_fwd_blur_supported(g.ref(), _p(x), _p(y)):
        return None
    assert tuple(x.shape) == g.in_shape
    _note('fwd', g)
    bits = torch.empty((y.numel() // 32,), dtype=torch.int32, device=x.device)
    check(L.ganlab_conv_fwd_blur_bits_f32(_p(x), _p(_packed(w, PACK_FWD, scale)), _p(_c(bias, 'bias')) if bias is not None
                                          else None, _p(y), bits.data_ptr(), g.ref(), bias_scale, slope, _st()),
          'conv_fwd_blur_bits')
    return y, bits


def conv_s2_blur_ok(g):
    """Does the thin transposed stride-2 kernel with the blur folded in (csrc/conv_s2_roll_blur.hip) take geometry ``g`` - an
    up layer for the generator's forward tail, a pooled layer for the critic's backward?"""
    import os
    if get_compute_dtype() != 'f32' or g.bf is not None or not g.s2 or os.environ.get('GANLAB_S2_ROLL_BLUR') == '0':
        return False
    key = ('s2blur', g.N, g.Cin, g.Hin, g.Win, g.Cout, g.up, g.pool)
    hit = _AFF_OK.get(key)
    if hit is None:
        hit = bool(_lib.lib().ganlab_conv_s2_blur_supported(g.ref()))
        _AFF_OK[key] = hit
    return hit


def k_conv_s2_fwd_blur_tail(a, s_, t_, w, bias, noise, noise_w, g, scale, bias_scale, act, slope, eps):
    """(y, mean, rstd): y = act(blur(conv(up2(a * s + t), w) * scale) + noise_w * noise + bias * bias_scale) and the
    InstanceNorm statistics of y, one kernel + the statistics' finish (``s_`` / ``t_`` None: plain input)."""
    a, w = _c(a, 'conv input'), _c(w, 'conv weight')
    assert g.up and tuple(a.shape) == g.in_shape
    _note('fwd', g)
    L = _lib.lib()
    y = _new(g.out_shape, a)
    mean, rstd = _new((g.N, g.Cout), a), _new((g.N, g.Cout), a)
    ws = torch.empty((L.ganlab_conv_s2_blur_workspace(g.ref()) + 7) // 8, dtype=torch.float64, device=a.device)
    wp = _packed(w, PACK_FWD, scale, s2_up=1)
    noise = _c(noise) if noise is not None else None
    if noise is not None:
        assert noise.numel() == g.N * g.Ho * g.Wo, (tuple(noise.shape), g.out_shape)
    check(L.ganlab_conv_s2_fwd_blur_tail_f32(_p(a), _p(wp), _p(s_), _p(t_), _p(bias), _p(noise), _p(noise_w), _p(y), _p(mean),
                                             _p(rstd), g.ref(), bias_scale, act, slope, eps, ctypes.c_void_p(ws.data_ptr()),
                                             ws.numel() * 8, _st()), 'conv_s2_fwd_blur_tail')
    return y, mean, rstd


def k_conv_s2_dgrad_blur_act(gy, w, bits, g, scale, slope, bias_scale, want_gb):
    """(gz, gb) = (lrelu'(bits) * blur(dgrad(gy, w) * scale), bias_scale * sum gz): the pooled conv g's input gradient fused
    with the backward of the LeakyReLU -> blur in front of it."""
    gy, w = _c(gy, 'conv grad_out'), _c(w, 'conv weight')
    assert g.pool and tuple(gy.shape) == g.out_shape and bits.numel() * 32 == g.N * g.Cin * g.Hin * g.Win
    _note('dgrad', g)
    L = _lib.lib()
    gz = _new(g.in_shape, gy)
    gb = _new((g.Cin,), gy) if want_gb else None
    ws = torch.empty((L.ganlab_conv_s2_blur_workspace(g.ref()) + 7) // 8, dtype=torch.float64, device=gy.device) \
        if want_gb else None
    wp = _packed(w, PACK_DGRAD, scale, s2_up=0)
    check(L.ganlab_conv_s2_dgrad_blur_act_bits_f32(_p(gy), _p(wp), bits.data_ptr(), _p(gz), _p(gb), g.ref(), slope, bias_scale,
                                                   ctypes.c_void_p(ws.data_ptr()) if ws is not None else None,
                                                   ws.numel() * 8 if ws is not None else 0, _st()),
          'conv_s2_dgrad_blur_act_bits')
    return gz, gb


def k_conv_wgrad(gy, x, g, scale):
    gy, x = _c(gy, 'conv grad_out'), _c(x, 'conv input')
    assert tuple(gy.shape) == g.out_shape and tuple(x.shape) == g.in_shape
    _note('wgrad', g)
    L = _lib.lib()
    gw = _take('gw', (g.Cout, g.Cin, g.ks, g.ks), x)
    if x3_wgrad_ok(g):
        return _wgrad_x3(gy, x, None, None, gw, g, scale)
    if g.bf is not None:
        if g.bf_fused is not None:
            # the rolling-row kernel reads the half-resolution operand (x of conv(up2 x), gy of pool2(conv x)) in place
            nbytes = L.ganlab_conv_wgrad_bf16_workspace(ctypes.byref(g.bf_fused))
            if nbytes > 0:
                ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=x.device)
                check(L.ganlab_conv_wgrad_bf16(_p(gy), _p(x), _p(gw), ctypes.byref(g.bf_fused), scale, _p(ws),
                                               ws.numel() * 4, _st()), 'conv_wgrad_bf16')
                return gw
        # otherwise (16-wide taps) the weight gradient takes the materialised upsample of x (up) or of the pooled layer's
        # output gradient (pool: up2(gy) / 4, the adjoint of the average)
        xin = k_up2(x, 1.0) if g.up else x
        if g.pool:
            gy = k_up2(gy, 0.25)
        nbytes = L.ganlab_conv_wgrad_bf16_workspace(ctypes.byref(g.bf))
        if nbytes == 0:
            # 16-pixel-wide maps: bf16 forward / input gradient, but the bf16 weight-gradient kernels walk 32-pixel
            # strips - the (exact) fp32 kernel takes the materialised input
            nbytes = L.ganlab_conv_wgrad_workspace(ctypes.byref(g.bf))
            ws = torch.empty((max(nbytes, 4) + 3) // 4, dtype=torch.float32, device=x.device)
            check(L.ganlab_conv_wgrad_f32(_p(gy), _p(xin), _p(gw), ctypes.byref(g.bf), scale, _p(ws), ws.numel() * 4,
                                          _st()), 'conv_wgrad')
            return gw
        ws = torch.empty((max(nbytes, 4) + 3) // 4, dtype=torch.float32, device=x.device)
        check(L.ganlab_conv_wgrad_bf16(_p(gy), _p(xin), _p(gw), ctypes.byref(g.bf), scale, _p(ws), ws.numel() * 4,
                                       _st()), 'conv_wgrad_bf16')
        return gw
    if g.s2:
        if x3_s2_wgrad_ok(g):
            return _wgrad_x3_s2(gy, x, None, None, gw, g, scale)
        nbytes = L.ganlab_conv_s2_wgrad_workspace(g.ref())
        ws = torch.empty((max(nbytes, 4) + 3) // 4, dtype=torch.float32, device=x.device)
        check(L.ganlab_conv_s2_wgrad_f32(_p(gy), _p(x), _p(gw), g.ref(), scale, _p(ws), ws.numel() * 4, _st()),
              'conv_s2_wgrad')
        return gw
    nbytes = L.ganlab_conv_wgrad_workspace(g.ref())
    ws = torch.empty((max(nbytes, 4) + 3) // 4, dtype=torch.float32, device=x.device)
    check(L.ganlab_conv_wgrad_f32(_p(gy), _p(x), _p(gw), g.ref(), scale, _p(ws), ws.numel() * 4, _st()),
          'conv_wgrad')
    return gw


def k_blur(x):
    x = _c(x)
    n, c, h, w = x.shape
    y = torch.empty_like(x)
    check(_lib.lib().ganlab_blur3x3_f32(_p(x), _p(y), n * c, h, w, _st()), 'blur3x3')
    return y


_MASK_BITS = [None]


def mask_bits_ok(x):
    """The LeakyReLU backward of this (N,C,H,W) activation can run from its sign BITS (rows of whole 32-bit words):
    GANLAB_MASK_BITS=0 keeps the float tensor (A/B knob)."""
    if _MASK_BITS[0] is None:
        import os
        _MASK_BITS[0] = os.environ.get('GANLAB_MASK_BITS') != '0'
    return _MASK_BITS[0] and x.dim() == 4 and bool(_lib.lib().ganlab_mask_bits_supported(int(x.shape[2]), int(x.shape[3])))


def mask_bits_ok_plane():
    """The A/B knob alone (GANLAB_MASK_BITS): callers check the plane size themselves."""
    if _MASK_BITS[0] is None:
        import os
        _MASK_BITS[0] = os.environ.get('GANLAB_MASK_BITS') != '0'
    return _MASK_BITS[0]


def k_blur_bits(x):
    """(blur(x), sign bits of x): the bits are all the LeakyReLU backward of x's producer needs (int32 words, bit e of the
    NCHW-linear element index e set iff x[e] > 0)."""
    x = _c(x)
    n, c, h, w = x.shape
    y = torch.empty_like(x)
    bits = torch.empty((x.numel() // 32,), dtype=torch.int32, device=x.device)
    check(_lib.lib().ganlab_blur3x3_bits_f32(_p(x), _p(y), bits.data_ptr(), n * c, h, w, _st()), 'blur3x3_bits')
    return y, bits


def _is_bits(y):
    return y.dtype == torch.int32


def decode_u8(images_u8_nhwc, res, mean, std, flip=None):
    """Real-image input path on the device: (N,Hs,Ws,C) uint8 -> box-downsample by the
    power-of-two factor Hs/res (bit-exact with PIL's BOX resize) -> (N,C,res,res) fp32 ``(v/255 - mean)/std``
    (data_config.py:307-341).  ``flip``: optional (N,) uint8/bool mask, horizontal mirror per image."""
    x = images_u8_nhwc
    if not x.is_cuda or x.dtype != torch.uint8 or x.dim() != 4:
        raise TypeError('decode_u8 needs a (N,H,W,C) uint8 tensor on the GPU')
    x = x.contiguous()
    n, hs, ws, c = x.shape
    if hs % res or ws % res or hs // res != ws // res:
        raise ValueError(f'cannot box-resize {hs}x{ws} to {res}x{res}')
    mean = torch.as_tensor(mean, dtype=torch.float32, device=x.device).contiguous()
    std = torch.as_tensor(std, dtype=torch.float32, device=x.device).contiguous()
    assert mean.numel() == c and std.numel() == c
    if flip is not None:
        flip = flip.to(device=x.device, dtype=torch.uint8).contiguous()
        assert flip.numel() == n
    out = torch.empty((n, c, res, res), dtype=torch.float32, device=x.device)
    check(_lib.lib().ganlab_u8_box_decode_f32(x.data_ptr(), _p(out), n, hs, ws, c, hs // res, _p(mean), _p(std),
                                              flip.data_ptr() if flip is not None else None, _st()), 'u8_box_decode')
    return out


def blur_fusable(x):
    """Whether the fused blur kernels take this (N,C,H,W) map (H even, W a multiple of 4)."""
    return x.dim() == 4 and bool(_lib.lib().ganlab_blur_fused_supported(int(x.shape[2]), int(x.shape[3])))


def _blur_ws(x):
    n, c, h, w = x.shape
    nbytes = _lib.lib().ganlab_blur_fused_workspace(n, c, h, w)
    return torch.empty((max(nbytes, 4) + 3) // 4, dtype=torch.float32, device=x.device)


def k_blur_bias_act(x, bias, noise, noise_w, bias_scale, act, slope):
    x = _c(x)
    n, c, h, w = x.shape
    bias = _c(bias) if bias is not None else None
    if noise is not None:
        noise, noise_w = _c(noise), _c(noise_w)
        assert noise.numel() == n * h * w and noise_w.numel() == c
    y = torch.empty_like(x)
    check(_lib.lib().ganlab_blur_bias_act_f32(_p(x), _p(bias), _p(noise), _p(noise_w), _p(y), n, c, h, w, bias_scale,
                                              act, slope, _st()), 'blur_bias_act')
    return y


def _stats_out(x, n, c, hw):
    L = _lib.lib()
    ws = torch.empty((max(L.ganlab_act_stats_workspace(n, c, hw), 8) + 7) // 8, dtype=torch.float64, device=x.device)
    return _new((n * c,), x), _new((n * c,), x), ws


def k_blur_bias_act_stats(x, bias, noise, noise_w, bias_scale, act, slope, eps):
    """k_blur_bias_act that also returns the InstanceNorm statistics (mean, rstd) of its output."""
    x = _c(x)
    n, c, h, w = x.shape
    bias = _c(bias) if bias is not None else None
    if noise is not None:
        noise, noise_w = _c(noise), _c(noise_w)
        assert noise.numel() == n * h * w and noise_w.numel() == c
    y = torch.empty_like(x)
    mean, rstd, ws = _stats_out(x, n, c, h * w)
    check(_lib.lib().ganlab_blur_bias_act_stats_f32(_p(x), _p(bias), _p(noise), _p(noise_w), _p(y), _p(mean), _p(rstd),
                                                    n, c, h, w, bias_scale, act, slope, eps, _p(ws), ws.numel() * 8,
                                                    _st()), 'blur_bias_act_stats')
    return y, mean, rstd


def k_bias_act_stats(x, bias, noise, noise_w, bias_scale, act, slope, eps):
    x = _c(x)
    n, c, hw = _nchw(x)
    y = torch.empty_like(x)
    bias = _c(bias) if bias is not None else None
    noise = _c(noise) if noise is not None else None
    noise_w = _c(noise_w) if noise_w is not None else None
    if noise is not None:
        assert noise.numel() == n * hw and noise_w.numel() == c
    mean, rstd, ws = _stats_out(x, n, c, hw)
    check(_lib.lib().ganlab_bias_act_stats_f32(_p(x), _p(bias), _p(noise), _p(noise_w), _p(y), _p(mean), _p(rstd), n, c,
                                               hw, bias_scale, act, slope, eps, _p(ws), ws.numel() * 8, _st()),
          'bias_act_stats')
    return y, mean, rstd


def k_blur_act_bwd(g, y, slope, bias_scale, want_gb):
    """``y``: the LeakyReLU output, or its sign bits (``k_blur_bits``)."""
    g = _c(g)
    n, c, h, w = g.shape
    out = torch.empty_like(g)
    gb = _take('gb', (c,), g) if want_gb else None
    ws = _blur_ws(g) if want_gb else None
    if _is_bits(y):
        assert y.numel() * 32 == g.numel()
        check(_lib.lib().ganlab_blur_act_bwd_bits_f32(_p(g), y.data_ptr(), _p(out), _p(gb), n, c, h, w, slope, bias_scale,
                                                      _p(ws), ws.numel() * 4 if ws is not None else 0, _st()),
              'blur_act_bwd_bits')
        return out, gb
    y = _c(y)
    assert g.shape == y.shape
    check(_lib.lib().ganlab_blur_act_bwd_f32(_p(g), _p(y), _p(out), _p(gb), n, c, h, w, slope, bias_scale, _p(ws),
                                             ws.numel() * 4 if ws is not None else 0, _st()), 'blur_act_bwd')
    return out, gb


def k_act_bwd_blur(g, y, noise, slope, bias_scale, want_gb, want_gnw):
    g = _c(g)
    n, c, h, w = g.shape
    out = torch.empty_like(g)
    gb = _take('gb', (c,), g) if want_gb else None
    gnw = _take('gnw', (c,), g) if want_gnw else None
    ws = _blur_ws(g) if (want_gb or want_gnw) else None
    if _is_bits(y):
        assert y.numel() * 32 == g.numel()
        check(_lib.lib().ganlab_act_bwd_blur_bits_f32(_p(g), y.data_ptr(), _p(_c(noise)) if want_gnw else None, _p(out),
                                                      _p(gb), _p(gnw), n, c, h, w, slope, bias_scale, _p(ws),
                                                      ws.numel() * 4 if ws is not None else 0, _st()), 'act_bwd_blur_bits')
        return out, gb, gnw
    y = _c(y)
    assert g.shape == y.shape
    check(_lib.lib().ganlab_act_bwd_blur_f32(_p(g), _p(y), _p(_c(noise)) if want_gnw else None, _p(out), _p(gb),
                                             _p(gnw), n, c, h, w, slope, bias_scale, _p(ws),
                                             ws.numel() * 4 if ws is not None else 0, _st()), 'act_bwd_blur')
    return out, gb, gnw


def k_up2(x, scale=1.0):
    x = _c(x)
    n, c, h, w = x.shape
    y = _new((n, c, 2 * h, 2 * w), x)
    check(_lib.lib().ganlab_up2_f32(_p(x), _p(y), n * c, h, w, scale, _st()), 'up2')
    return y


def k_pool2(x, scale=0.25):
    x = _c(x)
    n, c, h, w = x.shape
    assert h % 2 == 0 and w % 2 == 0
    y = _new((n, c, h // 2, w // 2), x)
    check(_lib.lib().ganlab_pool2_f32(_p(x), _p(y), n * c, h // 2, w // 2, scale, _st()), 'pool2')
    return y


# ---- table-driven resamplers: nn.Upsample(mode='bilinear'), NearestPool2d, BilinearPool2d (custom_layers.py:59-75) ---
RESAMPLE_MODES = ('bilinear_up', 'bilinear_down', 'nearest_down')
_RESAMPLE_TABLES = {}


def resample_matrix(mode, align_corners, n_in):
    """1-D interpolation matrix (n_out x n_in, float64 entries that are float32 numbers) of ``F.interpolate`` at scale
    2 / 0.5, with ATen's source-index arithmetic in float32: align_corners -> src = dst * (n_in-1)/(n_out-1); else
    src = (dst + .5) / scale - .5 clamped at 0 (bilinear), floor(dst / scale) (nearest)."""
    if mode not in RESAMPLE_MODES:
        raise ValueError(f'resampler mode {mode!r}: one of {RESAMPLE_MODES}')
    f32 = np.float32
    n_out = 2 * n_in if mode == 'bilinear_up' else n_in // 2
    if n_out < 1 or (mode != 'bilinear_up' and n_in % 2):
        raise ValueError(f'{mode}: size {n_in} has no 0.5x / 2x counterpart')
    m = np.zeros((n_out, n_in), np.float64)
    inv = f32(0.5) if mode == 'bilinear_up' else f32(2.0)
    for o in range(n_out):
        if mode == 'nearest_down':
            m[o, min(int(np.floor(f32(o) * inv)), n_in - 1)] = 1.0
            continue
        if align_corners:
            r = f32(n_in - 1) / f32(n_out - 1) if n_out > 1 else f32(0)
            src = r * f32(o)
        else:
            src = max(inv * (f32(o) + f32(0.5)) - f32(0.5), f32(0))
        i0 = min(int(src), n_in - 1)
        i1 = i0 + (1 if i0 < n_in - 1 else 0)
        l1 = f32(src) - f32(i0)
        m[o, i0] += float(f32(1) - l1)
        m[o, i1] += float(l1)
    return m


def _taps(m):
    nz = [np.nonzero(row)[0] for row in m]
    t = max(1, max(len(v) for v in nz))
    if t > 6:
        raise ValueError(f'resampler needs {t} taps per output (the kernel holds 6)')
    idx, w = np.zeros((m.shape[0], t), np.int32), np.zeros((m.shape[0], t), np.float32)
    for o, v in enumerate(nz):
        idx[o, :len(v)] = v
        w[o, :len(v)] = m[o, v]
    return idx, w, t


def _resample_tables(mode, align, n_in, adjoint, device):
    key = (mode, bool(align), int(n_in), bool(adjoint), str(device))
    tab = _RESAMPLE_TABLES.get(key)
    if tab is None:
        m = resample_matrix(mode, align, n_in)
        idx, w, t = _taps(m.T if adjoint else m)
        tab = (torch.from_numpy(idx).to(device), torch.from_numpy(w).to(device), t, idx.shape[0])
        _RESAMPLE_TABLES[key] = tab
    return tab


def k_resample(x, mode, align, hin, win, adjoint):
    """``adjoint=False``: (N, C, hin, win) -> the resampled map; ``True``: a map of the resampled size -> (N, C, hin, win)
    through the transposed interpolation matrices (the layer's backward; its own backward is the forward again)."""
    x = _c(x)
    n, c, h, w = x.shape
    iy, wy, ty, ho = _resample_tables(mode, align, hin, adjoint, x.device)
    ix, wx, tx, wo = _resample_tables(mode, align, win, adjoint, x.device)
    exp_h = hin if not adjoint else (2 * hin if mode == 'bilinear_up' else hin // 2)
    exp_w = win if not adjoint else (2 * win if mode == 'bilinear_up' else win // 2)
    if (h, w) != (exp_h, exp_w):
        raise ValueError(f'resample {mode} (adjoint={adjoint}): input {h}x{w}, expected {exp_h}x{exp_w}')
    y = _new((n, c, ho, wo), x)
    check(_lib.lib().ganlab_resample2d_f32(_p(x), _p(y), _p(iy), _p(wy), _p(ix), _p(wx), n * c, h, w, ho, wo, ty, tx,
                                           _st()), 'resample2d')
    return y


def _nchw(x):
    if x.dim() == 2:
        return x.shape[0], x.shape[1], 1
    n, c = x.shape[0], x.shape[1]
    hw = 1
    for s in x.shape[2:]:
        hw *= s
    return n, c, hw


def k_bias_act(x, bias, noise, noise_w, bias_scale, act, slope):
    x = _c(x)
    n, c, hw = _nchw(x)
    y = torch.empty_like(x)
    bias = _c(bias) if bias is not None else None
    noise = _c(noise) if noise is not None else None
    noise_w = _c(noise_w) if noise_w is not None else None
    if noise is not None:
        assert noise.numel() == n * hw and noise_w.numel() == c
    check(_lib.lib().ganlab_bias_act_f32(_p(x), _p(bias), _p(noise), _p(noise_w), _p(y), n, c, hw, bias_scale, act,
                                         slope, _st()), 'bias_act')
    return y


def k_act_bwd(gy, y, slope):
    gy, y = _c(gy), _c(y)
    assert gy.shape == y.shape
    gz = torch.empty_like(gy)
    check(_lib.lib().ganlab_act_bwd_f32(_p(gy), _p(y), _p(gz), gy.numel(), slope, _st()), 'act_bwd')
    return gz


def k_act_bwd_bias(gy, y, slope, scale):
    gy, y = _c(gy), _c(y)
    assert gy.shape == y.shape
    n, c, hw = _nchw(gy)
    L = _lib.lib()
    ws = torch.empty((L.ganlab_channel_sum_workspace(n, c, hw) + 3) // 4, dtype=torch.float32, device=gy.device)
    gz, gb = torch.empty_like(gy), _take('gb', (c,), gy)
    check(L.ganlab_act_bwd_bias_f32(_p(gy), _p(y), _p(gz), _p(gb), n, c, hw, slope, scale, _p(ws), ws.numel() * 4,
                                    _st()), 'act_bwd_bias')
    return gz, gb


def k_channel_sum(a, b=None, scale=1.0, sink=None):
    a = _c(a)
    n, c, hw = _nchw(a)
    L = _lib.lib()
    nbytes = L.ganlab_channel_sum_workspace(n, c, hw)
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=a.device)
    out = _take(sink, (c,), a) if sink is not None else _new((c,), a)
    b = _c(b) if b is not None else None
    check(L.ganlab_channel_sum_f32(_p(a), _p(b), _p(out), n, c, hw, scale, _p(ws), ws.numel() * 4, _st()),
          'channel_sum')
    return out


def k_axpby(x, y, a, b, out=None):
    """a*x + b*y; ``out`` may be ``y`` itself (running averages updated in place: a captured step must write the buffer
    the next replay reads)."""
    x = _c(x)
    y = _c(y) if y is not None else None
    if y is not None:
        assert x.shape == y.shape
    if out is None:
        out = torch.empty_like(x)
    assert out.is_contiguous() and out.shape == x.shape
    check(_lib.lib().ganlab_axpby_f32(_p(x), _p(y), _p(out), x.numel(), a, b, _st()), 'axpby')
    return out


def k_scale_dev(x, gout, a, shape=None):
    gout = _c(gout)
    x = _c(x) if x is not None else None
    out = torch.empty_like(x) if x is not None else _new(shape, gout)
    check(_lib.lib().ganlab_scale_dev_f32(_p(x), _p(gout), _p(out), out.numel(), a, _st()), 'scale_dev')
    return out


def k_sum(x, scale=1.0, squared=False):
    x = _c(x)
    L = _lib.lib()
    ws = torch.empty((L.ganlab_sum_workspace(x.numel()) + 3) // 4, dtype=torch.float32, device=x.device)
    out = _new((), x)
    check(L.ganlab_sum_f32(_p(x), _p(out), x.numel(), scale, int(squared), _p(ws), ws.numel() * 4, _st()), 'sum')
    return out


def randn(shape, seed, offset, device):
    """Counter-based N(0,1) (Philox4x32-10 + Box-Muller) - replaces torch.randn for latents
    (utils/latent_utils.py:15) and per-layer noise (stylegan/architectures.py:115-116)."""
    out = torch.empty(shape, dtype=torch.float32, device=device)
    check(_lib.lib().ganlab_randn_f32(_p(out), out.numel(), int(seed) & (2 ** 64 - 1), int(offset), _st()), 'randn')
    return out


def randn_dev(shape, seed, base, delta, device):
    """``randn`` at stream position ``*base + delta`` (``base``: the step-scalar block of graphs.GraphedStep)."""
    out = torch.empty(shape, dtype=torch.float32, device=device)
    check(_lib.lib().ganlab_randn_dev_f32(_p(out), out.numel(), int(seed) & (2 ** 64 - 1), _p(base), int(delta), _st()),
          'randn_dev')
    return out


def lerp_rows(a, b, t):
    a, b, t = _c(a), _c(b), _c(t)
    out = torch.empty_like(a)
    n = a.shape[0]
    check(_lib.lib().ganlab_lerp_rows_f32(_p(a), _p(b), _p(t), _p(out), n, a.numel() // n, _st()), 'lerp_rows')
    return out


def adam_step(p, g, m, v, lr, beta1, beta2, eps, wd, bc1, bc2):
    check(_lib.lib().ganlab_adam_f32(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, wd, bc1, bc2,
                                     _st()), 'adam')


def adam_step_dev(p, g, m, v, scalars, beta1, beta2, eps, wd):
    """``adam_step`` with (lr, 1 - beta1^t, 1 - beta2^t) read from the three floats at ``scalars`` (a device pointer)."""
    check(_lib.lib().ganlab_adam_dev_f32(_p(p), _p(g), _p(m), _p(v), p.numel(), ctypes.c_void_p(scalars), beta1, beta2, eps,
                                         wd, _st()), 'adam_dev')


def set_step_scalars(block, rng_base, floats):
    f = [float(v) for v in floats] + [0.0] * (6 - len(floats))
    check(_lib.lib().ganlab_set_step_scalars(ctypes.c_void_p(block.data_ptr()), int(rng_base) & (2 ** 64 - 1), *f[:6],
                                             _st()), 'set_step_scalars')


def ewma_step(lagged, p, beta):
    check(_lib.lib().ganlab_ewma_f32(_p(lagged), _p(p), p.numel(), beta, _st()), 'ewma')


# ---------------------------------------------------------------------------------------------- #
# convolution: three mutually-recursive Functions (bilinear form  C(x, w) = s * conv(up?(x), w))
# ---------------------------------------------------------------------------------------------- #
class _ConvFwd(Function):
    @staticmethod
    def forward(ctx, x, w, g, s):
        ctx.save_for_backward(x, w)
        ctx.g, ctx.s = g, s
        return k_conv_fwd(x, w, None, g, s)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx = _ConvDgrad.apply(gy, w, ctx.g, ctx.s) if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1] and _want_param_grads():
            _sink('gw', w)
            gw = _ConvWgrad.apply(gy, x, ctx.g, ctx.s)
            if _sunk('gw'):
                gw = None
        return gx, gw, None, None


class _ConvDgrad(Function):
    @staticmethod
    def forward(ctx, gy, w, g, s):
        ctx.save_for_backward(gy, w)
        ctx.g, ctx.s = g, s
        return k_conv_dgrad(gy, w, g, s)

    @staticmethod
    def backward(ctx, ggx):
        gy, w = ctx.saved_tensors
        d_gy = _ConvFwd.apply(ggx, w, ctx.g, ctx.s) if ctx.needs_input_grad[0] else None
        d_w = _ConvWgrad.apply(gy, ggx, ctx.g, ctx.s) if ctx.needs_input_grad[1] else None
        return d_gy, d_w, None, None


class _ConvDgradMask(Function):
    """gx = dgrad(gy, w) * lrelu'(x) in the dgrad kernel's epilogue: x, the conv's input, is the LeakyReLU output of the
    layer in front, whose backward then receives the gradient of its PRE-activation (see ``conv2d(defer_act_grad=)``)."""

    @staticmethod
    def forward(ctx, gy, w, x, g, s, slope):
        ctx.save_for_backward(gy, w, x)
        ctx.g, ctx.s, ctx.slope = g, s, slope
        return k_conv_dgrad_mask(gy, w, x, g, s, slope)

    @staticmethod
    def backward(ctx, ggx):
        gy, w, x = ctx.saved_tensors
        t = _ActBwd.apply(ggx, x, ctx.slope)
        d_gy = _ConvFwd.apply(t, w, ctx.g, ctx.s) if ctx.needs_input_grad[0] else None
        d_w = _ConvWgrad.apply(gy, t, ctx.g, ctx.s) if ctx.needs_input_grad[1] else None
        return d_gy, d_w, None, None, None, None


class _DeferredActGrad(Function):
    """Identity whose backward applies lrelu'(x): the explicit form of what _ConvDgradMask folds into the dgrad kernel,
    for consumers of a ``defer_act_grad`` tensor that cannot fold it in."""

    @staticmethod
    def forward(ctx, x, slope):
        ctx.save_for_backward(x)
        ctx.slope = slope
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        x, = ctx.saved_tensors
        return _ActBwd.apply(g, x, ctx.slope), None


class _ConvWgrad(Function):
    @staticmethod
    def forward(ctx, gy, x, g, s):
        ctx.save_for_backward(gy, x)
        ctx.g, ctx.s = g, s
        return k_conv_wgrad(gy, x, g, s)

    @staticmethod
    def backward(ctx, ggw):
        gy, x = ctx.saved_tensors
        d_gy = _ConvFwd.apply(x, ggw, ctx.g, ctx.s) if ctx.needs_input_grad[0] else None
        d_x = _ConvDgrad.apply(gy, ggw, ctx.g, ctx.s) if ctx.needs_input_grad[1] else None
        return d_gy, d_x, None, None


# The three maps below fold m = lrelu'(y) (y: the conv's activated output, a constant of all of them) into the conv
# gradient kernels; they are each other's derivatives, so R1 / WGAN-GP second-order sweeps stay on fused kernels.
class _ConvDgradAct(Function):
    """gx = dgrad(gy * m, w)."""

    @staticmethod
    def forward(ctx, gy, y, w, g, s, slope):
        ctx.save_for_backward(gy, y, w)
        ctx.g, ctx.s, ctx.slope = g, s, slope
        return k_conv_dgrad_act(gy, y, w, g, s, slope)

    @staticmethod
    def backward(ctx, ggx):
        gy, y, w = ctx.saved_tensors
        d_gy = _ConvFwdMask.apply(ggx, w, y, ctx.g, ctx.s, ctx.slope) if ctx.needs_input_grad[0] else None
        d_w = _ConvWgradAct.apply(gy, y, ggx, ctx.g, ctx.s, ctx.slope, 1.0, False)[0] if ctx.needs_input_grad[2] \
            else None
        return d_gy, None, d_w, None, None, None


class _ConvFwdMask(Function):
    """out = conv(x, w) * m."""

    @staticmethod
    def forward(ctx, x, w, y, g, s, slope):
        ctx.save_for_backward(x, w, y)
        ctx.g, ctx.s, ctx.slope = g, s, slope
        return k_conv_fwd_mask(x, w, y, g, s, slope)

    @staticmethod
    def backward(ctx, go):
        x, w, y = ctx.saved_tensors
        d_x = _ConvDgradAct.apply(go, y, w, ctx.g, ctx.s, ctx.slope) if ctx.needs_input_grad[0] else None
        d_w = _ConvWgradAct.apply(go, y, x, ctx.g, ctx.s, ctx.slope, 1.0, False)[0] if ctx.needs_input_grad[1] else None
        return d_x, d_w, None, None, None, None


class _ConvWgradAct(Function):
    """(gw, gb) = (wgrad(gy * m, x), bias_scale * sum(gy * m))."""

    @staticmethod
    def forward(ctx, gy, y, x, g, s, slope, bias_scale, want_gb):
        ctx.save_for_backward(gy, y, x)
        ctx.g, ctx.s, ctx.slope = g, s, slope
        gw, gb = k_conv_wgrad_act(gy, y, x, g, s, slope, bias_scale, want_gb)
        if gb is None:
            gb = gw.new_zeros(())
            ctx.mark_non_differentiable(gb)
        return gw, gb

    @staticmethod
    def backward(ctx, ggw, ggb):
        if ggb is not None and ggb.numel() > 1 and bool(ggb.ne(0).any()):
            raise NotImplementedError('second derivative through the fused fromRGB bias gradient')
        gy, y, x = ctx.saved_tensors
        d_gy = _ConvFwdMask.apply(x, ggw, y, ctx.g, ctx.s, ctx.slope) if ctx.needs_input_grad[0] else None
        d_x = _ConvDgradAct.apply(gy, y, ggw, ctx.g, ctx.s, ctx.slope) if ctx.needs_input_grad[2] else None
        return d_gy, None, d_x, None, None, None, None, None


BLUR_HANDOFF = '_ganlab_blur_handoff'


class BlurHandoff(object):
    """Link between the critic's  conv -> LeakyReLU -> blur  layer (A) and the pooled conv (B) that is its only reader
    (progan/architectures.py:254-284): B's input-gradient kernel can apply A's blur^T and LeakyReLU derivative and sum A's bias
    gradient on the way (csrc/conv_s2_roll_blur.hip).  A fills in what that takes at forward time; B's backward sets ``done``
    and ``gb``, A's backward - which always runs after it - then takes the incoming gradient as that of its pre-activation."""
    __slots__ = ('bits', 'slope', 'bias_scale', 'want_gb', 'gb', 'done')

    def __init__(self):
        self.bits, self.slope, self.bias_scale, self.want_gb, self.gb, self.done = None, 0.2, 1.0, False, None, False


RGB_HANDOFF = '_ganlab_rgb_handoff'


class RgbHandoff(object):
    """Link between the critic's fromRGB layer (1x1 conv of the image + LeakyReLU, progan/architectures.py:232-237) and the 3x3
    conv that is its only reader: where nobody needs fromRGB's input gradient (the critic steps on detached fakes, and on the
    real batch under ``no_grad_towards``), the reader's input-gradient kernel can finish fromRGB's backward itself - mask by
    the sign bits, weight / bias gradient sums against the image - and never write the 2 GiB gradient tensor between them
    (csrc/conv_roll_blur.hip, RB_RGB).  Only inside ``direct_param_grads`` (the sums go straight into the arena slots)."""
    __slots__ = ('bits', 'slope', 'img', 'w', 'bias', 'scale', 'bias_scale')

    def __init__(self):
        self.bits = None


def _rgb_fold_ok(h, g):
    """May the input gradient of the conv with geometry ``g`` swallow the backward of the fromRGB layer ``h`` describes?"""
    import os
    if h is None or h.bits is None or torch.is_grad_enabled() or not _DIRECT[0] or not _want_param_grads() or \
            os.environ.get('GANLAB_RGB_FOLD') == '0' or get_compute_dtype() != 'f32':
        return False
    x = h.img
    if x.requires_grad and not (_NO_GRAD_TOWARDS[0] is not None and x.grad_fn is None and x.data_ptr() == _NO_GRAD_TOWARDS[0]):
        return False            # somebody wants d / d image: the gradient tensor has to exist
    for p_ in (h.w, h.bias):
        if p_ is None:
            continue
        base = p_._base if p_._base is not None else p_
        if not base.requires_grad or getattr(base, '_ganlab_arena', None) is None or base.grad is None or \
                base.numel() != p_.numel():
            return False
    return bool(_lib.lib().ganlab_conv_dgrad_rgb_sums_supported(g.ref(), int(x.shape[1])))


def k_conv_dgrad_rgb_sums(gz, w, h, g, scale):
    """Input gradient of the 3x3 conv ``g`` on ``gz`` folded with fromRGB's backward: fromRGB's weight / bias gradients are
    added into their arena slots; nothing else is written."""
    gz, w = _c(gz, 'conv grad_out'), _c(w, 'conv weight')
    assert tuple(gz.shape) == g.out_shape
    _note('dgrad', g)
    L = _lib.lib()

    def slot(p_):
        base = p_._base if p_._base is not None else p_
        arena = base._ganlab_arena
        acc = getattr(base, '_ganlab_written', -1) == arena.serial
        base._ganlab_written = arena.serial
        return base.grad, int(acc)
    gw, acc_w = slot(h.w)
    gb, acc_b = slot(h.bias) if h.bias is not None else (None, 0)
    img = _c(h.img, 'image')
    ws = torch.empty((L.ganlab_conv_dgrad_rgb_sums_workspace(g.ref()) + 7) // 8, dtype=torch.float64, device=gz.device)
    check(L.ganlab_conv_dgrad_rgb_sums_f32(_p(gz), _p(_packed(w, PACK_DGRAD, scale)), h.bits.data_ptr(), _p(img), _p(gw), _p(gb),
                                           g.ref(), int(img.shape[1]), h.scale, h.bias_scale, h.slope, acc_w, acc_b,
                                           ctypes.c_void_p(ws.data_ptr()), ws.numel() * 8, _st()), 'conv_dgrad_rgb_sums')


class _ConvDgradBlurAct(Function):
    """(gzA, gbA) = (lrelu'(bits) * blur(dgrad_B(gy, w)), bias_scale * sum gzA) in one kernel; linear in gy and in w.  Its
    derivative is the composition the separate passes have: the adjoint of blur^T o mask (``_ActBwdBlur``), then the pooled
    conv's forward / weight gradient."""

    @staticmethod
    def forward(ctx, gy, w, bits, g, s, slope, bias_scale, want_gb):
        ctx.save_for_backward(gy, w, bits)
        ctx.set_materialize_grads(False)
        ctx.g, ctx.s, ctx.slope = g, s, slope
        gz, gb = k_conv_s2_dgrad_blur_act(gy, w, bits, g, s, slope, bias_scale, want_gb)
        if gb is None:
            gb = gz.new_zeros(())
            ctx.mark_non_differentiable(gb)
        return gz, gb

    @staticmethod
    def backward(ctx, gg, ggb):
        if ggb is not None:
            raise NotImplementedError('second derivative through the fused bias gradient of the blurred layer')
        gy, w, bits = ctx.saved_tensors
        if gg is None:
            return (None,) * 8
        t = _ActBwdBlur.apply(gg, bits, None, ctx.slope, 1.0, False, False)[0]
        d_gy = _ConvFwd.apply(t, w, ctx.g, ctx.s) if ctx.needs_input_grad[0] else None
        d_w = _ConvWgrad.apply(gy, t, ctx.g, ctx.s) if ctx.needs_input_grad[1] else None
        return d_gy, d_w, None, None, None, None, None, None


class _ActBwd(Function):
    """gz = gy * lrelu'(y) from the saved OUTPUT y (sign(y) == sign(pre-activation))."""

    @staticmethod
    def forward(ctx, gy, y, slope):
        ctx.save_for_backward(y)
        ctx.slope = slope
        return k_act_bwd(gy, y, slope)

    @staticmethod
    def backward(ctx, gg):
        y, = ctx.saved_tensors
        return _ActBwd.apply(gg, y, ctx.slope), None, None


class _ActBwdBias(Function):
    """(gz, gb) = (gy * lrelu'(y), bias_scale * sum_{n,hw} gz) in one pass over the tensors."""

    @staticmethod
    def forward(ctx, gy, y, slope, bias_scale):
        ctx.save_for_backward(y)
        ctx.slope, ctx.bias_scale = slope, bias_scale
        gz, gb = k_act_bwd_bias(gy, y, slope, bias_scale)
        return gz, gb

    @staticmethod
    def backward(ctx, ggz, ggb):
        y, = ctx.saved_tensors
        g = ggz
        if ggb is not None:
            shape = y.shape
            view = [1, shape[1]] + [1] * (len(shape) - 2)
            b = _Scale.apply(ggb.view(view).expand(shape), ctx.bias_scale)
            g = b if g is None else _Axpby.apply(g, b, 1.0, 1.0)
        return (_ActBwd.apply(g, y, ctx.slope) if g is not None else None), None, None, None


class _BlurActBwd(Function):
    """(out, gb) = (lrelu'(y) * blur(g), bias_scale * sum_{n,hw} out): the backward of  conv+bias+LeakyReLU
    -> blur  in one pass.  Linear in g; its adjoint is _ActBwdBlur."""

    @staticmethod
    def forward(ctx, g, y, slope, bias_scale, want_gb):
        ctx.save_for_backward(y)          # the LeakyReLU output, or its sign bits (k_blur_bits)
        ctx.set_materialize_grads(False)
        ctx.slope, ctx.bias_scale, ctx.gshape = slope, bias_scale, g.shape
        return k_blur_act_bwd(g, y, slope, bias_scale, want_gb)

    @staticmethod
    def backward(ctx, gout, ggb):
        y, = ctx.saved_tensors
        g = gout
        if ggb is not None:
            shape = ctx.gshape
            b = _Scale.apply(ggb.view(1, shape[1], 1, 1).expand(shape), ctx.bias_scale)
            g = b if g is None else _Axpby.apply(g, b, 1.0, 1.0)
        if g is None:
            return None, None, None, None, None
        return _ActBwdBlur.apply(g, y, None, ctx.slope, 1.0, False, False)[0], None, None, None, None


class _ActBwdBlur(Function):
    """(out, gb, gnw) = (blur(z), bias_scale*sum z, sum z*noise) with z = lrelu'(y) * g: the backward of
    blur -> noise/bias/LeakyReLU in one pass.  Linear in g; its adjoint is _BlurActBwd."""

    @staticmethod
    def forward(ctx, g, y, noise, slope, bias_scale, want_gb, want_gnw):
        ctx.save_for_backward(y)
        ctx.set_materialize_grads(False)
        ctx.slope = slope
        return k_act_bwd_blur(g, y, noise, slope, bias_scale, want_gb, want_gnw)

    @staticmethod
    def backward(ctx, gout, ggb, ggnw):
        if ggb is not None or ggnw is not None:
            raise NotImplementedError('double backward through the bias / noise-weight gradients of the fused blur')
        y, = ctx.saved_tensors
        if gout is None:
            return (None,) * 7
        return (_BlurActBwd.apply(gout, y, ctx.slope, 1.0, False)[0],) + (None,) * 6


class _BlurBiasAct(Function):
    """y = act(blur(x) + noise_w*noise + bias*bias_scale): the blur after a generator up-conv fused with
    StyleAddNoise / Conv2dBias / LeakyReLU (stylegan/architectures.py:331-360 + :105-119)."""

    @staticmethod
    def forward(ctx, x, bias, noise, noise_w, bias_scale, act, slope, stats_eps=None):
        stats = None
        if stats_eps is None:
            y = k_blur_bias_act(x, bias, noise, noise_w, bias_scale, act, slope)
        else:    # + the InstanceNorm statistics of y in the same pass (handed to instnorm_style as constants)
            y, mean, rstd = k_blur_bias_act_stats(x, bias, noise, noise_w, bias_scale, act, slope, stats_eps)
            ctx.mark_non_differentiable(mean, rstd)
            stats = (mean, rstd)
        ctx.save_for_backward(y if act != ACT_NONE else None, noise)
        ctx.bias_scale, ctx.act, ctx.slope = bias_scale, act, slope
        ctx.bias_shape = bias.shape if bias is not None else None
        ctx.nw_shape = noise_w.shape if noise_w is not None else None
        ctx.bias_ref, ctx.nw_ref = bias, noise_w
        return y if stats is None else (y,) + stats

    @staticmethod
    def backward(ctx, gy, *_):
        y, noise = ctx.saved_tensors
        params = _want_param_grads()
        want_b = ctx.bias_shape is not None and ctx.needs_input_grad[1] and params
        want_nw = ctx.nw_shape is not None and ctx.needs_input_grad[3] and params
        act = ctx.act != ACT_NONE
        _sink('gb', ctx.bias_ref if want_b else None)
        _sink('gnw', ctx.nw_ref if want_nw else None)
        gx, gb, gnw = _ActBwdBlur.apply(gy, y if act else gy, noise if want_nw else None,
                                        ctx.slope if act else 1.0, ctx.bias_scale, bool(want_b), bool(want_nw))
        if _sunk('gb'):
            want_b = False
        if _sunk('gnw'):
            want_nw = False
        return (gx if ctx.needs_input_grad[0] else None), (gb.view(ctx.bias_shape) if want_b else None), None, \
            (gnw.view(ctx.nw_shape) if want_nw else None), None, None, None, None


class _ChanSum(Function):
    """(N,C,...) -> (C,) sum, optionally weighted by a (N,1,...) map (noise-weight gradient)."""

    @staticmethod
    def forward(ctx, a, b, scale, sink=None):
        ctx.shape, ctx.scale = a.shape, scale
        ctx.has_b = b is not None
        return k_channel_sum(a, b, scale, sink)

    @staticmethod
    def backward(ctx, g):
        if ctx.has_b:
            raise NotImplementedError('double backward through the noise-weighted channel sum')
        shape = ctx.shape
        view = [1, shape[1]] + [1] * (len(shape) - 2)
        return _Scale.apply(g.view(view).expand(shape), ctx.scale), None, None, None


class _GroupBroadcast(Function):
    """(G,) -> (G*gs, 1, h, w): every sample of group g gets stat[g] on its whole plane (the minibatch-stddev feature map,
    custom_layers.py:135-139).  Its adjoint is a per-group sum: a channel-sum launch instead of the ATen reduction that
    autograd derives for ``expand``; the pair is closed under differentiation (R1 differentiates through it)."""

    @staticmethod
    def forward(ctx, stat, gs, h, w):
        ctx.dims = (stat.shape[0], gs, h, w)
        G = stat.shape[0]
        return stat.view(G, 1, 1, 1, 1).expand(G, gs, 1, h, w).reshape(G * gs, 1, h, w)

    @staticmethod
    def backward(ctx, g):
        G, gs, h, w = ctx.dims
        return _ChanSum.apply(g.reshape(1, G, gs * h * w), None, 1.0), None, None, None


class _ConvBiasAct(Function):
    """y = act(s*conv(up?(x), w) + bias*bias_scale) in ONE kernel (bias/LeakyReLU in the MFMA
    epilogue).  Reference: Conv2dEx.forward (+ nn.LeakyReLU) custom_layers.py:202-211."""

    @staticmethod
    def forward(ctx, x, w, bias, g, s, bias_scale, act, slope, blur=False, defer=False, in_slope=None, handoff_out=None,
                handoff_in=None, rgb_out=None, rgb_in=None):
        # defer: the (single) consumer of y applies this layer's lrelu'(y) to the gradient it sends back (its dgrad
        # epilogue), so backward takes gy as the pre-activation gradient.  in_slope: this conv IS such a consumer.
        ctx.g, ctx.s, ctx.bias_scale, ctx.act, ctx.slope, ctx.blur = g, s, bias_scale, act, slope, blur
        ctx.bias_ref = bias
        # an undefined incoming gradient stays undefined (RgbHandoff: the reader of a fromRGB layer may have finished this
        # layer's backward itself and sends nothing; materialised, that would be a 2 GiB tensor of zeros and two passes over it)
        ctx.set_materialize_grads(False)
        # handoff_out: this is layer A of a BlurHandoff pair (filled in below when the sign bits exist); handoff_in: layer B
        ctx.handoff_out, ctx.handoff_in = None, handoff_in
        ctx.rgb_in = rgb_in         # this conv reads a fromRGB layer's output (RgbHandoff)
        if act != ACT_NONE and not blur and not defer and in_slope is None and conv_act_bwd_fusable(g) and \
                (g.Ho * g.Wo) % 32 == 0 and mask_bits_ok_plane():
            # fromRGB: the gradient kernels take (gy, mask of y); the forward writes that mask as bits next to y
            y, bits = k_conv_fwd_bits(x, w, bias, g, s, bias_scale, act, slope)
            ctx.defer, ctx.in_slope = False, None
            ctx.bias_shape = bias.shape if bias is not None else None
            ctx.save_for_backward(x, w, bits)
            if rgb_out is not None and g.ks == 1:       # fromRGB: its only reader may finish this layer's backward
                rgb_out.bits, rgb_out.slope, rgb_out.img, rgb_out.w, rgb_out.bias = bits, slope, x, w, bias
                rgb_out.scale, rgb_out.bias_scale = s, bias_scale
            return y
        ctx.defer, ctx.in_slope = bool(defer), in_slope
        assert not (defer and blur)
        ctx.bias_shape = bias.shape if bias is not None else None
        if blur and act == ACT_LRELU:
            # thin layer: conv + bias + LeakyReLU + blur + the sign bits in one rolling-window kernel
            fused = k_conv_fwd_blur_bits(x, w, bias, g, s, bias_scale, slope)
            if fused is not None:
                ctx.save_for_backward(x, w, fused[1])
                ctx.handoff_out = _fill_handoff(handoff_out, fused[1], slope, bias_scale, bias)
                return fused[0]
        y = k_conv_fwd(x, w, bias, g, s, bias_scale, act, slope)
        # blur=True: the D block's  conv -> bias -> LeakyReLU -> blur  (progan/architectures.py:280-293);
        # forward is conv kernel + blur kernel, backward is ONE pass (blur^T, LeakyReLU', bias gradient)
        if blur and act != ACT_NONE and mask_bits_ok(y):
            # the backward only ever needs sign(y): the blur pass emits it as bits and y itself is not kept (a 2 GiB
            # tensor per pass at the top of the critic; the backward passes read 1/32 of its bytes)
            out, bits = k_blur_bits(y)
            ctx.save_for_backward(x, w, bits)
            ctx.handoff_out = _fill_handoff(handoff_out, bits, slope, bias_scale, bias)
            return out
        ctx.save_for_backward(x, w, y if (act != ACT_NONE and not defer) else None)
        return k_blur(y) if blur else y

    @staticmethod
    def backward(ctx, gy):
        if gy is None:
            return (None,) * 15
        x, w, y = ctx.saved_tensors
        params = _want_param_grads()
        want_b = ctx.bias_shape is not None and ctx.needs_input_grad[2] and params
        gb = None
        act = ACT_NONE if ctx.defer else ctx.act
        if act != ACT_NONE and not ctx.blur and ctx.in_slope is None and conv_act_bwd_fusable(ctx.g):
            # fromRGB: no separate  gz = gy * lrelu'(y)  pass - its gradient kernels take (gy, y)
            gx = _ConvDgradAct.apply(gy, y, w, ctx.g, ctx.s, ctx.slope) if _wants_input_grad(ctx, x) else None
            gw = None
            if params and (ctx.needs_input_grad[1] or want_b):
                _sink('gw', w if ctx.needs_input_grad[1] else None)
                _sink('gb', ctx.bias_ref if want_b else None)
                gw, gb = _ConvWgradAct.apply(gy, y, x, ctx.g, ctx.s, ctx.slope, ctx.bias_scale, bool(want_b))
                if _sunk('gw'):
                    gw = None
                if _sunk('gb'):
                    want_b = False
            return gx, (gw if ctx.needs_input_grad[1] else None), (gb.view(ctx.bias_shape) if want_b else None), \
                None, None, None, None, None, None, None, None, None, None, None, None
        h = ctx.handoff_out
        if h is not None and h.done:
            # the pooled conv behind the blur already applied blur^T and this layer's LeakyReLU derivative in its input-
            # gradient kernel (and summed the bias gradient): gy IS the pre-activation gradient
            gz, gb = gy, (h.gb if want_b else None)
            h.done, h.gb = False, None
        elif ctx.blur and act != ACT_NONE:
            _sink('gb', ctx.bias_ref if want_b else None)
            gz, gb = _BlurActBwd.apply(gy, y, ctx.slope, ctx.bias_scale, bool(want_b))
        else:
            _sink('gb', ctx.bias_ref if want_b else None)
            if ctx.blur:
                gy = _Blur.apply(gy)
            if act != ACT_NONE and want_b:
                gz, gb = _ActBwdBias.apply(gy, y, ctx.slope, ctx.bias_scale)
            else:
                gz = _ActBwd.apply(gy, y, ctx.slope) if act != ACT_NONE else gy
                if want_b:
                    gb = _ChanSum.apply(gz, None, ctx.bias_scale, 'gb')
        if _sunk('gb'):
            gb = None
        gx = None
        if _wants_input_grad(ctx, x) and ctx.in_slope is None and ctx.handoff_in is None and _rgb_fold_ok(ctx.rgb_in, ctx.g):
            # x is fromRGB's output and nobody needs fromRGB's input gradient: this conv's input-gradient kernel sums
            # fromRGB's weight / bias gradients into their arena slots; the gradient tensor in between is never written
            k_conv_dgrad_rgb_sums(gz, w, ctx.rgb_in, ctx.g, ctx.s)
        elif _wants_input_grad(ctx, x):
            hin = ctx.handoff_in
            if hin is not None:      # layer B of the pair: input gradient + layer A's blur^T, LeakyReLU', bias gradient
                gx, gba = _ConvDgradBlurAct.apply(gz, w, hin.bits, ctx.g, ctx.s, hin.slope, hin.bias_scale,
                                                  bool(hin.want_gb and params))
                hin.gb, hin.done = (gba if (hin.want_gb and params) else None), True
            else:
                gx = _ConvDgrad.apply(gz, w, ctx.g, ctx.s) if ctx.in_slope is None else \
                    _ConvDgradMask.apply(gz, w, x, ctx.g, ctx.s, ctx.in_slope)
        gw = None
        if ctx.needs_input_grad[1] and params:
            _sink('gw', w)
            gw = _ConvWgrad.apply(gz, x, ctx.g, ctx.s)
            if _sunk('gw'):
                gw = None
        return gx, gw, (gb.view(ctx.bias_shape) if want_b and gb is not None else None), None, None, None, None, \
            None, None, None, None, None, None, None, None


def _fill_handoff(h, bits, slope, bias_scale, bias):
    if h is not None:
        h.bits, h.slope, h.bias_scale = bits, slope, bias_scale
        h.want_gb = bias is not None and bias.requires_grad
        h.gb, h.done = None, False
    return h


class _BiasAct(Function):
    """y = act(x + noise_w*noise + bias*bias_scale)  (Conv2dBias + StyleAddNoise + LeakyReLU:
    custom_layers.py:213-226, stylegan/architectures.py:105-119)."""

    @staticmethod
    def forward(ctx, x, bias, noise, noise_w, bias_scale, act, slope, stats_eps=None):
        stats = None
        if stats_eps is None:
            y = k_bias_act(x, bias, noise, noise_w, bias_scale, act, slope)
        else:
            y, mean, rstd = k_bias_act_stats(x, bias, noise, noise_w, bias_scale, act, slope, stats_eps)
            ctx.mark_non_differentiable(mean, rstd)
            stats = (mean, rstd)
        ctx.save_for_backward(y if act != ACT_NONE else None, noise)
        ctx.bias_scale, ctx.act, ctx.slope = bias_scale, act, slope
        ctx.bias_shape = bias.shape if bias is not None else None
        ctx.nw_shape = noise_w.shape if noise_w is not None else None
        ctx.bias_ref, ctx.nw_ref = bias, noise_w
        return y if stats is None else (y,) + stats

    @staticmethod
    def backward(ctx, gy, *_):
        y, noise = ctx.saved_tensors
        params = _want_param_grads()
        want_b = ctx.bias_shape is not None and ctx.needs_input_grad[1] and params
        gb = gnw = None
        _sink('gb', ctx.bias_ref if want_b else None)
        if ctx.act != ACT_NONE and want_b:
            gz, gb = _ActBwdBias.apply(gy, y, ctx.slope, ctx.bias_scale)
        else:
            gz = _ActBwd.apply(gy, y, ctx.slope) if ctx.act != ACT_NONE else gy
            if want_b:
                gb = _ChanSum.apply(gz, None, ctx.bias_scale, 'gb')
        if _sunk('gb'):
            gb = None
        if ctx.nw_shape is not None and ctx.needs_input_grad[3] and params:
            _sink('gnw', ctx.nw_ref)
            gnw = _ChanSum.apply(gz, noise, 1.0, 'gnw').view(ctx.nw_shape)
            if _sunk('gnw'):
                gnw = None
        return (gz if ctx.needs_input_grad[0] else None), (gb.view(ctx.bias_shape) if gb is not None else None), \
            None, gnw, None, None, None, None


class _Blur(Function):
    @staticmethod
    def forward(ctx, x):
        return k_blur(x)

    @staticmethod
    def backward(ctx, g):
        return _Blur.apply(g)  # symmetric taps + zero padding -> self-adjoint


class _Pool2(Function):
    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        return k_pool2(x, scale)

    @staticmethod
    def backward(ctx, g):
        return _Up2.apply(g, ctx.scale), None


class _Up2(Function):
    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        return k_up2(x, scale)

    @staticmethod
    def backward(ctx, g):
        return _Pool2.apply(g, ctx.scale), None


class _Resample(Function):
    """Linear map x -> M_y x M_x^T; backward is the adjoint gather, whose backward is the forward (R1 reaches the critic's
    pooler through a double backward)."""

    @staticmethod
    def forward(ctx, x, mode, align, hin, win, adjoint):
        ctx.args = (mode, align, hin, win, adjoint)
        return k_resample(x, mode, align, hin, win, adjoint)

    @staticmethod
    def backward(ctx, g):
        mode, align, hin, win, adjoint = ctx.args
        return _Resample.apply(g, mode, align, hin, win, not adjoint), None, None, None, None, None


class _Scale(Function):
    @staticmethod
    def forward(ctx, x, a):
        ctx.a = a
        return k_axpby(x, None, a, 0.0)

    @staticmethod
    def backward(ctx, g):
        return _Scale.apply(g, ctx.a), None


class _Axpby(Function):
    """out = a*x + b*y (fade-in blends, progan/architectures.py:163-167, :311-315)."""

    @staticmethod
    def forward(ctx, x, y, a, b):
        ctx.a, ctx.b = a, b
        return k_axpby(x, y, a, b)

    @staticmethod
    def backward(ctx, g):
        gx = _Scale.apply(g, ctx.a) if ctx.needs_input_grad[0] else None
        gy = _Scale.apply(g, ctx.b) if ctx.needs_input_grad[1] else None
        return gx, gy, None, None


# ---------------------------------------------------------------------------------------------- #
# normalisation (generator side: first-order only)
# ---------------------------------------------------------------------------------------------- #
class _InstNormStyle(Function):
    """y = InstanceNorm(x; eps, biased var) * (ys + 1) + yb  with style (N, 2C) = [ys | yb]
    (custom_layers.py:98-99 + stylegan/architectures.py:524-526); style=None -> plain IN."""

    @staticmethod
    def forward(ctx, x, style, eps, mean=None, rstd=None):
        x = _c(x)
        n, c, hw = _nchw(x)
        L = _lib.lib()
        if mean is None:     # otherwise: produced by the kernel that wrote x (bias_act(..., stats_eps=eps))
            mean, rstd = _new((n * c,), x), _new((n * c,), x)
            check(L.ganlab_instnorm_stats_f32(_p(x), _p(mean), _p(rstd), n * c, hw, eps, _st()), 'instnorm_stats')
        assert mean.numel() == n * c and rstd.numel() == n * c
        style_c = _c(style) if style is not None else None
        if style_c is not None:
            assert style_c.numel() == n * 2 * c
        y = torch.empty_like(x)
        check(L.ganlab_instnorm_style_fwd_f32(_p(x), _p(mean), _p(rstd), _p(style_c), _p(y), n, c, hw, _st()),
              'instnorm_style_fwd')
        ctx.save_for_backward(x, mean, rstd, style_c)
        ctx.style_shape = style.shape if style is not None else None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, mean, rstd, style = ctx.saved_tensors
        gy = _c(gy)
        n, c, hw = _nchw(x)
        L = _lib.lib()
        s1, s2 = _new((n, c), x), _new((n, c), x)
        check(L.ganlab_instnorm_style_bwd_reduce_f32(_p(gy), _p(x), _p(mean), _p(rstd), _p(s1), _p(s2), n * c, hw,
                                                     _st()), 'instnorm_bwd_reduce')
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            check(L.ganlab_instnorm_style_bwd_apply_f32(_p(gy), _p(x), _p(mean), _p(rstd), _p(style), _p(s1), _p(s2),
                                                        _p(gx), n, c, hw, _st()), 'instnorm_bwd_apply')
        gstyle = None
        if style is not None and ctx.needs_input_grad[1]:
            gstyle = torch.stack((s2, s1), dim=1).reshape(ctx.style_shape)  # d/dys = sum gy*xhat ; d/dyb = sum gy
        return gx, gstyle, None, None, None


class _LayerTail(Function):
    """out = InstanceNorm(act(blur?(x) + noise_w*noise + bias*bias_scale)) * (ys+1) + yb: everything of a generator
    layer after its convolution (stylegan/architectures.py:497-526) as two forward passes (the first one also accumulates
    the InstanceNorm statistics) and, backward, the reduce pass + ONE apply pass that also undoes the LeakyReLU and sums
    the bias / noise-weight gradients (+ blur^T when ``blur``).  First order only, like _InstNormStyle."""

    @staticmethod
    def forward(ctx, x, bias, noise, noise_w, style, bias_scale, act, slope, blur, eps):
        kern = k_blur_bias_act_stats if blur else k_bias_act_stats
        noise = _c(noise) if noise is not None else None
        y, mean, rstd = kern(x, bias, noise, noise_w, bias_scale, act, slope, eps)
        n, c, hw = _nchw(y)
        style_c = _c(style) if style is not None else None
        if style_c is not None:
            assert style_c.numel() == n * 2 * c
        out = torch.empty_like(y)
        check(_lib.lib().ganlab_instnorm_style_fwd_f32(_p(y), _p(mean), _p(rstd), _p(style_c), _p(out), n, c, hw, _st()),
              'instnorm_style_fwd')
        ctx.save_for_backward(y, mean, rstd, style_c, noise)
        ctx.bias_scale, ctx.act, ctx.slope, ctx.blur = bias_scale, act, slope, blur
        ctx.bias_shape = bias.shape if bias is not None else None
        ctx.nw_shape = noise_w.shape if noise_w is not None else None
        ctx.bias_ref, ctx.nw_ref = bias, noise_w
        ctx.style_shape = style.shape if style is not None else None
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        ctx.want_x_grad, ctx.want_bias_grad = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        ctx.want_nw_grad, ctx.want_style_grad = ctx.needs_input_grad[3], ctx.needs_input_grad[4]
        gz, gb, gnw, gstyle = _layer_tail_backward(ctx, ctx.saved_tensors, gout, blur=ctx.blur)
        return (gz if ctx.want_x_grad else None), gb, None, gnw, gstyle, None, None, None, None, None


# ---------------------------------------------------------------------------------------------- #
# deferred InstanceNorm ("modulated" consumers; csrc/mod.hip)
# ---------------------------------------------------------------------------------------------- #
class Deferred(object):
    """The output ``a * s[n,c] + t[n,c]`` of a generator layer (InstanceNorm + AdaIN of the activated tensor ``a``) that
    has NOT been written to memory: its consumer applies it (per-sample weights / border-class bias, csrc/mod.hip).
    Gradient contract: whoever consumes ``a`` sends back d loss / d (a*s + t) - the gradient with respect to the
    NORMALISED tensor - as the gradient of ``a``; ``_LayerTailDeferred.backward`` is the InstanceNorm backward of that
    (exactly what the materialised path feeds it), so ``s`` / ``t`` carry no gradient of their own."""
    __slots__ = ('a', 's', 't', 'mean', 'rstd', 'style', 'link')

    def __init__(self, a, s, t, mean, rstd, style, link=None):
        self.a, self.s, self.t, self.mean, self.rstd, self.style, self.link = a, s, t, mean, rstd, style, link

    def read(self):
        """Called by every consumer of ``a`` (see ``RgbGradLink``); returns self."""
        if self.link is not None:
            self.link.readers += 1
        return self

    @property
    def shape(self):
        return self.a.shape


class RgbGradLink(object):
    """Link between a generator layer's tail (the producer of a ``Deferred``) and toRGB where toRGB is its ONLY reader (the
    last layer outside the fade-in, stylegan/architectures.py:398-402): toRGB's input gradient is a 1x1 conv of the <= 4-plane
    image gradient, cheaper to recompute inside the tail's two InstanceNorm backward passes than to write as C planes and
    read back twice (csrc/pointwise.hip, RgbSrc).  toRGB's backward fills ``grgb`` / ``wp`` / ``crgb`` and returns a
    placeholder for the gradient of ``a``; the tail's backward - which always runs after it - reads them instead."""
    __slots__ = ('readers', 'grgb', 'wp', 'crgb')

    def __init__(self):
        self.readers, self.grgb, self.wp, self.crgb = 0, None, None, 0


def _rgb_link_ok(link, n, c, crgb, hw):
    import os
    return link is not None and link.readers == 1 and get_compute_dtype() == 'f32' and \
        os.environ.get('GANLAB_TORGB_FOLD') != '0' and \
        bool(_lib.lib().ganlab_instnorm_bwd_rgb_supported(n, c, crgb, hw))


def _affine_from_stats(mean, rstd, style, n, c):
    """s = rstd * (ys + 1), t = yb - mean * s  as (N, C) tensors (style: (N, 2C) = [ys | yb] or None)."""
    s_, t_ = _new((n, c), mean), _new((n, c), mean)
    check(_lib.lib().ganlab_in_affine_f32(_p(mean), _p(rstd), _p(style), _p(s_), _p(t_), n, c, _st()), 'in_affine')
    return s_, t_


def _layer_tail_backward(ctx, saved_tail, gout, blur=False):
    """Shared by _LayerTail / _LayerTailDeferred / _ConvModTail: InstanceNorm + style backward of ``gout`` (the gradient
    with respect to the normalised tensor), LeakyReLU undone, bias / noise-weight / style gradients from the same pass.
    ``blur``: the tail sits behind a blur - the (self-adjoint) blur of gz runs in the SAME pass and gz is never written.
    Returns (gz or blur(gz), gb, gnw, gstyle)."""
    y, mean, rstd, style, noise = saved_tail
    n, c, hw = _nchw(y)
    L = _lib.lib()
    rgb = getattr(ctx, 'link', None)
    if rgb is not None and rgb.grgb is None:
        rgb = None
    params = _want_param_grads()
    want_b = ctx.bias_shape is not None and ctx.want_bias_grad and params
    want_nw = ctx.nw_shape is not None and ctx.want_nw_grad and params
    s1, s2 = _new((n, c), y), _new((n, c), y)
    if rgb is not None:            # ``gout`` is toRGB's placeholder: the gradient is recomputed from the image gradient
        assert not blur
        grgb, wp_rgb, crgb = rgb.grgb, rgb.wp, rgb.crgb
        rgb.grgb = rgb.wp = None
        wsr = torch.empty((L.ganlab_instnorm_bwd_reduce_rgb_workspace(n, c, hw) + 3) // 4, dtype=torch.float32,
                          device=y.device)
        check(L.ganlab_instnorm_style_bwd_reduce_rgb_f32(_p(grgb), _p(wp_rgb), crgb, _p(y), _p(mean), _p(rstd), _p(s1),
                                                         _p(s2), n, c, hw, _p(wsr), wsr.numel() * 4, _st()),
              'instnorm_bwd_reduce_rgb')
    else:
        gout = _c(gout)
        check(L.ganlab_instnorm_style_bwd_reduce_f32(_p(gout), _p(y), _p(mean), _p(rstd), _p(s1), _p(s2), n * c, hw,
                                                     _st()), 'instnorm_bwd_reduce')
    gz = gb = gnw = None
    if ctx.want_x_grad or want_b or want_nw:
        gz = torch.empty_like(y)
        _sink('gb', getattr(ctx, 'bias_ref', None) if want_b else None)
        _sink('gnw', getattr(ctx, 'nw_ref', None) if want_nw else None)
        gb = _take('gb', (c,), y) if want_b else None
        gnw = _take('gnw', (c,), y) if want_nw else None
        ws = torch.empty((L.ganlab_instnorm_bwd_act_workspace(n, c, hw) + 3) // 4, dtype=torch.float32,
                         device=y.device) if (want_b or want_nw) else None
        if rgb is not None:
            check(L.ganlab_instnorm_style_bwd_act_rgb_f32(_p(grgb), _p(wp_rgb), crgb, _p(y), _p(mean), _p(rstd), _p(style),
                                                          _p(s1), _p(s2), _p(noise) if want_nw else None, _p(gz), _p(gb),
                                                          _p(gnw), n, c, hw, ctx.act, ctx.slope, ctx.bias_scale, _p(ws),
                                                          ws.numel() * 4 if ws is not None else 0, _st()),
                  'instnorm_bwd_act_rgb')
        elif blur and blur_fusable(y):
            check(L.ganlab_instnorm_style_bwd_act_blur_f32(_p(gout), _p(y), _p(mean), _p(rstd), _p(style), _p(s1), _p(s2),
                                                           _p(noise) if want_nw else None, _p(gz), _p(gb), _p(gnw), n, c,
                                                           int(y.shape[2]), int(y.shape[3]), ctx.act, ctx.slope,
                                                           ctx.bias_scale, _p(ws), ws.numel() * 4 if ws is not None else 0,
                                                           _st()), 'instnorm_bwd_act_blur')
        else:
            check(L.ganlab_instnorm_style_bwd_act_f32(_p(gout), _p(y), _p(mean), _p(rstd), _p(style), _p(s1), _p(s2),
                                                      _p(noise) if want_nw else None, _p(gz), _p(gb), _p(gnw), n, c,
                                                      hw, ctx.act, ctx.slope, ctx.bias_scale, _p(ws),
                                                      ws.numel() * 4 if ws is not None else 0, _st()),
                  'instnorm_bwd_act')
            if blur and ctx.want_x_grad:
                gz = k_blur(gz)
    gstyle = None
    if style is not None and ctx.want_style_grad:
        gstyle = torch.stack((s2, s1), dim=1).reshape(ctx.style_shape)
    if _sunk('gb'):
        gb = None
    if _sunk('gnw'):
        gnw = None
    return gz, (gb.view(ctx.bias_shape) if gb is not None else None), \
        (gnw.view(ctx.nw_shape) if gnw is not None else None), gstyle


class _LayerTailDeferred(Function):
    """_LayerTail without its second pass: (a, s, t) with a = act(blur?(x) + noise_w*noise + bias*bias_scale) and the
    per-(n, c) scale / shift of InstanceNorm + style - see ``Deferred`` for the gradient contract."""

    @staticmethod
    def forward(ctx, x, bias, noise, noise_w, style, bias_scale, act, slope, blur, eps, link=None):
        kern = k_blur_bias_act_stats if blur else k_bias_act_stats
        noise = _c(noise) if noise is not None else None
        ctx.link = link
        y, mean, rstd = kern(x, bias, noise, noise_w, bias_scale, act, slope, eps)
        n, c, _ = _nchw(y)
        style_c = _c(style) if style is not None else None
        s_, t_ = _affine_from_stats(mean, rstd, style_c, n, c)
        # (save_for_backward, not attributes: y is an OUTPUT of this node - holding it on ctx would be a reference
        # cycle that only Python's cyclic collector breaks, i.e. 2 GiB tensors freed late and a growing allocator)
        ctx.save_for_backward(y, mean, rstd, style_c, noise)
        ctx.bias_scale, ctx.act, ctx.slope, ctx.blur = bias_scale, act, slope, blur
        ctx.bias_shape = bias.shape if bias is not None else None
        ctx.nw_shape = noise_w.shape if noise_w is not None else None
        ctx.bias_ref, ctx.nw_ref = bias, noise_w
        ctx.style_shape = style.shape if style is not None else None
        ctx.want_x_grad, ctx.want_bias_grad = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        ctx.want_nw_grad, ctx.want_style_grad = ctx.needs_input_grad[3], ctx.needs_input_grad[4]
        ctx.mark_non_differentiable(s_, t_, mean, rstd)
        return y, s_, t_, mean, rstd

    @staticmethod
    @once_differentiable
    def backward(ctx, g_b, *_):
        gz, gb, gnw, gstyle = _layer_tail_backward(ctx, ctx.saved_tensors, g_b, blur=ctx.blur)
        gx = gz if ctx.want_x_grad else None
        return gx, gb, None, gnw, gstyle, None, None, None, None, None, None


class _Materialize(Function):
    """b = a*s + t written out (the second pass of round 1) for consumers without a modulated kernel; the gradient
    passes through unchanged (it IS d/db, which the deferred tail's backward expects)."""

    @staticmethod
    def forward(ctx, a, mean, rstd, style):
        n, c, hw = _nchw(a)
        out = torch.empty_like(a)
        check(_lib.lib().ganlab_instnorm_style_fwd_f32(_p(a), _p(mean), _p(rstd), _p(style), _p(out), n, c, hw, _st()),
              'instnorm_style_fwd')
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        return g, None, None, None


def materialize(x):
    """Tensor for a plain consumer: a ``Deferred`` is normalised + styled now, a tensor is returned as is."""
    if isinstance(x, Deferred):
        x.read()
        return _Materialize.apply(x.a, x.mean, x.rstd, x.style)
    return x


_AFF_OK = {}      # (kind, shape, weight shape) -> bool: the support queries are pure functions of the geometry


def mod_conv_shape_ok(shape, weight, padding=1):
    """Can the 3x3 layer ``weight`` consume a deferred tensor of ``shape`` through the thin rolling-window kernel that also
    finishes the layer (noise, bias, LeakyReLU, statistics in its epilogue)?"""
    if get_compute_dtype() != 'f32' or padding != 1:
        return False
    key = ('mod', tuple(int(v) for v in shape), tuple(weight.shape))
    hit = _AFF_OK.get(key)
    if hit is None:
        n, cin, h, w = key[1]
        hit = False
        if tuple(weight.shape[1:]) == (cin, 3, 3):
            g = ConvGeom(n, cin, h, w, int(weight.shape[0]), 3, 1, 0, 0)
            hit = bool(_lib.lib().ganlab_mod_conv_supported(ctypes.byref(g)))
        _AFF_OK[key] = hit
    return hit


def conv_tail_shape_ok(shape, weight, padding=1):
    """The same one-pass layer on the tile kernels (thicker layers): affine on load, noise + bias + LeakyReLU + statistics in
    the conv's epilogue - where the affine-on-load forward AND weight gradient take the geometry."""
    if get_compute_dtype() != 'f32' or padding != 1:
        return False
    key = ('tail', tuple(int(v) for v in shape), tuple(weight.shape))
    hit = _AFF_OK.get(key)
    if hit is None:
        n, cin, h, w = key[1]
        hit = False
        if tuple(weight.shape[1:]) == (cin, 3, 3) and conv_aff_ok(shape, weight, False, padding):
            g = ConvGeom(n, cin, h, w, int(weight.shape[0]), 3, 1, 0, 0)
            hit = _lib.lib().ganlab_conv_fwd_aff_tail_chunks(ctypes.byref(g)) > 0
        _AFF_OK[key] = hit
    return hit


def mod_conv_ok(d, weight, padding=1):
    return isinstance(d, Deferred) and (mod_conv_shape_ok(d.a.shape, weight, padding) or
                                        conv_tail_shape_ok(d.a.shape, weight, padding))


def conv_aff_ok(shape, weight, up=False, padding=1):
    """Can the 3x3 conv ``weight`` (behind a nearest 2x upsample when ``up``) read a deferred tensor of ``shape`` through
    the affine-on-load kernels - forward AND weight gradient?"""
    if get_compute_dtype() != 'f32' or padding != 1:
        return False
    key = ('up' if up else 'plain', tuple(int(v) for v in shape), tuple(weight.shape))
    hit = _AFF_OK.get(key)
    if hit is None:
        n, cin, h, w = key[1]
        hit = False
        if tuple(weight.shape[1:]) == (cin, 3, 3):
            g = ConvGeom(n, cin, h, w, int(weight.shape[0]), 3, 1, 1 if up else 0, 0)
            L = _lib.lib()
            hit = (L.ganlab_conv_s2_aff_supported(ctypes.byref(g)) if up else L.ganlab_conv_aff_supported(ctypes.byref(g))) == 3
        _AFF_OK[key] = hit
    return hit


def deferrable(x):
    """A layer output of this shape can stay un-normalised for an affine-on-load consumer: the fused statistics pass
    takes it (plane >= 32x32, rows of whole float4s)."""
    import os
    if os.environ.get('GANLAB_DEFER') == '0':          # A/B knob: keep the two-pass layer tail of round 1
        return False
    n, c, h, w = x.shape
    return get_compute_dtype() == 'f32' and h * w >= STATS_MIN_PLANE and w % 4 == 0 and h % 2 == 0


def k_conv_fwd_aff(a, s_, t_, w, g, scale):
    """conv(up2?(a*s + t), w) * scale with the affine applied while the patch is staged (zero padding stays zero)."""
    a, w = _c(a, 'conv input'), _c(w, 'conv weight')
    assert tuple(a.shape) == g.in_shape and tuple(s_.shape) == (g.N, g.Cin) and tuple(t_.shape) == (g.N, g.Cin)
    _note('fwd', g)
    y = _new(g.out_shape, a)
    L = _lib.lib()
    if g.up and x3_s2_ok(g):
        check(L.ganlab_conv_s2_fwd_aff_x3(_p(a), _packed_x3_s2(w, 1, scale).data_ptr(), _p(s_), _p(t_), None, _p(y), g.ref(), 1.0,
                                          ACT_NONE, 0.2, _st()), 'conv_s2_fwd_aff_x3')
    elif g.up:
        assert g.s2
        wp = _packed(w, PACK_FWD, scale, s2_up=1)
        check(L.ganlab_conv_s2_fwd_aff_f32(_p(a), _p(wp), _p(s_), _p(t_), None, _p(y), g.ref(), 1.0, ACT_NONE, 0.2, _st()),
              'conv_s2_fwd_aff')
    elif x3_ok(g):
        check(L.ganlab_conv_fwd_aff_x3(_p(a), _packed_x3(w, PACK_FWD, scale).data_ptr(), _p(s_), _p(t_), None, _p(y), g.ref(), 1.0,
                                       ACT_NONE, 0.2, _st()), 'conv_fwd_aff_x3')
    else:
        wp = _packed(w, PACK_FWD, scale)
        check(L.ganlab_conv_fwd_aff_f32(_p(a), _p(wp), _p(s_), _p(t_), None, _p(y), g.ref(), 1.0, ACT_NONE, 0.2, _st()),
              'conv_fwd_aff')
    return y


def k_conv_wgrad_aff(gy, a, s_, t_, g, scale):
    """Weight gradient of conv(up2?(a*s + t), w): the x operand is normalised on the fly, like the forward's."""
    gy, a = _c(gy, 'conv grad_out'), _c(a, 'conv input')
    assert tuple(gy.shape) == g.out_shape and tuple(a.shape) == g.in_shape
    _note('wgrad', g)
    L = _lib.lib()
    gw = _take('gw', (g.Cout, g.Cin, 3, 3), a)
    if not g.up and x3_wgrad_ok(g):
        return _wgrad_x3(gy, a, _c(s_), _c(t_), gw, g, scale)
    if g.up and x3_s2_wgrad_ok(g):
        return _wgrad_x3_s2(gy, a, _c(s_), _c(t_), gw, g, scale)
    if g.up:
        ws = torch.empty((max(L.ganlab_conv_s2_wgrad_workspace(g.ref()), 4) + 3) // 4, dtype=torch.float32, device=a.device)
        check(L.ganlab_conv_s2_wgrad_aff_f32(_p(gy), _p(a), _p(s_), _p(t_), _p(gw), g.ref(), scale, _p(ws), ws.numel() * 4,
                                             _st()), 'conv_s2_wgrad_aff')
    else:
        ws = torch.empty((max(L.ganlab_conv_wgrad_workspace(g.ref()), 4) + 3) // 4, dtype=torch.float32, device=a.device)
        check(L.ganlab_conv_wgrad_aff_f32(_p(gy), _p(a), _p(s_), _p(t_), _p(gw), g.ref(), scale, _p(ws), ws.numel() * 4,
                                          _st()), 'conv_wgrad_aff')
    return gw


class _ConvAff(Function):
    """y = scale * conv(up2?(b), w) for the deferred tensor b = a*s + t (``Deferred``): the consumer's half of the
    deferred InstanceNorm.  Backward: d/da := d loss / d b (the contract of ``Deferred``) from the plain input-gradient
    kernel on the shared weights; the weight gradient contracts gy with the same on-the-fly b.  First order only."""

    @staticmethod
    def forward(ctx, a, s_, t_, w, g, scale):
        ctx.save_for_backward(a, s_, t_, w)
        ctx.g, ctx.scale = g, scale
        return k_conv_fwd_aff(a, s_, t_, w, g, scale)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        a, s_, t_, w = ctx.saved_tensors
        ga = k_conv_dgrad(gy, w, ctx.g, ctx.scale) if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[3] and _want_param_grads():
            _sink('gw', w)
            gw = k_conv_wgrad_aff(gy, a, s_, t_, ctx.g, ctx.scale)
            if _sunk('gw'):
                gw = None
        return ga, None, None, gw, None, None


def conv_aff(d, weight, scale, up=False):
    """3x3 'same' conv (behind a nearest 2x upsample when ``up``) of a ``Deferred`` tensor; see ``conv_aff_ok``."""
    n, cin, h, w = d.read().a.shape
    g = Geom(n, cin, h, w, weight.shape[0], 3, 1, 1 if up else 0, 0)
    return _ConvAff.apply(d.a, d.s, d.t, weight, g, float(scale))


class _ConvModTail(Function):
    """A plain thin 3x3 generator layer consuming a deferred input and producing a deferred output in ONE pass over the
    activations:  a_out = act(conv(a_in * s + t (zero padded), w) * scale + noise_w*noise + bias)  with the InstanceNorm
    statistics of a_out from the epilogue (csrc/mod.hip).  Backward: InstanceNorm backward of the incoming d/db_out (round-1
    kernels) -> gz; d/db_in = plain input-gradient kernel on the shared weights; the weight gradient is the rolling-window
    kernel with the affine on its x operand."""

    @staticmethod
    def forward(ctx, a_in, s_in, t_in, w, bias, noise, noise_w, style, scale, bias_scale, act, slope, eps, link=None):
        a_in, w = _c(a_in), _c(w)
        n, cin, h, wd = a_in.shape
        cout = w.shape[0]
        L = _lib.lib()
        ctx.link = link
        g = Geom(n, cin, h, wd, cout, 3, 1, 0)
        _note('fwd', g)
        noise = _c(noise) if noise is not None else None
        y = _new((n, cout, h, wd), a_in)
        mean, rstd = _new((n, cout), a_in), _new((n, cout), a_in)
        thin = mod_conv_shape_ok(a_in.shape, w)
        if not thin and x3_ok(g):                  # thick layer on the split-product kernel, same one-pass form
            chunks = L.ganlab_conv_fwd_aff_tail_x3_chunks(g.ref())
            ws = torch.empty((n * cout * chunks * 2,), dtype=torch.float64, device=a_in.device)
            check(L.ganlab_conv_fwd_aff_tail_x3(_p(a_in), _packed_x3(w, PACK_FWD, scale).data_ptr(), _p(s_in), _p(t_in), _p(bias),
                                                _p(noise), _p(noise_w), _p(y), _p(mean), _p(rstd), g.ref(), bias_scale, act, slope,
                                                eps, _p(ws), ws.numel() * 8, _st()), 'conv_fwd_aff_tail_x3')
            wp = None
        else:
            wp = _packed(w, PACK_FWD, scale)
        if wp is None:
            pass
        elif thin:                                 # thin layer: the rolling-window kernel
            chunks = L.ganlab_mod_conv_stat_chunks(g.ref())
            ws = torch.empty((n * cout * chunks * 2,), dtype=torch.float64, device=a_in.device)
            check(L.ganlab_mod_conv_fwd_f32(_p(a_in), _p(wp), _p(s_in), _p(t_in), _p(bias), _p(noise), _p(noise_w), _p(y),
                                            _p(mean), _p(rstd), g.ref(), bias_scale, act, slope, eps, _p(ws), ws.numel() * 8,
                                            _st()), 'mod_conv_fwd')
        else:                                      # thicker layers: the tile kernels' TAIL epilogue
            chunks = L.ganlab_conv_fwd_aff_tail_chunks(g.ref())
            ws = torch.empty((n * cout * chunks * 2,), dtype=torch.float64, device=a_in.device)
            check(L.ganlab_conv_fwd_aff_tail_f32(_p(a_in), _p(wp), _p(s_in), _p(t_in), _p(bias), _p(noise), _p(noise_w), _p(y),
                                                 _p(mean), _p(rstd), g.ref(), bias_scale, act, slope, eps, _p(ws),
                                                 ws.numel() * 8, _st()), 'conv_fwd_aff_tail')
        style_c = _c(style) if style is not None else None
        s_, t_ = _affine_from_stats(mean, rstd, style_c, n, cout)
        ctx.save_for_backward(a_in, s_in, t_in, w, y, mean, rstd, style_c, noise)
        ctx.g, ctx.scale = g, scale
        ctx.bias_scale, ctx.act, ctx.slope, ctx.blur = bias_scale, act, slope, False
        ctx.bias_shape = bias.shape if bias is not None else None
        ctx.nw_shape = noise_w.shape if noise_w is not None else None
        ctx.bias_ref, ctx.nw_ref = bias, noise_w
        ctx.style_shape = style.shape if style is not None else None
        ctx.want_x_grad = ctx.needs_input_grad[0] or ctx.needs_input_grad[3]
        ctx.want_bias_grad, ctx.want_nw_grad = ctx.needs_input_grad[4], ctx.needs_input_grad[6]
        ctx.want_style_grad = ctx.needs_input_grad[7]
        ctx.mark_non_differentiable(s_, t_, mean, rstd)
        return y, s_, t_, mean, rstd

    @staticmethod
    @once_differentiable
    def backward(ctx, g_b, *_):
        a_in, s_in, t_in, w = ctx.saved_tensors[:4]
        gz, gb, gnw, gstyle = _layer_tail_backward(ctx, ctx.saved_tensors[4:], g_b)
        g = ctx.g
        ga = gw = None
        if ctx.needs_input_grad[0]:
            ga = k_conv_dgrad(gz, w, g, ctx.scale)              # d/d(a_in*s + t): shared weights, plain kernel
        if ctx.needs_input_grad[3] and _want_param_grads():
            _sink('gw', w)
            gw = k_conv_wgrad_aff(gz, a_in, s_in, t_in, g, ctx.scale)
            if _sunk('gw'):
                gw = None
        return ga, None, None, gw, gb, None, gnw, gstyle, None, None, None, None, None, None


class _UpConvBlurTail(Function):
    """A generator layer that opens a resolution,  Upsample -> conv3x3 -> blur -> +noise -> +bias -> LeakyReLU  with the
    InstanceNorm statistics of the result (stylegan/architectures.py:292-334, 497-526), in ONE pass over the activations
    (csrc/conv_s2_roll_blur.hip) - from a plain input or a deferred one (``s_in`` / ``t_in``: affine on load) to a deferred
    output (see ``Deferred``).  Backward: the layer tail's (InstanceNorm backward + LeakyReLU' + blur^T in one pass), then
    the up-conv's input gradient on the shared weights and its weight gradient with the same on-the-fly input."""

    @staticmethod
    def forward(ctx, a_in, s_in, t_in, w, bias, noise, noise_w, style, g, scale, bias_scale, act, slope, eps):
        noise = _c(noise) if noise is not None else None
        y, mean, rstd = k_conv_s2_fwd_blur_tail(a_in, s_in, t_in, w, bias, noise, noise_w, g, scale, bias_scale, act, slope,
                                                eps)
        style_c = _c(style) if style is not None else None
        s_, t_ = _affine_from_stats(mean, rstd, style_c, g.N, g.Cout)
        ctx.save_for_backward(a_in, s_in, t_in, w, y, mean, rstd, style_c, noise)
        ctx.g, ctx.scale = g, scale
        ctx.bias_scale, ctx.act, ctx.slope = bias_scale, act, slope
        ctx.bias_shape = bias.shape if bias is not None else None
        ctx.nw_shape = noise_w.shape if noise_w is not None else None
        ctx.style_shape = style.shape if style is not None else None
        ctx.bias_ref, ctx.nw_ref = bias, noise_w
        ctx.want_x_grad = ctx.needs_input_grad[0] or ctx.needs_input_grad[3]
        ctx.want_bias_grad, ctx.want_nw_grad = ctx.needs_input_grad[4], ctx.needs_input_grad[6]
        ctx.want_style_grad = ctx.needs_input_grad[7]
        ctx.mark_non_differentiable(s_, t_, mean, rstd)
        return y, s_, t_, mean, rstd

    @staticmethod
    @once_differentiable
    def backward(ctx, g_b, *_):
        a_in, s_in, t_in, w = ctx.saved_tensors[:4]
        gz, gb, gnw, gstyle = _layer_tail_backward(ctx, ctx.saved_tensors[4:], g_b, blur=True)   # d / d (conv output)
        g = ctx.g
        ga = gw = None
        if ctx.needs_input_grad[0]:
            ga = k_conv_dgrad(gz, w, g, ctx.scale)              # d/d(a_in*s + t) resp. d/d a_in: shared weights, plain kernel
        if ctx.needs_input_grad[3] and _want_param_grads():
            _sink('gw', w)
            gw = k_conv_wgrad_aff(gz, a_in, s_in, t_in, g, ctx.scale) if s_in is not None else \
                k_conv_wgrad(gz, a_in, g, ctx.scale)
            if _sunk('gw'):
                gw = None
        return ga, None, None, gw, gb, None, gnw, gstyle, None, None, None, None, None, None


def upconv_blur_tail_ok(shape, weight, deferred, padding=1):
    """Can the layer  Upsample -> conv3x3(weight) -> blur -> tail  on an input of ``shape`` run as the one-pass kernel (and,
    for a deferred input, its weight gradient with the affine on load)?"""
    if padding != 1 or tuple(weight.shape[2:]) != (3, 3) or int(weight.shape[1]) != int(shape[1]):
        return False
    n, cin, h, w = (int(v) for v in shape)
    try:
        g = Geom(n, cin, h, w, int(weight.shape[0]), 3, 1, 1, 0)
    except ValueError:
        return False
    if not conv_s2_blur_ok(g) or not _lib.lib().ganlab_blur_fused_supported(2 * h, 2 * w):
        return False
    return conv_aff_ok(shape, weight, True, padding) if deferred else True


def upconv_blur_tail(x, weight, scale, bias=None, noise=None, noise_w=None, style=None, bias_scale=1.0, act=None, slope=0.2,
                     eps=1e-8):
    """``x``: a tensor or a ``Deferred``; returns the layer's output as a ``Deferred`` (see ``upconv_blur_tail_ok``)."""
    a_in, s_in, t_in = (x.a, x.s, x.t) if isinstance(x, Deferred) else (x, None, None)
    if isinstance(x, Deferred):
        x.read()
    n, cin, h, w = a_in.shape
    g = Geom(n, cin, h, w, weight.shape[0], 3, 1, 1, 0)
    a = ACT_LRELU if act == 'lrelu' else ACT_NONE
    y, s_, t_, mean, rstd = _UpConvBlurTail.apply(a_in, s_in, t_in, weight, bias, noise, noise_w, style, g, float(scale),
                                                  float(bias_scale), a, float(slope), float(eps))
    return Deferred(y, s_, t_, mean, rstd, style)


class _ToRGBMod(Function):
    """toRGB (1x1, stylegan/architectures.py torgb) of a deferred tensor: per-sample weights w*s and bias b + w.t, built
    and folded back by two small kernels (csrc/mod.hip)."""

    @staticmethod
    def forward(ctx, a, s_, t_, w, bias, scale, bias_scale, link=None):
        a, w = _c(a), _c(w)
        n, cin, h, wd = a.shape
        cout = w.shape[0]
        L = _lib.lib()
        ctx.link = link
        weff, beff = _new((n, cin, 4), a), _new((n, 4), a)
        check(L.ganlab_mod_torgb_prep_f32(_p(w), _p(bias), _p(s_), _p(t_), _p(weff), _p(beff), n, cin, cout, scale,
                                          bias_scale, _st()), 'mod_torgb_prep')
        y = _new((n, cout, h, wd), a)
        check(L.ganlab_mod_torgb_fwd_f32(_p(a), _p(weff), _p(beff), _p(y), n, cin, cout, h * wd, _st()), 'mod_torgb_fwd')
        ctx.save_for_backward(a, s_, t_, w)
        ctx.scale, ctx.bias_scale = scale, bias_scale
        ctx.bias_shape = bias.shape if bias is not None else None
        ctx.bias_ref = bias
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        a, s_, t_, w = ctx.saved_tensors
        gy = _c(gy)
        n, cin, h, wd = a.shape
        cout = w.shape[0]
        L = _lib.lib()
        g = Geom(n, cin, h, wd, cout, 1, 0, 0)
        ga = None
        if ctx.link is not None:                         # a link left armed by a backward whose tail never ran is disarmed here
            ctx.link.grgb = ctx.link.wp = None
        if ctx.needs_input_grad[0]:                      # d/d(a*s + t)
            link = ctx.link
            if _rgb_link_ok(link, n, cin, cout, h * wd):
                # the producer's InstanceNorm backward recomputes it from gy (RgbGradLink); the engine only needs the shape
                link.grgb, link.wp, link.crgb = gy, _packed(w, PACK_DGRAD, ctx.scale), cout
                ga = torch.zeros((1,), dtype=a.dtype, device=a.device).expand(a.shape)      # (a 4-byte memset: never garbage)
            else:
                ga = k_conv_dgrad(gy, w, g, ctx.scale)
        gw = gb = None
        want_w = ctx.needs_input_grad[3]
        want_b = ctx.bias_shape is not None and ctx.needs_input_grad[4]
        if (want_w or want_b) and _want_param_grads():
            out = _new((n, 68), a)
            ws = torch.empty((L.ganlab_mod_torgb_cross_workspace(n) + 3) // 4, dtype=torch.float32, device=a.device)
            check(L.ganlab_mod_torgb_cross_f32(_p(a), _p(gy), _p(out), n, cin, cout, h * wd, _p(ws), ws.numel() * 4,
                                               _st()), 'mod_torgb_cross')
            _sink('gw', w if want_w else None)
            _sink('gb', ctx.bias_ref if want_b else None)
            gw = _take('gw', (cout, cin, 1, 1), a) if want_w else None
            gb = _take('gb', tuple(ctx.bias_shape), a) if want_b else None
            check(L.ganlab_mod_torgb_wgrad_f32(_p(out), _p(s_), _p(t_), _p(gw), _p(gb), n, cin, cout, ctx.scale,
                                               ctx.bias_scale, _st()), 'mod_torgb_wgrad')
            if _sunk('gw'):
                gw = None
            if _sunk('gb'):
                gb = None
        return ga, None, None, gw, gb, None, None, None


class _PixelNorm(Function):
    @staticmethod
    def forward(ctx, x, eps):
        x = _c(x)
        n, c, hw = _nchw(x)
        y = torch.empty_like(x)
        check(_lib.lib().ganlab_pixelnorm_fwd_f32(_p(x), _p(y), n, c, hw, eps, _st()), 'pixelnorm_fwd')
        ctx.save_for_backward(x)
        ctx.eps = eps
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, = ctx.saved_tensors
        gy = _c(gy)
        n, c, hw = _nchw(x)
        gx = torch.empty_like(x)
        check(_lib.lib().ganlab_pixelnorm_bwd_f32(_p(gy), _p(x), _p(gx), n, c, hw, ctx.eps, _st()), 'pixelnorm_bwd')
        return gx, None


# ---------------------------------------------------------------------------------------------- #
# BatchNorm / LayerNorm building blocks (ResNet GAN path): every piece is closed under differentiation,
# so torch.autograd composes first and second derivatives (WGAN-GP through LayerNorm) from HIP kernels.
# ---------------------------------------------------------------------------------------------- #
class _ChanAffine(Function):
    """y[n,c,...] = x[n,c,...] * scale[c] + shift[c]  (scale / shift may be None)."""

    @staticmethod
    def forward(ctx, x, scale, shift):
        x = _c(x)
        n, c, hw = _nchw(x)
        sc = _c(scale) if scale is not None else None
        sh = _c(shift) if shift is not None else None
        y = torch.empty_like(x)
        check(_lib.lib().ganlab_chan_affine_f32(_p(x), _p(sc), _p(sh), _p(y), n, c, hw, _st()), 'chan_affine')
        ctx.save_for_backward(x, sc)
        ctx.has = (scale is not None, shift is not None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, sc = ctx.saved_tensors
        gx = _ChanAffine.apply(g, sc, None) if ctx.needs_input_grad[0] else None
        gs = _ChanSum.apply(_Mul.apply(g, x), None, 1.0) if (ctx.has[0] and ctx.needs_input_grad[1]) else None
        gt = _ChanSum.apply(g, None, 1.0) if (ctx.has[1] and ctx.needs_input_grad[2]) else None
        return gx, gs, gt


class _Mul(Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        assert a.shape == b.shape
        out = torch.empty_like(a)
        check(_lib.lib().ganlab_mul_f32(_p(a), _p(b), _p(out), a.numel(), _st()), 'mul')
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        return (_Mul.apply(g, b) if ctx.needs_input_grad[0] else None,
                _Mul.apply(g, a) if ctx.needs_input_grad[1] else None)


class _Tanh(Function):
    """nn.Tanh: the ResNet generators' output layer, and the hidden activation of ``--nonlinearity tanh``
    (config.py:208,254; resnetgan/learner.py:180-181).  The backward is a Function of its own so that a gradient penalty
    differentiates through it (a critic with tanh activations under WGAN-GP / R1)."""
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        y = torch.empty_like(x)
        check(_lib.lib().ganlab_tanh_fwd_f32(_p(x), _p(y), x.numel(), _st()), 'tanh_fwd')
        ctx.save_for_backward(y)
        ctx.set_materialize_grads(False)
        return y

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None
        y, = ctx.saved_tensors
        return _TanhBwd.apply(g, y)


class _TanhBwd(Function):
    """gx = g * (1 - y^2), y = tanh(x): linear in g, and d gx / d y = -2 g y (the second order of tanh)."""
    @staticmethod
    def forward(ctx, g, y):
        g, y = _c(g), _c(y)
        gx = torch.empty_like(g)
        check(_lib.lib().ganlab_tanh_bwd_f32(_p(g), _p(y), _p(gx), g.numel(), _st()), 'tanh_bwd')
        ctx.save_for_backward(g, y)
        return gx

    @staticmethod
    def backward(ctx, go):
        g, y = ctx.saved_tensors
        gg = _TanhBwd.apply(go, y) if ctx.needs_input_grad[0] else None
        gy = scale(mul(mul(go, g), y), -2.0) if ctx.needs_input_grad[1] else None
        return gg, gy


def chan_affine(x, scale=None, shift=None):
    return _ChanAffine.apply(x, scale, shift)


def mul(a, b):
    return _Mul.apply(a, b)


def tanh(x):
    return _Tanh.apply(x)


def channel_sum(x):
    return _ChanSum.apply(x, None, 1.0)


def _row_workspace(rows, length, device):
    nbytes = _lib.lib().ganlab_ln_rowsums_workspace(rows, length)
    return torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=device)


class _BatchNormTrain(Function):
    """nn.BatchNorm2d in training mode on fused kernels (csrc/norm.hip): batch statistics (2 launches), the C-element
    bookkeeping - rstd, scale, running estimates, batch counter - in one (``bn_finalize``), normalise + affine (1);
    backward sums = the parameter gradients (2) and the input gradient (1).  First order only - the generator is never
    inside a gradient penalty.  Returns (y, batch mean, biased batch variance)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, running_mean, running_var, momentum, batches, act_slope):
        x = _c(x)
        n, c, hw = _nchw(x)
        m = n * hw
        L = _lib.lib()
        ws = _row_workspace(c, m, x.device)
        mom = _new((c, 3), x)
        check(L.ganlab_bn_stats_f32(_p(x), _p(mom), n, c, hw, ctypes.c_void_p(ws.data_ptr()), ws.numel() * 8, _st()),
              'bn_stats')
        fin = _new((4, c), x)                                # mean, var, rstd, scale = rstd * weight
        for buf in (running_mean, running_var, batches):
            assert buf is None or (buf.is_contiguous() and buf.device == x.device)
        assert batches is None or batches.dtype == torch.int64
        check(L.ganlab_bn_finalize_f32(_p(mom), _p(_c(weight)) if weight is not None else None,
                                       _p(running_mean) if running_mean is not None else None,
                                       _p(running_var) if running_var is not None else None,
                                       ctypes.c_void_p(batches.data_ptr()) if batches is not None else None, _p(fin), c,
                                       float(eps), float(momentum), m / max(m - 1, 1), _st()), 'bn_finalize')
        mean, var, rstd, scale = fin[0], fin[1], fin[2], fin[3]
        y = torch.empty_like(x)
        # act_slope: the LeakyReLU / ReLU behind the normalisation (resnetgan/resblocks.py:48-49) applied by the same pass;
        # its backward is applied by bn_bwd_sums as it loads the gradient (sign(y) = sign of the pre-activation)
        check(L.ganlab_bn_apply_f32(_p(x), _p(mean), _p(scale), _p(_c(bias)) if bias is not None else None, _p(y), n, c,
                                    hw, ACT_LRELU if act_slope is not None else ACT_NONE,
                                    float(act_slope) if act_slope is not None else 1.0, _st()), 'bn_apply')
        ctx.save_for_backward(x, mean, rstd, scale, y if act_slope is not None else None)
        ctx.act_slope = act_slope
        ctx.has_weight, ctx.has_bias = weight is not None, bias is not None
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, _gm, _gv):
        x, mean, rstd, scale, yact = ctx.saved_tensors
        gy = _c(gy)
        n, c, hw = _nchw(x)
        L = _lib.lib()
        ws = _row_workspace(c, n * hw, x.device)
        sums = _new((c, 3), x)
        gz = torch.empty_like(gy) if yact is not None else None      # gy * lrelu'(y), written by the sums pass
        check(L.ganlab_bn_bwd_sums_f32(_p(gy), _p(x), _p(mean), _p(rstd), _p(sums), n, c, hw,
                                       ctypes.c_void_p(ws.data_ptr()), ws.numel() * 8, _p(yact), _p(gz),
                                       float(ctx.act_slope) if yact is not None else 1.0, _st()), 'bn_bwd_sums')
        if gz is not None:
            gy = gz
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            check(L.ganlab_bn_bwd_apply_f32(_p(gy), _p(x), _p(mean), _p(rstd), _p(sums), _p(scale), _p(gx), n, c, hw,
                                            _st()), 'bn_bwd_apply')
        want_p = _want_param_grads()
        gw = sums[:, 1] if (ctx.has_weight and want_p and ctx.needs_input_grad[1]) else None
        gb = sums[:, 0] if (ctx.has_bias and want_p and ctx.needs_input_grad[2]) else None
        return gx, gw, gb, None, None, None, None, None, None


def batch_norm(x, weight, bias, running_mean, running_var, training, momentum=0.1, eps=1e-5, batches=None,
               act_slope=None):
    """nn.BatchNorm2d semantics (biased variance for the normalisation, unbiased for the running estimate): fused
    kernels in training mode (``batches``: the module's ``num_batches_tracked``, counted by the same launch that moves the
    running estimates), one per-channel affine with the running statistics in eval mode."""
    if training:
        return _BatchNormTrain.apply(x, weight, bias, float(eps), running_mean, running_var, float(momentum), batches,
                                     act_slope)[0]
    xc = chan_affine(x, None, -running_mean)          # centred, like the training path
    rstd = torch.rsqrt(running_var + eps)
    y = chan_affine(xc, rstd * weight if weight is not None else rstd, bias)
    return y if act_slope is None else bias_act(y, act='lrelu', slope=act_slope)


def batch_norm_composed(x, weight, bias, eps=1e-5):
    """Training-mode BatchNorm composed from channel sums / per-channel affines / products (any order of
    differentiation through autograd): the cross-check of the fused kernels in the tests."""
    n, c, hw = _nchw(x)
    m = n * hw
    mean = channel_sum(x) / m
    xc = chan_affine(x, None, -mean)
    var = channel_sum(mul(xc, xc)) / m
    rstd = torch.rsqrt(var + eps)
    return chan_affine(xc, rstd * weight if weight is not None else rstd, bias)


def _ln_rowsums(a, wa, x, mean, rstd, n, m, b2=None, w2=None, yact=None, gz=None, slope=1.0):
    """[N][3] row sums (see ganlab_ln_rowsums_f32): sum a*wa, sum a*wa*xhat, sum a*b2*w2; with ``yact`` a is first multiplied
    by lrelu'(yact) and that product is also written to ``gz``."""
    L = _lib.lib()
    nbytes = L.ganlab_ln_rowsums_workspace(n, m)
    ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=x.device)
    out = _new((n, 3), x)
    check(L.ganlab_ln_rowsums_f32(_p(a), _p(wa), _p(x), _p(mean), _p(rstd), _p(b2), _p(w2), _p(out), n, m,
                                  ctypes.c_void_p(ws.data_ptr()), ws.numel() * 8, _p(yact), _p(gz), float(slope), _st()),
          'ln_rowsums')
    return out


def _ln_project(a, wa, x, mean, rstd, sums, wo, n, m):
    """wo * P_x(a * wa) with P_x(g) = rstd * (g - mean(g) - xhat * mean(g * xhat)) per row."""
    out = torch.empty_like(x)
    check(_lib.lib().ganlab_ln_project_f32(_p(a), _p(wa), _p(x), _p(mean), _p(rstd), _p(sums), _p(wo), _p(out), n, m,
                                           _st()), 'ln_project')
    return out


class _LayerNorm(Function):
    """nn.LayerNorm(x.shape[1:]) with elementwise affine on fused kernels (csrc/norm.hip): 2 launches forward,
    4 backward, 7 for the backward of the backward (WGAN-GP through the critic)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, act_slope):
        x = _c(x)
        n = x.shape[0]
        m = x.numel() // n
        w = _c(weight).reshape(m) if weight is not None else None
        b = _c(bias).reshape(m) if bias is not None else None
        L = _lib.lib()
        mean, rstd = _new((n,), x), _new((n,), x)
        nbytes = L.ganlab_row_stats_workspace(n, m)
        ws = torch.empty((max(nbytes, 8) + 7) // 8, dtype=torch.float64, device=x.device)
        check(L.ganlab_row_stats_f32(_p(x), _p(mean), _p(rstd), n, m, eps, _p(ws), ws.numel() * 8, _st()), 'ln_stats')
        y = torch.empty_like(x)
        # act_slope: the LeakyReLU / ReLU behind the normalisation (resnetgan/resblocks.py:48-49) in the same pass; the
        # backward applies its derivative while loading the gradient for the row sums (sign(y) = sign of the pre-activation)
        check(L.ganlab_ln_affine_fwd_f32(_p(x), _p(mean), _p(rstd), _p(w), _p(b), _p(y), n, m,
                                         ACT_LRELU if act_slope is not None else ACT_NONE,
                                         float(act_slope) if act_slope is not None else 1.0, _st()), 'ln_affine_fwd')
        # the weight INPUT is saved: the double backward reaches the parameter
        ctx.save_for_backward(x, weight, mean, rstd, y if act_slope is not None else None)
        ctx.act_slope = act_slope
        ctx.wshape = weight.shape if weight is not None else None
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, mean, rstd, yact = ctx.saved_tensors
        w = weight.reshape(-1) if weight is not None else None      # tracked under create_graph
        want_p = _want_param_grads() and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        gx, gw, gb = _LayerNormBwd.apply(gy, x, w, mean, rstd, want_p, yact, ctx.act_slope)
        if gw is not None and ctx.wshape is not None:
            gw = gw.reshape(ctx.wshape)
            gb = gb.reshape(ctx.wshape) if ctx.has_bias else None
        else:
            gw = gb = None
        return gx, gw, gb if ctx.has_bias else None, None, None


class _LayerNormBwd(Function):
    """(gy, x, w) -> (gx, gw, gb); its own backward is the analytic double backward for a cotangent of gx (the
    gradient-penalty pass never keeps gw / gb: ops.input_grad_only).  With ``yact`` (the output of a LayerNorm fused with
    its LeakyReLU) gy is the gradient BEHIND the activation: gz = gy * lrelu'(yact) is formed by the row-sums pass and takes
    gy's place everywhere below; lrelu'' = 0, so the double backward only gains the same mask on its d/d gy."""

    @staticmethod
    def forward(ctx, gy, x, w, mean, rstd, want_param_grads, yact=None, act_slope=None):
        gy = _c(gy)
        w = _c(w) if w is not None else None
        n = x.shape[0]
        m = x.numel() // n
        L = _lib.lib()
        if yact is not None:
            gz = torch.empty_like(gy)
            sums = _ln_rowsums(gy, w, x, mean, rstd, n, m, yact=yact, gz=gz, slope=act_slope)
            gy = gz
        else:
            sums = _ln_rowsums(gy, w, x, mean, rstd, n, m)             # a = mean(ghat), beta = mean(ghat * xhat)
        gw = gb = None
        if want_param_grads and w is not None:       # projection and parameter gradients in one pass over (gy, x)
            gx, gw, gb = torch.empty_like(x), _new((m,), x), _new((m,), x)
            check(L.ganlab_ln_bwd_cols_f32(_p(gy), _p(w), _p(x), _p(mean), _p(rstd), _p(sums), _p(gx), _p(gw), _p(gb), n, m,
                                           _st()), 'ln_bwd_cols')
        else:
            gx = _ln_project(gy, w, x, mean, rstd, sums, None, n, m)   # P_x(gy * w)
        ctx.save_for_backward(gy, x, w, mean, rstd, gx, sums, yact)
        ctx.act_slope = act_slope
        ctx.set_materialize_grads(False)
        return gx, gw, gb

    @staticmethod
    @once_differentiable
    def backward(ctx, u, ggw, ggb):
        gy, x, w, mean, rstd, gx, sums, yact = ctx.saved_tensors
        if ggw is not None or ggb is not None:
            raise NotImplementedError('LayerNorm double backward through the parameter gradients is not needed by the '
                                      'GAN losses (the penalty differentiates the INPUT gradient only)')
        if u is None:
            return None, None, None, None, None, None, None, None
        u = _c(u)
        n = x.shape[0]
        m = x.numel() // n
        L = _lib.lib()
        usums = _ln_rowsums(u, None, x, mean, rstd, n, m, b2=gy, w2=w)   # sum u, sum u*xhat, sum u*ghat
        pu = _ln_project(u, None, x, mean, rstd, usums, None, n, m)      # P_x(u)
        g_w = None
        if w is not None:
            g_gy = torch.empty_like(gy)
            check(L.ganlab_colscale_f32(_p(pu), _p(w), _p(g_gy), n, m, _st()), 'ln_colscale')
            if _want_param_grads() and ctx.needs_input_grad[2]:
                g_w = _new((m,), x)
                check(L.ganlab_coldot_f32(_p(gy), _p(pu), None, None, _p(g_w), None, n, m, _st()), 'ln_coldot')
        else:
            g_gy = pu
        g_x = torch.empty_like(x)       # the per-row coefficients come from (sums, usums) inside the kernel
        check(L.ganlab_ln_bwdbwd_apply_f32(_p(x), _p(mean), _p(rstd), _p(pu), _p(gx), _p(sums), _p(usums), _p(g_x),
                                           n, m, _st()), 'ln_bwdbwd_apply')
        if yact is not None:
            g_gy = k_act_bwd(g_gy, yact, ctx.act_slope)
        return g_gy, g_x, g_w, None, None, None, None, None


def layer_norm(x, weight, bias, eps=1e-5, act_slope=None):
    """nn.LayerNorm(normalized_shape = x.shape[1:]) semantics: per-sample statistics over all features (biased
    variance), then the elementwise affine - fused kernels with an analytic double backward (csrc/norm.hip).
    ``act_slope``: LeakyReLU(act_slope) (0 = ReLU) of the result in the same passes, forward and backward."""
    return _LayerNorm.apply(x, weight, bias, float(eps), None if act_slope is None else float(act_slope))


def layer_norm_composed(x, weight, bias, eps=1e-5):
    """The same operator composed from channel sums / per-channel affines / products (every piece closed under
    differentiation, so autograd derives any order): the cross-check of the fused kernels in the tests."""
    shape = x.shape
    n = shape[0]
    f = x.numel() // n
    xr = x.reshape(1, n, f)
    mean = channel_sum(xr) / f
    xc = chan_affine(xr, None, -mean)
    var = channel_sum(mul(xc, xc)) / f
    y = chan_affine(xc, torch.rsqrt(var + eps), None)
    if weight is not None:
        y = chan_affine(y.reshape(n, f, 1), weight.reshape(f), bias.reshape(f) if bias is not None else None)
    return y.reshape(shape)


# ---------------------------------------------------------------------------------------------- #
# minibatch stddev statistic with explicit second order (discriminator, R1 path)
# ---------------------------------------------------------------------------------------------- #
class _MbstdStat(Function):
    @staticmethod
    def forward(ctx, x, gs, eps):
        x = _c(x)
        b = x.shape[0]
        G, F = b // gs, x.numel() // b
        stat = _new((G,), x)
        check(_lib.lib().ganlab_mbstd_fwd_f32(_p(x), _p(stat), G, gs, F, eps, _st()), 'mbstd_fwd')
        ctx.save_for_backward(x)
        ctx.gs, ctx.eps = gs, eps
        return stat

    @staticmethod
    def backward(ctx, gstat):
        x, = ctx.saved_tensors
        return _MbstdBwd.apply(x, gstat, ctx.gs, ctx.eps), None, None


class _MbstdBwd(Function):
    @staticmethod
    def forward(ctx, x, gstat, gs, eps):
        x, gstat = _c(x), _c(gstat)
        b = x.shape[0]
        G, F = b // gs, x.numel() // b
        gx = torch.empty_like(x)
        check(_lib.lib().ganlab_mbstd_bwd_f32(_p(x), _p(gstat), _p(gx), G, gs, F, eps, _st()), 'mbstd_bwd')
        ctx.save_for_backward(x, gstat)
        ctx.gs, ctx.eps = gs, eps
        return gx

    @staticmethod
    @once_differentiable
    def backward(ctx, ggx):
        x, gstat = ctx.saved_tensors
        ggx = _c(ggx)
        b = x.shape[0]
        G, F = b // ctx.gs, x.numel() // b
        g_gstat, g_x = _new((G,), x), torch.empty_like(x)
        check(_lib.lib().ganlab_mbstd_bwdbwd_f32(_p(x), _p(gstat), _p(ggx), _p(g_gstat), _p(g_x), G, ctx.gs, F,
                                                 ctx.eps, _st()), 'mbstd_bwdbwd')
        return g_x, g_gstat, None, None


# ---------------------------------------------------------------------------------------------- #
# scalar reductions / losses (first-order)
# ---------------------------------------------------------------------------------------------- #
class _Sum(Function):
    """scale * sum(x) or scale * sum(x^2) -> 0-dim tensor; the cotangent never leaves the device."""

    @staticmethod
    def forward(ctx, x, scale, squared):
        ctx.save_for_backward(x if squared else None)
        ctx.scale, ctx.squared, ctx.shape = scale, squared, x.shape
        return k_sum(x, scale, squared)

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        x, = ctx.saved_tensors
        if ctx.squared:
            return k_scale_dev(x, gout.reshape(1), 2.0 * ctx.scale), None, None
        return k_scale_dev(None, gout.reshape(1), ctx.scale, ctx.shape), None, None


class _BceMean(Function):
    @staticmethod
    def forward(ctx, x, target):
        x = _c(x)
        out = _new((), x)
        check(_lib.lib().ganlab_bce_logits_fwd_f32(_p(x), _p(out), x.numel(), target, _st()), 'bce_fwd')
        ctx.save_for_backward(x)
        ctx.target = target
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        x, = ctx.saved_tensors
        gx = torch.empty_like(x)
        g1 = _c(gout).reshape(1)
        check(_lib.lib().ganlab_bce_logits_bwd_f32(_p(x), _p(g1), _p(gx), x.numel(), ctx.target, _st()), 'bce_bwd')
        return gx, None


class _ChNormPenalty(Function):
    """scale * sum_{n,hw} (||g[n,:,hw]||_2 - gamma)^2  (WGAN-GP, resnetgan/learner.py:817-823)."""

    @staticmethod
    def forward(ctx, g, gamma, scale):
        g = _c(g)
        n, c, hw = _nchw(g)
        L = _lib.lib()
        ws = torch.empty((L.ganlab_sum_workspace(n * hw) + 3) // 4, dtype=torch.float32, device=g.device)
        out = _new((), g)
        check(L.ganlab_chnorm_penalty_fwd_f32(_p(g), _p(out), n, c, hw, gamma, scale, _p(ws), ws.numel() * 4, _st()),
              'chnorm_penalty_fwd')
        ctx.save_for_backward(g)
        ctx.gamma, ctx.scale = gamma, scale
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        g, = ctx.saved_tensors
        n, c, hw = _nchw(g)
        gg = torch.empty_like(g)
        g1 = _c(gout).reshape(1)
        check(_lib.lib().ganlab_chnorm_penalty_bwd_f32(_p(g), _p(g1), _p(gg), n, c, hw, ctx.gamma, ctx.scale, _st()),
              'chnorm_penalty_bwd')
        return gg, None, None


# ---------------------------------------------------------------------------------------------- #
# functional API
# ---------------------------------------------------------------------------------------------- #
ACT_DEFERRED = '_ganlab_act_deferred'    # attribute on a conv2d(defer_act_grad=True) result that really deferred


def conv2d(x, weight, bias=None, scale=1.0, padding=0, up=False, bias_scale=1.0, act=None, slope=0.2, pool=False,
           blur=False, defer_act_grad=False, in_act_slope=None, in_blur_handoff=None, in_rgb_handoff=None):
    """blur?(act(avgpool2?(scale*conv2d(up2?(x), weight, padding)) + bias*bias_scale)) on the matrix cores.
    ``pool``: the D down layer conv -> AvgPool2d(2) -> +bias -> LeakyReLU (progan/architectures.py:261-284)
    as one stride-2 kernel when the shape qualifies, else composed from the plain kernels.
    ``blur``: the binomial blur that follows conv+bias+LeakyReLU in a D block; its backward is fused with
    the LeakyReLU / bias backward.
    ``defer_act_grad`` / ``in_act_slope``: a pair of layers A -> B where B is a conv and the ONLY reader of A's
    LeakyReLU output (D block k's pooled conv -> block k+1's first conv).  A(defer_act_grad=True) may leave the
    multiplication by lrelu'(y_A) to B; when it does its result carries the attribute ``ACT_DEFERRED`` and the caller
    MUST run B with in_act_slope=<A's slope> - B's dgrad epilogue (or an explicit pass) then applies it."""
    n, cin, h, w = x.shape
    cout, cin_w, ks, _ = weight.shape
    if in_act_slope is not None:
        in_act_slope = float(in_act_slope)
        fold = not up and not pool and ks == 3 and (bias is not None or act == 'lrelu') and \
            conv_dgrad_mask_ok(Geom(n, cin, h, w, cout, ks, padding, 0, 0))
        if not fold:
            x, in_act_slope = _DeferredActGrad.apply(x, in_act_slope), None
    if blur:
        y = None
        if not pool and not up and ks in (1, 3) and act == 'lrelu':
            g = Geom(n, cin, h, w, cout, ks, padding, 0, 0)
            if _lib.lib().ganlab_blur_fused_supported(g.Ho, g.Wo):
                h = BlurHandoff()
                y = _ConvBiasAct.apply(x, weight, bias, g, float(scale), float(bias_scale), ACT_LRELU, float(slope),
                                       True, False, in_act_slope, h, None, None, in_rgb_handoff)
                if h.bits is not None:      # its only reader may take over this layer's blur^T / LeakyReLU' (fused_sequential)
                    setattr(y, BLUR_HANDOFF, h)
        return y if y is not None else _Blur.apply(conv2d(x, weight, bias, scale, padding, up, bias_scale, act, slope,
                                                          pool, in_act_slope=in_act_slope))
    if pool and not (not up and pool_fusable(n, cin, h, w, cout, ks, padding)):
        y = avg_pool2(conv2d(x, weight, None, scale, padding, up))
        return bias_act(y, bias, bias_scale=bias_scale, act=act, slope=slope)
    if ks == 4 and padding == 0 and h == 4 and w == 4 and not up:
        # 4x4 valid conv on a 4x4 map == linear over (ci, ky, kx)  (progan/architectures.py:227-229)
        y = linear(x.reshape(n, cin * 16), weight.reshape(cout, cin * 16), bias, scale, bias_scale, act, slope)
        return y.view(n, cout, 1, 1)
    g = Geom(n, cin, h, w, cout, ks, padding, up, pool)
    a = ACT_LRELU if act == 'lrelu' else ACT_NONE
    if bias is None and a == ACT_NONE:
        assert in_act_slope is None
        return _ConvFwd.apply(x, weight, g, float(scale))
    defer = bool(defer_act_grad) and a == ACT_LRELU and not conv_act_bwd_fusable(g)
    hin = None
    if in_blur_handoff is not None and pool and in_act_slope is None and in_blur_handoff.bits is not None and \
            conv_s2_blur_ok(g):
        hin = in_blur_handoff
    rgb = RgbHandoff() if (ks == 1 and a == ACT_LRELU and not up and not pool and cin <= 3) else None
    y = _ConvBiasAct.apply(x, weight, bias, g, float(scale), float(bias_scale), a, float(slope), False, defer,
                           in_act_slope, None, hin, rgb, in_rgb_handoff if (ks == 3 and not up and not pool) else None)
    if rgb is not None and rgb.bits is not None:
        setattr(y, RGB_HANDOFF, rgb)
    if defer:
        setattr(y, ACT_DEFERRED, True)
    return y


def linear(x, weight, bias=None, scale=1.0, bias_scale=1.0, act=None, slope=0.2):
    """act(scale * x @ weight.T + bias*bias_scale): a 1x1 conv over B 'pixels' of a 1x1 image."""
    n, cin = x.shape
    cout = weight.shape[0]
    y = conv2d(x.view(n, cin, 1, 1), weight.view(cout, cin, 1, 1), bias, scale, 0, False, bias_scale, act, slope)
    return y.view(n, cout)


STATS_MIN_PLANE = 1024   # planes below 32x32: the separate one-block-per-plane statistics kernel costs nothing


def bias_act(x, bias=None, noise=None, noise_w=None, bias_scale=1.0, act=None, slope=0.2, blur=False, stats_eps=None):
    """act(blur?(x) + noise_w*noise + bias*bias_scale); ``blur``: the binomial blur in front (one fused pass).
    ``stats_eps``: also return the InstanceNorm statistics of the result, ``(y, (mean, rstd))`` - or ``(y, None)``
    where the plane is too small / ragged for the fused accumulation - to hand to ``instnorm_style(stats=...)``."""
    a = ACT_LRELU if act == 'lrelu' else ACT_NONE
    eps = None
    if stats_eps is not None and x.dim() == 4:
        hw = x.shape[2] * x.shape[3]
        if hw >= STATS_MIN_PLANE and hw % 4 == 0:
            eps = float(stats_eps)
    if blur:
        if blur_fusable(x):
            out = _BlurBiasAct.apply(x, bias, noise, noise_w, float(bias_scale), a, float(slope), eps)
            return _with_stats(out, stats_eps, eps)
        x = _Blur.apply(x)
    return _with_stats(_BiasAct.apply(x, bias, noise, noise_w, float(bias_scale), a, float(slope), eps), stats_eps, eps)


def _with_stats(out, asked, fused):
    if asked is None:
        return out
    return (out[0], (out[1], out[2])) if fused is not None else (out, None)


def blur(x):
    return _Blur.apply(x)


def avg_pool2(x):
    return _Pool2.apply(x, 0.25)


def upsample2(x):
    return _Up2.apply(x, 1.0)


def resample(x, mode, align_corners=False):
    """``F.interpolate`` at scale 2 (``'bilinear_up'``) / 0.5 (``'bilinear_down'``, ``'nearest_down'``) of an NCHW map:
    nn.Upsample(mode='bilinear') / BilinearPool2d / NearestPool2d of the reference (custom_layers.py:59-75)."""
    if x.dim() == 3:
        x = x.view(-1, *x.shape)
    return _Resample.apply(x, mode, bool(align_corners), int(x.shape[2]), int(x.shape[3]), False)


def lerp(a, b, alpha):
    """a*(1-alpha) + b*alpha."""
    return _Axpby.apply(a, b, float(1.0 - alpha), float(alpha))


def scale(x, a):
    return _Scale.apply(x, float(a))


def add(a, b):
    """Residual sum a + b."""
    return _Axpby.apply(a, b, 1.0, 1.0)


def global_avg_pool(x):
    """nn.AvgPool2d(kernel_size=H) on an (N,C,H,H) map -> (N,C,1,1): plane sums through the channel-sum kernel."""
    n, c, h, w = x.shape
    return (_ChanSum.apply(x.reshape(1, n * c, h * w), None, 1.0 / (h * w))).view(n, c, 1, 1)


def layer_tail(x, bias=None, noise=None, noise_w=None, style=None, bias_scale=1.0, act=None, slope=0.2, blur=False,
               eps=1e-8):
    """InstanceNorm+style of act(blur?(x) + noise_w*noise + bias*bias_scale): the generator layer after its conv.
    Large planes take the fused forward / backward passes (_LayerTail); small or ragged ones compose the two ops."""
    a = ACT_LRELU if act == 'lrelu' else ACT_NONE
    if x.dim() == 4 and noise is not None and noise.dim() == 4 and noise.shape[0] != x.shape[0]:
        noise = noise.expand(x.shape[0], *noise.shape[1:])
    hw = x.shape[2] * x.shape[3] if x.dim() == 4 else 0
    if hw >= STATS_MIN_PLANE and hw % 4 == 0 and (not blur or blur_fusable(x)):
        return _LayerTail.apply(x, bias, noise, noise_w, style, float(bias_scale), a, float(slope), bool(blur),
                                float(eps))
    y = bias_act(x, bias, noise, noise_w, bias_scale, act, slope, blur)
    return instnorm_style(y, style, eps)


def instnorm_style(x, style=None, eps=1e-8, stats=None):
    """``stats``: (mean, rstd) of ``x`` for this ``eps`` from ``bias_act(..., stats_eps=eps)`` (saves the pass)."""
    if stats is None:
        return _InstNormStyle.apply(x, style, float(eps))
    return _InstNormStyle.apply(x, style, float(eps), stats[0], stats[1])


def pixelnorm(x, eps=1e-8):
    return _PixelNorm.apply(x, float(eps))


def style_mod(x, style):
    """AdaIN's affine WITHOUT the normalisation (``use_instancenorm=False``, stylegan/architectures.py:497-526 with the
    norm list empty): ``x * (ys + 1) + yb`` with ``style`` = (N, 2C) = [ys | yb].  A per-(sample, channel) affine is the
    per-channel affine kernel on the (1, N*C, H, W) view; its backward is made of differentiable ops, so R1-style
    double backward is closed."""
    n, c = int(x.shape[0]), int(x.shape[1])
    st = style.reshape(n, 2, c)
    ys1 = (st[:, 0] + 1.0).reshape(-1).contiguous()
    yb = st[:, 1].reshape(-1).contiguous()
    return chan_affine(x.reshape(1, n * c, *x.shape[2:]), ys1, yb).reshape(x.shape)


def group_broadcast(stat, group_size, h, w):
    return _GroupBroadcast.apply(stat, int(group_size), int(h), int(w))


def mbstd_stat(x, group_size, eps=1e-8):
    return _MbstdStat.apply(x, int(group_size), float(eps))


def sum_all(x, scale=1.0):
    return _Sum.apply(x, float(scale), False)


def sumsq_all(x, scale=1.0):
    return _Sum.apply(x, float(scale), True)


def bce_logits_mean(x, target):
    return _BceMean.apply(x, float(target))


def chnorm_penalty(g, gamma, scale):
    return _ChNormPenalty.apply(g, float(gamma), float(scale))


def layer_tail_deferred(x, bias=None, noise=None, noise_w=None, style=None, bias_scale=1.0, act=None, slope=0.2,
                        blur=False, eps=1e-8):
    """ops.layer_tail without the normalisation pass: returns a ``Deferred`` (see there) for a modulated consumer."""
    a = ACT_LRELU if act == 'lrelu' else ACT_NONE
    link = RgbGradLink() if not blur else None
    y, s_, t_, mean, rstd = _LayerTailDeferred.apply(x, bias, noise, noise_w, style, float(bias_scale), a, float(slope),
                                                     bool(blur), float(eps), link)
    return Deferred(y, s_, t_, mean, rstd, style, link)


def conv_mod_tail(d, weight, scale, bias=None, noise=None, noise_w=None, style=None, bias_scale=1.0, act=None, slope=0.2,
                  eps=1e-8):
    """3x3 layer + its tail on a deferred input, deferred output (``_ConvModTail``)."""
    a = ACT_LRELU if act == 'lrelu' else ACT_NONE
    d.read()
    link = RgbGradLink()
    y, s_, t_, mean, rstd = _ConvModTail.apply(d.a, d.s, d.t, weight, bias, noise, noise_w, style, float(scale),
                                               float(bias_scale), a, float(slope), float(eps), link)
    return Deferred(y, s_, t_, mean, rstd, style, link)


def torgb_mod(d, weight, bias, scale, bias_scale=1.0):
    d.read()
    return _ToRGBMod.apply(d.a, d.s, d.t, weight, bias, float(scale), float(bias_scale), d.link)


def torgb_mod_ok(d, weight):
    if not isinstance(d, Deferred):
        return False
    n, c, h, w = d.a.shape
    return tuple(weight.shape[1:]) == (c, 1, 1) and weight.shape[0] <= 4 and c <= 16 and (h * w) % 4 == 0
