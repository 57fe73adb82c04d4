#!/usr/bin/env python3
"""Benchmark of the G+D training step (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W                      # headline: BASELINE config #3
    N > 1, either way:
      * under a launcher (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
        --master-port P bench.py --gpus N ...): RANK / LOCAL_RANK / WORLD_SIZE come from the environment and
        WORLD_SIZE must equal --gpus (anything else exits non-zero: no silent single-rank run);
      * bare (python bench.py --gpus N ...): this process touches no GPU, starts exactly that launcher as a child with
        N fresh ranks, relays rank 0's JSON line and exits with the child's code.
    python bench.py --gpus 2 --dry-run-dist   # CPU rehearsal of the N-rank path: gloo ranks, host logic only
    python bench.py --config {2,4,5} ...                               # the other BASELINE configurations

Default (= --config 3): images/sec of the full G+D step (1 D-iteration + 1 G-iteration), StyleGAN 1024^2, batch 32
per GPU, fp32, nonsaturating loss + R1 (lambda 10) + drift, stabilised phase at the final resolution, synthetic
FFHQ-shaped data resident in HBM.
  --config 2: StyleGAN 128^2, batch 8, bf16-compute convolutions (fp32 storage / masters), same losses.
  --config 4: ProGAN 256^2: the FULL 4 -> 256 fade-in schedule runs first through ``learner.train()`` (shortened
              ``nimg_transition``, stated; its wall time is reported as ``schedule_seconds``), then K main iterations
              of the stabilised 256^2 phase are timed (WGAN + WGAN-GP + drift, batch 32).
  --config 5: ResNet GAN 64^2, batch 64, WGAN + WGAN-GP; a step = one main iteration = 1 G + 5 critic iterations.

Prints ONE JSON line on rank 0.
``roofline``: the dominant kernel family is the MFMA implicit-GEMM 3x3 conv; the configuration's north-star instance
is timed with device events on the launch stream INSIDE the timed steps and priced with its algorithmic FLOPs
(2*9*Cin*Cout*H*W*B); ``kernel`` / ``grid`` are read back from the library (the symbol that was dispatched).
``executed_tflops_step`` / ``frac_of_mfma_peak_step``: convolution FLOPs the step actually EXECUTES (counted in the
launchers) per second, over the MFMA peak of the compute dtype (with the split-product kernels on the fraction is reported as
``fp32_equivalent_over_fp32_mfma_peak_step`` instead: a speed label, the thick layers' products run on the bf16 cores); ``algorithmic_tflops_step`` prices the same step at
the reference's pass count and is a label, not a roofline fraction.
``cpu_baseline``: the CPU oracle's (oracle/, kind "port") G+D step of the same network on the host cores at a reduced
batch, 1 warm-up + 3 timed steps, median - a reported baseline, not a target.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# 4 G passes + 14 D passes per image and G+D step, algorithmic (reference pass count)
ALGO_TFLOP_PER_IMAGE = {('stylegan', 1024): 1.536, ('stylegan', 128): 0.752, ('progan', 256): 1.013}
PEAK_F32_MFMA_TFLOPS = 157.3                     # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2
PEAK_BF16_MFMA_TFLOPS = 2500.0                   # MI355X_MICROARCH.md: dense bf16 MFMA (~2.5 PF, no sparsity)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=None)
    p.add_argument('--warmup', type=int, default=None,
                   help='untimed steps (default 1; config 2: 5, so that the step graphs are captured before the timed region)')
    p.add_argument('--config', type=int, choices=(2, 3, 4, 5), default=3, help='BASELINE.json configuration (3 = headline)')
    p.add_argument('--res', type=int, default=None, help='override the resolution (StyleGAN configs)')
    p.add_argument('--batch', type=int, default=None, help='override the per-GPU batch')
    p.add_argument('--dtype', choices=('f32', 'bf16'), default=None,
                   help="compute dtype of the 3x3 convolutions (config 2 defaults to bf16)")
    p.add_argument('--nimg-transition', type=int, default=4096, help='config 4: images per phase of the schedule')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--cpu-baseline-res', type=int, default=None)
    p.add_argument('--cpu-baseline-batch', type=int, default=1)
    p.add_argument('--cpu-baseline-steps', type=int, default=3)
    p.add_argument('--no-roofline', action='store_true')
    p.add_argument('--step-graph', choices=('auto', '0', '1'), default='auto',
                   help='replay the stabilised iteration as HIP graphs (gan_lab_amd/graphs.py GraphedStep): auto = the '
                        "learner's own rule (single process, resolutions up to 256 - the launch-bound configurations)")
    p.add_argument('--dry-run-dist', action='store_true',
                   help='rehearse the multi-rank path on CPU: gloo ranks build a small learner (host logic only), '
                        'broadcast its arenas and run the bucketed gradient exchange; no GPU is touched')
    a = p.parse_args()
    dflt = {2: ('stylegan', 128, 8, 'bf16', 20), 3: ('stylegan', 1024, 32, 'f32', 3), 4: ('progan', 256, 32, 'f32', 10),
            5: ('resnetgan', 64, 64, 'f32', 5)}[a.config]
    a.model = dflt[0]
    a.res = a.res or dflt[1]
    a.batch = a.batch or dflt[2]
    a.dtype = a.dtype or dflt[3]
    a.steps = a.steps or dflt[4]
    if a.warmup is None:
        a.warmup = 5 if a.config == 2 else 1
    return a


# ------------------------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------------------------
def _quiet(fn, *a, **k):
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def build_learner(res, batch, device, dtype='f32', model='stylegan', seed=1234, **extra):
    """Stabilised-phase learner at the final resolution (StyleGAN: nonsaturating + R1; ProGAN: WGAN + WGAN-GP)."""
    from gan_lab_amd.config import make_config
    bs_dict = {r: batch for r in (4, 8, 16, 32, 64, 128, 256, 512, 1024)}
    common = dict(dev='cuda', pin_memory=False, res_samples=res, res_dataset=res, batch_size=batch, bs_dict=bs_dict,
                  num_iters_save_model=10 ** 9, log_every=0, compute_dtype=dtype, random_seed=seed)
    common.update(extra)
    if model == 'stylegan':
        from gan_lab_amd.stylegan.learner import StyleGANLearner
        cfg = make_config('stylegan', loss='nonsaturating', gradient_penalty='r1', lda=10., init_res=res,
                          cutoff_trunc_trick=4 if res >= 64 else None, **common)
        learner = _quiet(StyleGANLearner, cfg)
    else:
        from gan_lab_amd.progan.learner import ProGANLearner
        common.setdefault('init_res', res)
        learner = _quiet(ProGANLearner, make_config('progan', **common))
    learner.beta = learner.get_smoothing_ewma_beta(half_life=10.)
    learner.gen_model.train()
    learner.disc_model.train()
    return learner


def one_step(learner, real):
    if learner.use_step_graph and learner.step_graph.eligible():
        return learner.step_graph(real)       # eager for its first calls, then HIP-graph replays (graphs.GraphedStep)
    learner.set_requires_grad_disc(True)
    ld = learner.d_step(real, defer_update=True)
    learner.set_requires_grad_disc(False)
    lg = learner.g_step(d_update_pending=True)
    return ld, lg


class Workload(object):
    """What a 'step' is for one BASELINE configuration, and which conv instance its roofline prices."""

    def __init__(self, a, torch):
        self.a, self.torch = a, torch
        self.extra = {}
        m, res, b = a.model, a.res, a.batch
        if m == 'stylegan':
            self.learner = build_learner(res, b, 'cuda', a.dtype, 'stylegan')
            self.real = torch.rand(b, 3, res, res, device='cuda') * 2 - 1
            c = 16 if res >= 1024 else max(16, min(512, 8192 // (res // 2)))
            self.northstar = (b, c, res)              # the top-resolution 3x3 conv (the "modulated conv" of the north star)
            self.images_per_step = b
            self.what = (f'StyleGAN res_samples={res} FFHQ-shaped synthetic, bs={b}/GPU '
                         f'{"fp32" if a.dtype == "f32" else "bf16 compute / fp32 storage+master"}, nonsaturating + '
                         f'R1(lambda=10) + drift, stabilised phase, 1 D-iter + 1 G-iter')
        elif m == 'progan':
            from gan_lab_amd.utils.data_utils import SyntheticImageLoader
            nimg = a.nimg_transition
            self.learner = build_learner(res, b, 'cuda', a.dtype, 'progan', init_res=4, nimg_transition=nimg)
            dl = SyntheticImageLoader(1 << 22, b, 4, device='cuda')
            n_phases = 1 + 2 * (len(bin(res)) - len(bin(4)))          # 4 stab, then fade + stab per doubling
            iters = n_phases * ((nimg + b * a.world - 1) // (b * a.world)) + 8
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _quiet(self.learner.train, dl, num_main_iters=iters)
            torch.cuda.synchronize()
            L = self.learner
            assert L.gen_model.curr_res == res and not L.gen_model.fade_in_phase, (L.gen_model.curr_res, L.gen_model.alpha)
            self.extra = {'schedule_seconds': round(time.perf_counter() - t0, 2), 'schedule_iterations': iters,
                          'schedule': f'4 -> {res}, nimg_transition={nimg}, bs={b}/GPU at every resolution'}
            self.real = torch.rand(b, 3, res, res, device='cuda') * 2 - 1
            self.northstar = (b, max(16, min(512, 8192 // (res // 2))), res)
            self.images_per_step = b
            self.what = (f'ProGAN res_samples={res} CelebA-HQ-shaped synthetic, bs={b}/GPU fp32, WGAN + WGAN-GP + drift; '
                         f'full 4->{res} fade-in schedule through learner.train() first (schedule_seconds), then the '
                         f'stabilised {res}^2 phase timed, 1 D-iter + 1 G-iter')
        else:
            from gan_lab_amd.config import make_config
            from gan_lab_amd.resnetgan.learner import GANLearner
            from gan_lab_amd.utils.data_utils import SyntheticImageLoader
            cfg = make_config('resnetgan', dev='cuda', pin_memory=False, batch_size=b, res_samples=res, res_dataset=res,
                              num_iters_save_model=10 ** 9, log_every=0, random_seed=1234)
            self.learner = _quiet(GANLearner, cfg)
            self.dl = SyntheticImageLoader(1 << 22, b, res, device='cuda')
            self.nd = cfg.num_disc_iters
            self.northstar = (b, 64, res)             # the residual blocks' 64 -> 64 3x3 conv at the top resolution
            self.images_per_step = b * self.nd        # real images consumed per main iteration
            self.what = (f'ResNet GAN {res}x{res} LSUN-shaped synthetic, bs={b}/GPU fp32, WGAN + WGAN-GP, one main '
                         f'iteration = 1 G-iter + {self.nd} critic iters (learner.train())')

    def step(self):
        if self.a.model == 'resnetgan':
            _quiet(self.learner.train, self.dl, num_main_iters=1)
            ll = self.learner.last_losses if self.learner.log_every else {}
            return ll.get('loss_d'), ll.get('loss_g')
        return one_step(self.learner, self.real)


# ------------------------------------------------------------------------------------------------------------------
# roofline of the dominant kernel
# ------------------------------------------------------------------------------------------------------------------
# the per-kernel HBM traffic table under profiles/ comes from the default command (config 3) resp. --config 2: only the
# instance that kernel symbol ran THERE is priced with it
NORTHSTAR_OF_TRAFFIC_TABLE = {False: (16, 1024, 32)}


def kernel_traffic(symbol):
    """HBM bytes per launch of ``symbol`` from the newest tracked per-kernel traffic table (tools/step_traffic.sh:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over the bench command, gfx950 corrections applied) -
    (bytes, source file), or None.  Measured by the profiler, not inside this run: the source is named."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_step_traffic.json')))
    if not files or not symbol:
        return None
    short = re.sub(r'\(.*$', '', re.sub(r'^void ', '', symbol).replace('(anonymous namespace)::', ''))[:70]
    try:
        rec = json.load(open(files[-1]))['kernels'].get(short)
    except (OSError, ValueError, KeyError):
        return None
    if not rec:
        return None
    return round(rec['read_bytes_per_launch'] + rec['written_bytes_per_launch']), os.path.relpath(files[-1], ROOT)


def measure_dominant_kernel(torch, batch, c, res, reps=5):
    """Average launch duration of one 3x3 conv instance in an isolated loop (device events on the launch stream) and
    the symbol / grid the library dispatched for it."""
    from gan_lab_amd import _lib, ops
    x = torch.randn(batch, c, res, res, device='cuda')
    w = torch.randn(c, c, 3, 3, device='cuda')
    g = ops.Geom(batch, c, res, res, c, 3, 1, 0)
    bf = g.bf is not None
    for _ in range(2):
        ops.k_conv_fwd(x, w, None, g, 0.05)
    kernel, grid = _lib.last_launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.k_conv_fwd(x, w, None, g, 0.05)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = 2.0 * 9 * c * c * res * res * batch
    # The thick fp32 layers run as split products on the bf16 matrix cores (csrc/conv_x3.hip): SIX bf16 MFMA products per
    # fp32 product.  Such a launch is priced in the bf16 FLOPs it executes against the bf16 MFMA peak; the fp32 product
    # rate (a sixth of it) is reported beside it.
    x3 = 'conv_x3' in (kernel or '')
    if x3:
        flops *= 6.0
    ach = flops / (ms * 1e-3) / 1e12
    peak = PEAK_BF16_MFMA_TFLOPS if (bf or x3) else PEAK_F32_MFMA_TFLOPS
    out = {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
           'traffic': None, 'kernel': kernel, 'grid': grid, 'instance': f'3x3 {c}->{c} @{res}^2 x{batch}',
           'ms_per_launch': round(ms, 4), 'flops_per_launch': flops}
    if x3:
        out['arithmetic'] = '3xbf16 split products, fp32 accumulate'
        out['bf16_products_per_fp32_product'] = 6
        out['fp32_equivalent_tflops'] = round(ach / 6.0, 2)
        out['fp32_mfma_peak'] = PEAK_F32_MFMA_TFLOPS
    t = kernel_traffic(kernel)
    if t is not None and (c, res, batch) == NORTHSTAR_OF_TRAFFIC_TABLE.get(bf):
        out['traffic'], out['traffic_source'] = t
    return out


class InSituKernelTimer(object):
    """HIP events around every launch of ONE conv geometry (forward and input gradient: the same kernel) during the
    timed steps - the dominant kernel's launch duration in the state the training step actually runs it in (clocks,
    caches, neighbours), rather than in an isolated loop."""

    MAX_EVENTS = 256      # bracketed launches: the first ones of the timed region (an event pair costs the host ~8 us,
                          # which a launch-bound configuration would otherwise pay 200 times per step)

    def __init__(self, torch, ops, batch, c, res):
        self.torch, self.ops, self.key = torch, ops, (batch, c, res, res, c, 3, 1, 0, 0)
        self.events = []
        self._orig = None

    def _match(self, g):
        return (g.N, g.Cin, g.Hin, g.Win, g.Cout, g.ks, g.pad, g.up, g.pool) == self.key and not g.s2

    def __enter__(self):
        ops, torch = self.ops, self.torch
        f0, d0 = self._orig = (ops.k_conv_fwd, ops.k_conv_dgrad)     # whatever is installed now: timers nest

        def prepack(w, mode, scale, g):
            if g.bf is not None:
                ops._packed_bf16(w, mode, scale)
            elif ops.x3_ok(g, mode == ops.PACK_DGRAD):
                ops._packed_x3(w, mode, scale)
            else:
                ops._packed(w, mode, scale)

        def fwd(x, w, bias, g, *a, **k):
            if len(self.events) >= self.MAX_EVENTS or not self._match(g) or torch.cuda.is_current_stream_capturing():
                return f0(x, w, bias, g, *a, **k)      # (an event recorded into a graph capture cannot be read back)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            prepack(w, ops.PACK_FWD, a[0] if a else k['scale'], g)    # keep the (cached) weight packing out of the bracket
            e0.record()
            y = f0(x, w, bias, g, *a, **k)
            e1.record()
            self.events.append((e0, e1))
            return y

        def dgrad(gy, w, g, scale):
            if len(self.events) >= self.MAX_EVENTS or not self._match(g) or torch.cuda.is_current_stream_capturing():
                return d0(gy, w, g, scale)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            prepack(w, ops.PACK_DGRAD, scale, g)
            e0.record()
            gx = d0(gy, w, g, scale)
            e1.record()
            self.events.append((e0, e1))
            return gx
        ops.k_conv_fwd, ops.k_conv_dgrad = fwd, dgrad
        return self

    def __exit__(self, *exc):
        self.ops.k_conv_fwd, self.ops.k_conv_dgrad = self._orig
        return False

    def mean_ms(self):
        if not self.events:
            return None, 0
        self.torch.cuda.synchronize()
        ts = [a.elapsed_time(b) for a, b in self.events]
        return sum(ts) / len(ts), len(ts)


class FlopCounter(object):
    """Convolution FLOPs the timed steps execute, by pass kind (gan_lab_amd.ops launch observer)."""

    def __init__(self, ops):
        self.ops, self.by_kind, self.launches = ops, {'fwd': 0.0, 'dgrad': 0.0, 'wgrad': 0.0}, 0

    def __call__(self, kind, g):
        self.by_kind[kind] += self.ops.conv_flops(g)
        self.launches += 1

    def __enter__(self):
        self._prev = self.ops.set_launch_observer(self)
        return self

    def __exit__(self, *exc):
        self.ops.set_launch_observer(self._prev)
        return False

    @property
    def total(self):
        return sum(self.by_kind.values())


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle, on the host cores)
# ------------------------------------------------------------------------------------------------------------------
def host_cpu():
    """(model name, physical cores, logical cpus) from /proc/cpuinfo."""
    model, cores, logical = None, set(), 0
    try:
        phys = core = None
        for line in open('/proc/cpuinfo'):
            k, _, v = line.partition(':')
            k, v = k.strip(), v.strip()
            if k == 'processor':
                logical += 1
            elif k == 'model name' and model is None:
                model = v
            elif k == 'physical id':
                phys = v
            elif k == 'core id':
                core = v
            elif not k and phys is not None:
                cores.add((phys, core))
                phys = core = None
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    usable = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    n_phys = len(cores) or max(1, logical // 2)
    # the container's CPU quota (cgroup v2 cpu.max): a one-GPU box of the pool shows all 256 logical CPUs of the host but
    # is granted 16 CPUs' worth of time - 128 threads on that only take turns.  `cores` of the record is what the
    # baseline can actually use.
    quota = None
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            quota = max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        pass
    n = min(n_phys, usable)
    return model or 'unknown', (min(n, quota) if quota else n), usable


def cpu_baseline(torch, res, batch, steps=3):
    """The oracle's G+D step (same math, same loss config) on the host cores: 1 warm-up + ``steps`` timed, median."""
    from gan_lab_amd import progressive as P
    from gan_lab_amd.progan.architectures import StyleDiscriminator
    from gan_lab_amd.stylegan.architectures import StyleGenerator
    from oracle import nets, step
    import numpy as np
    model, n_phys, usable = host_cpu()
    torch.set_num_threads(n_phys)
    torch.manual_seed(0)
    P.StyleGAN.reset_state()
    g = StyleGenerator(final_res=res, blur_type='binomial',
                       truncation_trick_params={'beta': .995, 'psi': .7, 'cutoff_stage': 4 if res >= 64 else None})
    d = StyleDiscriminator(final_res=res, blur_type='binomial')
    for _ in range(int(np.log2(res)) - 2):
        g.increase_scale()
        d.increase_scale()
    gan = step.FunctionalGAN(g.state_dict(), d.state_dict(), nets.make_cfg(), model='stylegan',
                             loss='nonsaturating', gp='r1', lda=10., eps_drift=.001)
    del g, d
    L = nets.stylegen_num_layers(gan.g)
    noise = lambda: [torch.randn(batch, 1, 4 * 2 ** (n // 2), 4 * 2 ** (n // 2)) for n in range(L)]  # noqa: E731
    real = torch.rand(batch, 3, res, res) * 2 - 1
    times = []
    for i in range(1 + steps):
        t0 = time.perf_counter()
        gan.d_step(torch.randn(batch, 512), real, noise())
        gan.g_step(torch.randn(batch, 512), noise(), beta=0.999)
        times.append(time.perf_counter() - t0)
    med = statistics.median(times[1:])
    return {'value': round(batch / med, 5), 'unit': 'images/sec', 'cores': n_phys, 'kind': 'port',
            'cpu_model': model, 'logical_cpus': usable, 'torch_threads': torch.get_num_threads(),
            'step_seconds': [round(t, 2) for t in times],
            'sample': f'G+D step of StyleGAN {res}^2 at batch {batch} (oracle/step.py FunctionalGAN, torch CPU fp32, '
                      f'{n_phys} threads = the physical cores of {model} this container may use [cgroup cpu.max]): 1 warm-up + '
                      f'{steps} timed steps, median '
                      f'{med:.1f} s'}


def _baseline_record(torch, batch, images_per_step, times, what, model, n_phys, usable):
    med = statistics.median(times[1:])
    return {'value': round(images_per_step / med, 5), 'unit': 'images/sec', 'cores': n_phys, 'kind': 'port',
            'cpu_model': model, 'logical_cpus': usable, 'torch_threads': torch.get_num_threads(),
            'step_seconds': [round(t, 2) for t in times],
            'sample': f'{what} at batch {batch} (torch CPU fp32, {n_phys} threads = the usable physical cores of {model}): '
                      f'1 warm-up + {len(times) - 1} timed steps, median {med:.1f} s'}


def cpu_baseline_progan(torch, res, batch, steps=2):
    """Config #4's CPU yardstick: the oracle's G+D step of the full-width ProGAN at the final resolution (WGAN +
    WGAN-GP + drift, PixelNorm; oracle/nets.py + oracle/step.py FunctionalGAN) at a reduced batch, stated."""
    from gan_lab_amd import progressive as P
    from gan_lab_amd.progan.architectures import ProDiscriminator, ProGenerator
    from oracle import nets, step
    import numpy as np
    model, n_phys, usable = host_cpu()
    torch.set_num_threads(n_phys)
    torch.manual_seed(0)
    P.ProGAN.reset_state()
    g, d = ProGenerator(final_res=res, blur_type='binomial'), ProDiscriminator(final_res=res, blur_type='binomial')
    for _ in range(int(np.log2(res)) - 2):
        g.increase_scale()
        d.increase_scale()
    gan = step.FunctionalGAN(g.state_dict(), d.state_dict(), nets.make_cfg(use_pixelnorm=True), model='progan',
                             loss='wgan', gp='wgan-gp', lda=10., eps_drift=.001)
    del g, d
    real = torch.rand(batch, 3, res, res) * 2 - 1
    times = []
    for _ in range(1 + steps):
        t0 = time.perf_counter()
        gan.d_step(torch.randn(batch, 512), real, eps_interp=torch.rand(batch, 1, 1, 1))
        gan.g_step(torch.randn(batch, 512), beta=0.999)
        times.append(time.perf_counter() - t0)
    return _baseline_record(torch, batch, batch, times, f'G+D step of ProGAN {res}^2, stabilised phase (oracle/step.py '
                            f'FunctionalGAN: WGAN + WGAN-GP + drift)', model, n_phys, usable)


def cpu_baseline_resnet(torch, res, batch, steps=2, nd=5):
    """Config #5's CPU yardstick: one main iteration (1 G-iter + ``nd`` critic iters, WGAN + WGAN-GP) of the ResNet GAN
    through oracle/resnet.py ResnetFunctionalGAN at a reduced batch, stated; images = real images consumed."""
    from gan_lab_amd.resnetgan import architectures as A
    from oracle import resnet
    model, n_phys, usable = host_cpu()
    torch.set_num_threads(n_phys)
    torch.manual_seed(0)
    g = (A.Generator64PixResnet if res == 64 else A.Generator32PixResnet)()
    d = (A.Discriminator64PixResnet if res == 64 else A.Discriminator32PixResnet)()
    gan = resnet.ResnetFunctionalGAN(g.state_dict(), d.state_dict(), res)
    zlen = g.len_latent if hasattr(g, 'len_latent') else 128
    del g, d
    times = []
    for _ in range(1 + steps):
        t0 = time.perf_counter()
        gan.g_step(torch.randn(batch, zlen))
        for _ in range(nd):
            gan.d_step(torch.randn(batch, zlen), torch.rand(batch, 3, res, res) * 2 - 1, torch.rand(batch, 1, 1, 1))
        times.append(time.perf_counter() - t0)
    return _baseline_record(torch, batch, batch * nd, times, f'main iteration (1 G-iter + {nd} critic iters) of ResNet '
                            f'GAN {res}^2 (oracle/resnet.py ResnetFunctionalGAN: WGAN + WGAN-GP)', model, n_phys, usable)


# ------------------------------------------------------------------------------------------------------------------
# multi-rank launch
# ------------------------------------------------------------------------------------------------------------------
def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(a):
    """``python bench.py --gpus N`` (N > 1) outside a launcher: start N FRESH ranks through torch.distributed.run as a
    child process (this parent has made no GPU call and makes none: a process that has initialised the GPU must not be
    replaced or forked), relay rank 0's one JSON line to stdout and return the child's exit code."""
    import subprocess
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={a.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # dmabuf IPC: RCCL across processes needs it on this pool
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // a.gpus)))
    print(f'bench.py: starting {a.gpus} ranks: {" ".join(cmd)}', file=sys.stderr)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env)       # stderr is inherited
    record = None
    for raw in proc.stdout:
        line = raw.decode(errors='replace').rstrip('\n')
        if line.startswith('{') and line.endswith('}'):
            record = line
        elif line:
            print(line, file=sys.stderr)
    rc = proc.wait()
    if record is not None:
        print(record, flush=True)
    if rc == 0 and record is None:
        print('bench.py: the ranks exited 0 without a JSON record', file=sys.stderr)
        rc = 1
    return rc


def dry_run_dist(a, json_fd):
    """CPU rehearsal of what an N-rank run does around the kernels (gloo, GANLAB_HOST_LOGIC_ONLY): process group from
    the launcher's environment, the learner's construction (arenas, rank-0 parameter / EWMA broadcast, shared Philox
    seed), then K 'steps' whose backward is a plain torch expression on the arena parameters - enough to fire the
    bucket hooks of parallel.GradReducer, agree on the launch order and mean-reduce the flat gradient arenas exactly as
    d_step / g_step do.  Checks that every rank ends with identical, correctly averaged gradients."""
    import contextlib
    import io
    import torch
    import torch.distributed as dist
    os.environ['GANLAB_HOST_LOGIC_ONLY'] = '1'
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29511')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    torch.manual_seed(1234 + rank)
    from gan_lab_amd import parallel, progressive as P
    from gan_lab_amd.config import make_config
    from gan_lab_amd.stylegan.learner import StyleGANLearner
    P.FMAP_BASE, P.FMAP_MAX = 256, 16
    with contextlib.redirect_stdout(io.StringIO()):
        L = StyleGANLearner(make_config('stylegan', dev='cpu', pin_memory=False, res_samples=32, res_dataset=32,
                                        init_res=32, batch_size=4, len_latent=16, len_dlatent=16, mapping_num_fcs=2,
                                        cutoff_trunc_trick=None, log_every=0, random_seed=7))
    L.reducer = parallel.GradReducer(bucket_mb=0.02)           # several buckets even for this toy network
    ok, launched = True, []

    class _DirectWrite(torch.autograd.Function):
        """What ops.direct_param_grads does on the GPU: the gradient kernel writes the parameter's arena slot itself and the
        Function hands the engine NO gradient for it.  The bucket hooks must still fire for such a parameter (AccumulateGrad
        runs its post-accumulate hooks on an undefined gradient too) - otherwise every single-use parameter would silently
        leave its bucket to start() and the overlap would be gone."""
        @staticmethod
        def forward(ctx, p, coef):
            ctx.p, ctx.coef = p, coef
            return (p.detach() * coef).sum()

        @staticmethod
        def backward(ctx, g):
            ctx.p.grad.add_(g * ctx.coef)
            return None, None
    dist.barrier()
    t0 = time.perf_counter()
    for it in range(a.steps):
        for arena in (L.arena_d, L.arena_g):
            arena.zero_grad()
            coef = float(rank + 1 + it)
            # d loss / d p == coef on this rank; every other parameter takes the direct-write route
            loss = sum(_DirectWrite.apply(p, coef) if i % 2 else (p * coef).sum() for i, p in enumerate(arena.params))
            L.reducer.arm(arena)
            loss.backward()
            L.reducer.start(arena.gflat)
            L.reducer.finish()
            want = sum(float(r + 1 + it) for r in range(world)) / world
            used = torch.cat([arena.gflat[o:o + n] for o, n in zip(arena.offsets, arena.sizes)])
            ok = ok and bool(torch.allclose(used, torch.full_like(used, want), rtol=1e-6, atol=0))
            launched.append(L.reducer.bucket_report(arena)['launched_in_backward'])
    dist.barrier()
    dt = time.perf_counter() - t0
    flags = torch.tensor([int(ok)])
    dist.all_reduce(flags, op=dist.ReduceOp.MIN)
    rep = L.reducer.bucket_report(L.arena_d)
    if rank == 0:
        out = {'metric': 'dry run of the multi-rank path (gloo, host logic only; no GPU)', 'value': None, 'unit': None,
               'n_gpus': world, 'world_size_observed': dist.get_world_size(), 'backend': dist.get_backend(),
               'steps': a.steps, 'warmup': 0, 'ms_per_step': round(dt / max(a.steps, 1) * 1e3, 3), 'dry_run': True,
               'gradients_averaged_correctly_on_every_rank': bool(flags.item()),
               'd_arena_buckets': len(rep['bounds']), 'bucket_order_agreed': rep['agreed'],
               'buckets_launched_inside_backward_per_reduction': launched,
               'every_other_parameter_written_directly': True}
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    dist.barrier()
    dist.destroy_process_group()
    return 0 if bool(flags.item()) else 1


# ------------------------------------------------------------------------------------------------------------------
def main():
    a = parse()
    if a.step_graph != 'auto':
        os.environ['GANLAB_STEP_GRAPH'] = a.step_graph
    env_world = os.environ.get('WORLD_SIZE')
    if env_world is None and a.gpus > 1:
        sys.exit(launch_ranks(a))                   # parent: no torch.cuda call before or after
    world = int(env_world or '1')
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if a.gpus != world:
        # never a silent single-rank (or wrong-size) run: the record's n_gpus must be what was asked for
        print(f'bench.py: --gpus {a.gpus} but the launcher set WORLD_SIZE={world}; start it as "python bench.py --gpus '
              f'{a.gpus}" (it launches the ranks itself) or with --nproc-per-node {a.gpus}', file=sys.stderr)
        sys.exit(2)
    # stdout carries exactly ONE line, the JSON record: everything else that writes to file descriptor 1 on the way
    # (RCCL prints a version banner there when its communicator is created, libraries print notices) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if a.dry_run_dist:
        sys.exit(dry_run_dist(a, json_fd))
    import torch
    import torch.distributed as dist
    a.world = world
    use_dist = world > 1 or os.environ.get('GANLAB_DIST_WORLD1') == '1'   # the knob: RCCL path on one rank
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        if torch.cuda.device_count() < world:
            print(f'bench.py: {world} ranks but only {torch.cuda.device_count()} GPUs are visible', file=sys.stderr)
            sys.exit(2)
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', rank=rank, world_size=world)
        assert dist.get_world_size() == world
    else:
        torch.cuda.set_device(0)
    from gan_lab_amd import _lib, ops as _ops
    _lib.lib()
    torch.manual_seed(1234 + rank)      # parameter init differs per rank on purpose: rank 0's values are broadcast

    wl = Workload(a, torch)             # builds the learner (device RNG seeded from config.random_seed + rank)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        wl.step()
    nb, nc, nr = wl.northstar
    insitu = InSituKernelTimer(torch, _ops, nb, nc, nr)
    insitu_top = InSituKernelTimer(torch, _ops, a.batch, 256, 64)      # config 3: the kernel with the largest share
    flops = FlopCounter(_ops)
    barrier()
    t0 = time.perf_counter()
    with insitu, insitu_top:
        for _ in range(a.steps):
            ld, lg = wl.step()
    barrier()
    dt = time.perf_counter() - t0
    # executed conv FLOPs of a step: counted on ONE more step outside the timed region (every stabilised step launches
    # the same kernels; the Python observer would cost a launch-bound configuration ~1 us per launch inside it)
    with flops:
        prev = os.environ.get('GANLAB_STEP_GRAPH')
        os.environ['GANLAB_STEP_GRAPH'] = '0'          # the observer sits in the Python launchers: count an eager step
        try:
            wl.step()
        finally:
            if prev is None:
                del os.environ['GANLAB_STEP_GRAPH']
            else:
                os.environ['GANLAB_STEP_GRAPH'] = prev
    torch.cuda.synchronize()
    exchange = None
    if use_dist:
        tt = torch.tensor([dt], device='cuda', dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        red = getattr(wl.learner, 'reducer', None)
        if red is not None and hasattr(wl.learner, 'arena_d'):
            rd, rg = red.bucket_report(wl.learner.arena_d), red.bucket_report(wl.learner.arena_g)
            exchange = {'backend': dist.get_backend(), 'bucket_mb': 32,
                        'd_buckets': len(rd['bounds']) if rd else None, 'g_buckets': len(rg['bounds']) if rg else None,
                        'd_launched_inside_backward': rd['launched_in_backward'] if rd else None,
                        'g_launched_inside_backward': rg['launched_in_backward'] if rg else None,
                        'order_agreed': bool(rd and rd['agreed'] and rg and rg['agreed'])}
    ld, lg = (float(ld) if ld is not None else None), (float(lg) if lg is not None else None)
    peak_mem = torch.cuda.max_memory_allocated() / 2 ** 30

    if rank == 0:
        ips = world * wl.images_per_step * a.steps / dt
        peak = PEAK_F32_MFMA_TFLOPS if a.dtype == 'f32' else PEAK_BF16_MFMA_TFLOPS
        exec_tflops = flops.total * a.steps / dt / 1e12      # this rank's executed conv FLOPs per second
        algo = ALGO_TFLOP_PER_IMAGE.get((a.model, a.res))
        name = {'stylegan': 'StyleGAN', 'progan': 'ProGAN', 'resnetgan': 'ResNetGAN'}[a.model]
        metric = f'images/sec (G+D step), {name} {a.res}^2 bs{a.batch}/GPU'
        if a.model == 'resnetgan':
            # NOT the G+D-step definition of the other configurations: every critic iteration's real batch is counted
            metric = (f'real images consumed/sec incl. {wl.nd} critic iters (main iteration = 1 G-iter + {wl.nd} '
                      f'critic iters), {name} {a.res}^2 bs{a.batch}/GPU')
        x3_on = a.dtype == 'f32' and _ops.x3_enabled()
        out = {
            'metric': metric,
            'value': round(ips, 4), 'unit': 'images/sec', 'n_gpus': world,
            'world_size_observed': dist.get_world_size() if use_dist else 1, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(dt / a.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
            # fp32 in, fp32 out, fp32 accumulation everywhere.  The >= 64-channel 3x3 layers (forward, input gradient, plain
            # weight gradient) multiply on the bf16 matrix cores as split products - three bf16 planes per operand that sum to
            # it exactly, six products per fp32 product - as close to float64 as the exact-fp32 MFMA kernels (tools/
            # x3_bench.py: 0.24-0.51 of ATen's rms error); the thin layers and everything else stay on exact fp32 MFMA / VALU.
            # GANLAB_X3=0 restores the exact-fp32 kernels everywhere (profiles/ holds both lines).
            'arithmetic': ('3xbf16 split products, fp32 accumulate (thick 3x3 layers) + exact fp32 MFMA (thin layers)'
                           if (a.dtype == 'f32' and _ops.x3_enabled()) else ('exact fp32 MFMA' if a.dtype == 'f32' else
                                                                             'bf16 products, fp32 accumulate')),
            'config': {'workload': wl.what, 'baseline_config': a.config, 'global_batch': world * a.batch,
                       'per_gpu_batch': a.batch, 'parallelism': f'dp{world}' if world > 1 else 'single'},
            # executed: what the launchers ran (per GPU) over the MFMA peak of one GPU - a roofline fraction of the step
            'executed_tflops_step': round(exec_tflops, 2),
            # with the split-product kernels on, the thick layers' FLOPs run on the bf16 matrix cores (six bf16 products per fp32
            # product): fp32-equivalent FLOPs over the fp32 peak is then a speed label, not a fraction of one pipe's roofline
            'frac_of_mfma_peak_step': None if x3_on else round(exec_tflops / peak, 4),
            'fp32_equivalent_over_fp32_mfma_peak_step': round(exec_tflops / peak, 4) if x3_on else None,
            'executed_conv_tflop_per_step': round(flops.total / 1e12, 3),
            'executed_conv_tflop_by_pass': {k: round(v / 1e12, 3) for k, v in flops.by_kind.items()},
            'conv_launches_per_step': flops.launches,
            # algorithmic: the reference's pass count (4 G + 14 D conv passes) - NOT a roofline fraction:
            # the implementation executes fewer FLOPs (stride-2 fusion, shared D(real), no discarded weight gradients)
            'algorithmic_tflops_step': round(ips / world * algo, 2) if algo else None,
            'loss_d': ld, 'loss_g': lg, 'peak_mem_gib': round(peak_mem, 1),
        }
        if exchange is not None:
            out['gradient_exchange'] = exchange
        sg = getattr(wl.learner, '_step_graph', None)
        out['step_graph'] = {'replayed': bool(sg is not None and sg.graphs),
                             'graphs': len(sg.graphs) if sg is not None else 0}
        out.update(wl.extra)
    del wl
    torch.cuda.empty_cache()
    if rank == 0:
        def with_insitu(r, timer):
            ms_t, n_t = timer.mean_ms()
            if r is not None and ms_t is not None:
                r['isolated_ms_per_launch'], r['isolated_achieved'] = r['ms_per_launch'], r['achieved']
                r['ms_per_launch'] = round(ms_t, 4)
                r['achieved'] = round(r['flops_per_launch'] / (ms_t * 1e-3) / 1e12, 2)
                r['frac'] = round(r['achieved'] / r['peak'], 4)
                if 'fp32_equivalent_tflops' in r:
                    r['isolated_fp32_equivalent_tflops'] = r['fp32_equivalent_tflops']
                    r['fp32_equivalent_tflops'] = round(r['achieved'] / 6.0, 2)
                r['launches_timed_in_step'] = n_t
            return r
        # the instance as launched INSIDE the timed steps (device events around each launch, forward and input
        # gradient): the figure rocprofv3's per-kernel average of the step agrees with; the isolated loop (2 warm-up +
        # 5 launches on a cold chip) is kept beside it as isolated_*
        with _ops.compute_dtype(a.dtype):
            out['roofline'] = None if a.no_roofline else with_insitu(measure_dominant_kernel(torch, nb, nc, nr), insitu)
        torch.cuda.empty_cache()
        if not a.no_roofline and a.config == 3 and a.res == 1024:
            # the kernel with the largest share of the step (profiles/*_step_kernel_stats.csv: conv_fwd_kernel
            # <KS=3,MB=4,32x8>, the 64..512-channel stride-1 layers): its 256->256 @64^2 instance
            out['roofline_top_kernel_by_time'] = with_insitu(measure_dominant_kernel(torch, a.batch, 256, 64), insitu_top)
            torch.cuda.empty_cache()
        out['cpu_baseline'] = None
        if world == 1 and not a.no_cpu_baseline:
            if a.model == 'stylegan':
                out['cpu_baseline'] = cpu_baseline(torch, a.cpu_baseline_res or a.res, a.cpu_baseline_batch,
                                                   a.cpu_baseline_steps)
            elif a.model == 'progan':
                out['cpu_baseline'] = cpu_baseline_progan(torch, a.cpu_baseline_res or a.res, max(a.cpu_baseline_batch, 4),
                                                          a.cpu_baseline_steps)
            else:
                out['cpu_baseline'] = cpu_baseline_resnet(torch, a.res, min(a.batch, 16), a.cpu_baseline_steps)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
