#!/usr/bin/env python3
"""Headline benchmark: images/sec of the full G+D training step (1 D-iteration + 1 G-iteration),
StyleGAN 1024^2, batch 32 per GPU, fp32, nonsaturating loss + R1 (lambda 10) + drift, stabilised phase
at the final resolution, synthetic FFHQ-shaped data (BASELINE.json metric / configs[2]).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
             --master-port P bench.py --gpus N --steps K --warmup W)

Prints ONE JSON line on rank 0.  ``roofline``: the dominant kernel family is the fp32 MFMA
implicit-GEMM conv; its north-star instance (3x3, 16->16 channels, 1024^2, batch 32) is timed with
device events on the launch stream and priced with its algorithmic FLOPs 2*9*16*16*1024^2*32.
``cpu_baseline``: the CPU oracle's (oracle/, kind "port") G+D step on the host cores at a reduced
batch - a reported baseline, not a target.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TFLOP_PER_IMAGE = {1024: 1.536, 128: 0.752}      # 4 G passes + 14 D passes
PEAK_F32_MFMA_TFLOPS = 157.3                     # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2
PEAK_BF16_MFMA_TFLOPS = 2500.0                   # MI355X_MICROARCH.md: dense bf16 MFMA (~2.5 PF, no sparsity)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=3)
    p.add_argument('--warmup', type=int, default=1)
    p.add_argument('--res', type=int, default=1024)
    p.add_argument('--batch', type=int, default=32, help='per-GPU batch')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--cpu-baseline-res', type=int, default=None)
    p.add_argument('--cpu-baseline-batch', type=int, default=1)
    p.add_argument('--no-roofline', action='store_true')
    p.add_argument('--dtype', choices=('f32', 'bf16'), default='f32',
                   help="compute dtype of the 3x3 convolutions; 'bf16' = BASELINE config #2 (use with --res 128 --batch 8)")
    return p.parse_args()


def build_learner(res, batch, device, dtype='f32'):
    from gan_lab_amd.config import make_config
    from gan_lab_amd.stylegan.learner import StyleGANLearner
    bs_dict = {r: batch for r in (4, 8, 16, 32, 64, 128, 256, 512, 1024)}
    cfg = make_config('stylegan', dev='cuda', pin_memory=False, loss='nonsaturating', gradient_penalty='r1',
                      lda=10., res_samples=res, res_dataset=res, init_res=res, batch_size=batch, bs_dict=bs_dict,
                      num_iters_save_model=10 ** 9, log_every=0, compute_dtype=dtype,
                      cutoff_trunc_trick=4 if res >= 64 else None)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        learner = StyleGANLearner(cfg)
    learner.beta = learner.get_smoothing_ewma_beta(half_life=10.)
    learner.gen_model.train()
    learner.disc_model.train()
    return learner


def one_step(learner, real):
    learner.set_requires_grad_disc(True)
    ld = learner.d_step(real, defer_update=True)
    learner.set_requires_grad_disc(False)
    lg = learner.g_step(d_update_pending=True)
    return ld, lg


def measure_dominant_kernel(torch, batch, res, reps=5, dtype='f32', c=None, r=None):
    """Average launch duration of one 3x3 conv kernel instance (default: the north-star layer of the top resolution),
    device events on the launch stream."""
    from gan_lab_amd import ops
    if c is None:
        c = 16 if res >= 1024 else max(16, min(512, 8192 // (res // 2)))
    res = r or res
    x = torch.randn(batch, c, res, res, device='cuda')
    w = torch.randn(c, c, 3, 3, device='cuda')
    g = ops.Geom(batch, c, res, res, c, 3, 1, 0)
    bf = g.bf is not None
    for _ in range(2):
        ops.k_conv_fwd(x, w, None, g, 0.05)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.k_conv_fwd(x, w, None, g, 0.05)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = 2.0 * 9 * c * c * res * res * batch
    ach = flops / (ms * 1e-3) / 1e12
    peak = PEAK_BF16_MFMA_TFLOPS if bf else PEAK_F32_MFMA_TFLOPS
    return {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s',
            'frac': round(ach / peak, 4),
            # HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate
            # passes) on this kernel, profiles/r01g_northstar_conv_pmc.csv: 2.55e9 read (1.19x the algorithmic 2 GiB:
            # the 8-of-72 column halo and the strip-boundary rows that miss L2) + 2.17e9 written;
            # profiles/r01d_northstar_conv_pmc.csv: MFMA pipes busy 0.79-0.80 of the kernel's cycles
            'traffic': 4.722e9 if (c == 16 and res == 1024 and batch == 32 and not bf) else None,
            'kernel': (f'conv_fwd_bf16_kernel<64co x 8x32> {c}->{c} @{res}^2 x{batch}' if bf else
                       f'conv_fwd_roll_kernel<KS=3,16co,64px column strips,4 rows/step> {c}->{c} @{res}^2 x{batch}'
                       if (c <= 16 and res % 64 == 0) else
                       f'conv_fwd_kernel<KS=3,MB={1 if c <= 16 else (2 if c <= 32 else 4)},32x8> '
                       f'{c}->{c} @{res}^2 x{batch}'),
            'ms_per_launch': round(ms, 4), 'flops_per_launch': flops}


class InSituKernelTimer(object):
    """HIP events around every launch of ONE conv geometry (forward and input gradient: the same kernel) during the
    timed steps - the dominant kernel's launch duration in the state the training step actually runs it in (clocks,
    caches, neighbours), rather than in an isolated loop."""

    def __init__(self, torch, ops, batch, c, res):
        self.torch, self.ops, self.key = torch, ops, (batch, c, res, res, c, 3, 1, 0, 0)
        self.events = []
        self._orig = None

    def _match(self, g):
        return (g.N, g.Cin, g.Hin, g.Win, g.Cout, g.ks, g.pad, g.up, g.pool) == self.key and not g.s2 and g.bf is None

    def __enter__(self):
        ops, torch = self.ops, self.torch
        f0, d0 = self._orig = (ops.k_conv_fwd, ops.k_conv_dgrad)     # whatever is installed now: timers nest

        def fwd(x, w, bias, g, *a, **k):
            if not self._match(g):
                return f0(x, w, bias, g, *a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ops._packed(w, ops.PACK_FWD, a[0] if a else k['scale'])     # keep the (cached) weight packing out of the bracket
            e0.record()
            y = f0(x, w, bias, g, *a, **k)
            e1.record()
            self.events.append((e0, e1))
            return y

        def dgrad(gy, w, g, scale):
            if not self._match(g):
                return d0(gy, w, g, scale)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ops._packed(w, ops.PACK_DGRAD, scale)
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


def cpu_baseline(torch, res, batch):
    """The oracle's G+D step (same math, same loss config) on the host cores; bounded sample."""
    from gan_lab_amd import progressive as P
    from gan_lab_amd.progan.architectures import StyleDiscriminator
    from gan_lab_amd.stylegan.architectures import StyleGenerator
    from oracle import nets, step
    torch.manual_seed(0)
    P.StyleGAN.reset_state()
    g = StyleGenerator(final_res=res, blur_type='binomial',
                       truncation_trick_params={'beta': .995, 'psi': .7, 'cutoff_stage': 4 if res >= 64 else None})
    d = StyleDiscriminator(final_res=res, blur_type='binomial')
    import numpy as np
    for _ in range(int(np.log2(res)) - 2):
        g.increase_scale()
        d.increase_scale()
    gan = step.FunctionalGAN(g.state_dict(), d.state_dict(), nets.make_cfg(), model='stylegan',
                             loss='nonsaturating', gp='r1', lda=10., eps_drift=.001)
    del g, d
    L = nets.stylegen_num_layers(gan.g)
    noise = lambda: [torch.randn(batch, 1, 4 * 2 ** (n // 2), 4 * 2 ** (n // 2)) for n in range(L)]  # noqa: E731
    real = torch.rand(batch, 3, res, res) * 2 - 1
    t0 = time.perf_counter()
    gan.d_step(torch.randn(batch, 512), real, noise())
    gan.g_step(torch.randn(batch, 512), noise(), beta=0.999)
    dt = time.perf_counter() - t0
    return {'value': round(batch / dt, 5), 'unit': 'images/sec', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'1 G+D step of StyleGAN {res}^2 at batch {batch} (oracle/step.py FunctionalGAN, torch CPU '
                      f'fp32, {torch.get_num_threads()} threads), {dt:.1f} s'}


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    use_dist = world > 1 or os.environ.get('GANLAB_DIST_WORLD1') == '1'   # the knob: RCCL path on one rank
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    if a.gpus != world and rank == 0 and world > 1:
        print(f'warning: --gpus {a.gpus} but WORLD_SIZE={world}', file=sys.stderr)
    from gan_lab_amd import _lib, rng
    _lib.lib()
    rng.manual_seed(1234, rank)
    torch.manual_seed(1234 + rank)

    learner = build_learner(a.res, a.batch, 'cuda', a.dtype)
    # synthetic FFHQ-shaped reals, resident in HBM before the timed region: U(-1,1) fp32 (B,3,R,R)
    real = torch.rand(a.batch, 3, a.res, a.res, device='cuda') * 2 - 1

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        one_step(learner, real)
    from gan_lab_amd import ops as _ops
    ns_c = 16 if a.res >= 1024 else max(16, min(512, 8192 // (a.res // 2)))
    insitu = InSituKernelTimer(torch, _ops, a.batch, ns_c, a.res)
    insitu_top = InSituKernelTimer(torch, _ops, a.batch, 256, 64)      # the kernel with the largest share of the step
    barrier()
    t0 = time.perf_counter()
    with insitu, insitu_top:
        for _ in range(a.steps):
            ld, lg = one_step(learner, real)
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], device='cuda', dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ld, lg = float(ld), float(lg)
    peak_mem = torch.cuda.max_memory_allocated() / 2 ** 30

    if rank == 0:
        ips = world * a.batch * a.steps / dt
        out = {
            'metric': 'images/sec (G+D step), StyleGAN 1024^2 bs32/GPU' if a.res == 1024 and a.batch == 32 else
                      f'images/sec (G+D step), StyleGAN {a.res}^2 bs{a.batch}/GPU',
            'value': round(ips, 4), 'unit': 'images/sec', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(dt / a.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
            'config': {'workload': f'StyleGAN res_samples={a.res} FFHQ-shaped synthetic, bs={a.batch}/GPU '
                                   f'{"fp32" if a.dtype == "f32" else "bf16 compute / fp32 storage+master"}, '
                                   f'nonsaturating + R1(lambda=10) + drift, stabilised phase, 1 D-iter + 1 G-iter',
                       'global_batch': world * a.batch, 'per_gpu_batch': a.batch,
                       'parallelism': f'dp{world}' if world > 1 else 'single'},
            'achieved_tflops_step': round(ips * TFLOP_PER_IMAGE.get(a.res, 0.0), 2) if a.res in TFLOP_PER_IMAGE
            else None,
            'frac_of_mfma_peak_step': round(
                ips * TFLOP_PER_IMAGE[a.res] /
                ((PEAK_F32_MFMA_TFLOPS if a.dtype == 'f32' else PEAK_BF16_MFMA_TFLOPS) * world), 4)
            if a.res in TFLOP_PER_IMAGE else None,
            'loss_d': ld, 'loss_g': lg, 'peak_mem_gib': round(peak_mem, 1),
        }
    del learner, real
    torch.cuda.empty_cache()
    if rank == 0:
        out['roofline'] = None if a.no_roofline else measure_dominant_kernel(torch, a.batch, a.res, dtype=a.dtype)
        torch.cuda.empty_cache()
        def with_insitu(r, timer):
            ms_t, n_t = timer.mean_ms()
            if r is not None and ms_t is not None:
                r['isolated_ms_per_launch'], r['isolated_achieved'] = r['ms_per_launch'], r['achieved']
                r['ms_per_launch'] = round(ms_t, 4)
                r['achieved'] = round(r['flops_per_launch'] / (ms_t * 1e-3) / 1e12, 2)
                r['frac'] = round(r['achieved'] / r['peak'], 4)
                r['launches_timed_in_step'] = n_t
            return r
        # the same kernel instance as launched INSIDE the timed steps (device events around each launch, forward and
        # input gradient): the figure rocprofv3's per-kernel average of the step agrees with; the isolated loop
        # (2 warm-up + 5 launches on a cold chip) is kept beside it as isolated_*
        out['roofline'] = with_insitu(out['roofline'], insitu)
        if not a.no_roofline and a.res == 1024 and a.dtype == 'f32':
            # the kernel with the largest share of the step (profiles/r01c_step_kernel_stats.csv: conv_fwd_kernel
            # <KS=3,MB=4,32x8>, 15.6% of the step, the 64..512-channel stride-1 layers): its 256->256 @64^2 instance
            out['roofline_top_kernel_by_time'] = with_insitu(
                measure_dominant_kernel(torch, a.batch, a.res, c=256, r=64), insitu_top)
            out['roofline_top_kernel_by_time']['traffic'] = None
            torch.cuda.empty_cache()
        if world == 1 and not a.no_cpu_baseline:
            cres = a.cpu_baseline_res or a.res
            out['cpu_baseline'] = cpu_baseline(torch, cres, a.cpu_baseline_batch)
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
