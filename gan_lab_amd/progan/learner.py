"""ProGANLearner on the HIP path (drop-in surface of gan_lab/progan/learner.py).

Same constructor / ``train(train_dl, valid_dl=None, z_valid_dl=None, num_main_iters=None,
num_gen_iters=None, num_disc_iters=None)`` contract and the same training algorithm
(progan/learner.py:418-1030): phase machine -> D iterations -> G iterations -> EWMA -> alpha -> LR
schedule; but
  * every tensor op of the step runs in the hand-written kernels (gan_lab_amd.ops),
  * the phase machine is the host-only ``schedule.PhaseSchedule`` (pinned to the reference's trace),
  * parameters / gradients / Adam moments / the EWMA shadow live in flat arenas (optim.py),
  * with torch.distributed initialised the step is data parallel over RCCL (parallel.py): the D
    gradient all-reduce overlaps the next G forward,
  * the fade-in of the real images (``up(down(x))*(1-a) + x*a``, :771-779) runs on the device,
  * losses are only synchronised to the host every ``log_every`` iterations.
Validation metrics and sample grids (:223-416, :1187-1234) are inference-only forwards of
the same kernels: ``compute_metrics`` / ``make_image_grid`` below (numbers and a uint8 grid; the matplotlib
figure of the reference is UI and stays out of scope).
"""
import copy
import os
import warnings

import torch
from torch import nn

from .. import ops, parallel, rng
from .._int import LearnerConfigCopy
from ..optim import EwmaTracker, ParamArena
from ..resnetgan.learner import GANLearner
from ..schedule import FINAL, GROW, STABILISE, PhaseSchedule, ewma_beta
from ..utils import backprop_utils as bp
from ..utils.backprop_utils import configure_adam_for_gan
from ..utils.latent_utils import gen_rand_latent_vars
from .architectures import ProDiscriminator, ProGenerator
from .base import ProGAN

NONREDEFINABLE_ATTRS = ('model', 'init_res', 'res_samples', 'res_dataset', 'len_latent', 'num_classes',
                        'class_condition', 'use_auxiliary_classifier', 'model_upsample_type',
                        'model_downsample_type', 'align_corners', 'blur_type', 'nonlinearity', 'use_equalized_lr',
                        'normalize_z', 'use_pixelnorm', 'mbstd_group_size', 'use_ewma_gen',)
REDEFINABLE_FROM_LEARNER_ATTRS = ('batch_size', 'loss', 'gradient_penalty', 'optimizer', 'lr_sched',
                                  'latent_distribution',)
COMPUTE_EWMA_VIA_HALFLIFE = True
EWMA_SMOOTHING_HALFLIFE = 10.
EWMA_SMOOTHING_BETA = .999

_EXCL_G = ['prev_torgb.conv2d.weight', 'prev_torgb.conv2d.bias']
_EXCL_D = ['prev_fromrgb.0.conv2d.weight', 'prev_fromrgb.0.conv2d.bias']


class ProGANLearner(GANLearner):
    """GAN learner for progressively grown architectures (ProGAN here, StyleGAN in the subclass)."""
    _family = ProGAN
    _nonredefinable = NONREDEFINABLE_ATTRS

    def __init__(self, config):
        super().__init__(config)
        self.curr_phase_num = 0
        self.lagged_params = None
        self._progressively_grow = True
        self.sched = None
        self.reducer = parallel.GradReducer()
        self.log_every = getattr(config, 'log_every', 50)
        self.share_gp_forward = True     # see d_step(): common-subexpression elimination of D(real)
        self.last_losses = {}
        # validation bookkeeping (progan/learner.py:207-221)
        self.gen_metrics_num = self.disc_metrics_num = 0
        self.grid_inputs_constructed = False
        self._img_grid_constructed = False
        self.rand_idxs = None
        self.valid_label = None
        self.last_metrics = {}
        if self.model == 'ProGAN':
            self._init_progressive(config, self.__class__.__name__)

    # ------------------------------------------------------------------------------------------------
    def _build_networks(self):
        c = self.config
        gen = ProGenerator(final_res=c.res_samples, len_latent=c.len_latent, upsampler=self.gen_model_upsampler,
                           blur_type=c.blur_type, nl=self.nl, num_classes=self.num_classes_gen,
                           equalized_lr=c.use_equalized_lr, normalize_z=c.normalize_z,
                           use_pixelnorm=c.use_pixelnorm)
        disc = ProDiscriminator(final_res=c.res_samples, pooler=self.disc_model_downsampler, blur_type=c.blur_type,
                                nl=self.nl, num_classes=self.num_classes_disc, equalized_lr=c.use_equalized_lr,
                                mbstd_group_size=c.mbstd_group_size)
        return gen, disc

    def _init_progressive(self, config, learner_name):
        """Common constructor body of ProGANLearner / StyleGANLearner (progan/learner.py:104-206)."""
        self.config = LearnerConfigCopy(config, learner_name, self._nonredefinable, REDEFINABLE_FROM_LEARNER_ATTRS)
        self._is_data_configed = False
        self._update_data_config(raise_exception=False)
        self.latent_distribution = self.config.latent_distribution
        self._family.reset_state()
        rng.seed_from_config(self.config.random_seed)     # latents / per-layer noise: per-rank Philox streams
        self.gen_model, self.disc_model = self._build_networks()
        assert self.config.init_res <= self.config.res_samples
        if self.config.init_res > 4:
            import numpy as np
            l2 = int(np.log2(self.config.init_res))
            if float(self.config.init_res) != 2 ** l2:
                raise ValueError('Only resolutions that are powers of 2 are supported.')
            for _ in range(l2 - 2):
                self.gen_model.increase_scale()
                self.disc_model.increase_scale()
            self.gen_model.fade_in_phase = False
        assert self.gen_model.cls_base.__dict__ == self.disc_model.cls_base.__dict__
        self.gen_model.to(self.config.dev)
        self.disc_model.to(self.config.dev)
        self.batch_size = self.config.bs_dict[self.gen_model.curr_res]
        self._loss = config.loss.casefold()
        self._set_loss()
        self._make_arenas(first=True)
        self._set_optimizer()
        self.eps = self.config.eps_drift > 0
        if parallel.rank() == 0:
            print('-------- Initialized Model Configuration --------')
            print(self.config)
            print('-------------------------------------------------')
            print('\n    Ready to train!\n')

    # ------------------------------------------------------------------------------------------------
    def _make_arenas(self, first=False, old_lagged=None):
        """Flat parameter / gradient arenas for G and D (+ the EWMA shadow of G)."""
        dev = self.config.dev
        self._graph_gen = getattr(self, '_graph_gen', 0) + 1    # captured step graphs point into the old arenas (graphs.py)
        self.reducer.abandon()                   # hooks of the arenas this call replaces (growth / checkpoint load)
        ops.bump_weight_epoch()                  # packed weights of the old arenas (their aliases keep the memory alive)
        self.arena_g = ParamArena(self.gen_model.named_parameters(), dev)
        self.arena_d = ParamArena(self.disc_model.named_parameters(), dev)
        # Replicas start from rank 0's values at construction AND after every growth event: the new blocks and the
        # fresh torgb / fromrgb are drawn from each process's own torch RNG (custom_layers.Conv2dEx), which
        # config.random_seed=-1 seeds differently per rank.  Surviving parameters are already identical on every rank
        # (same averaged gradients, same Adam), so the broadcast only moves what diverged - but it is the whole flat
        # arena in one collective either way (a growth event is rare; 100 MB over xGMI is < 1 ms).
        parallel.broadcast_params(self.arena_g.flat)
        parallel.broadcast_params(self.arena_d.flat)
        self.ewma = None
        self.gen_model_lagged = None
        if self.config.use_ewma_gen:
            if first or old_lagged is None:
                self.ewma = EwmaTracker(self.arena_g)
            else:
                # progan/learner.py:662-684: torgb.* becomes prev_torgb.*, new parameters start from
                # their current value, order follows named_parameters()
                self.ewma = EwmaTracker.__new__(EwmaTracker)
                self.ewma.rebuild(self.arena_g, old_lagged, rename={'torgb.conv2d.weight': 'prev_torgb.conv2d.weight',
                                                                    'torgb.conv2d.bias': 'prev_torgb.conv2d.bias'})
            # built from the (already broadcast) arena + the old shadow, so identical by construction; the explicit
            # broadcast keeps that true even if an old shadow had drifted (e.g. a rank resumed from another file)
            parallel.broadcast_params(self.ewma.flat)
            self.lagged_params = self.ewma.lagged_params

    def _set_optimizer(self):
        """Fresh Adam (state reset) over the current parameter set; prev_torgb / prev_fromrgb are left
        out once the fade-in is over (progan/learner.py:1064-1095)."""
        if self._optimizer != 'adam':
            if self._optimizer in ('rmsprop', 'momentum', 'sgd'):
                raise NotImplementedError(f'{self._optimizer} optimizer not yet implemented.')
            raise ValueError("config does not support this optimizer.\nSupported Optimizers are: "
                             "[ 'adam', 'rmsprop', 'momentum', 'sgd' ]")
        c = self.config
        adam_gan = configure_adam_for_gan(lr_base=c.lr_base, betas=(c.beta1, c.beta2), eps=c.eps, wd=c.wd)
        fade = self.gen_model.fade_in_phase
        self._graph_gen = getattr(self, '_graph_gen', 0) + 1    # ... and at the old optimisers' moment buffers
        self.opt_gen = adam_gan(params=list(self.gen_model.most_parameters(excluded_params=[] if fade else _EXCL_G)))
        self.opt_disc = adam_gan(params=list(self.disc_model.most_parameters(excluded_params=[] if fade else _EXCL_D)))

    def _set_scheduler(self):
        """LambdaLR whose factor follows the current resolution (progan/learner.py:1034-1062)."""
        if self._lr_sched == 'resolution dependent':
            self.scheduler_fn = lambda _: self.config.lr_fctr_dict[self.gen_model.curr_res]
        elif self._lr_sched == 'linear decay':
            self.scheduler_fn = lambda it: 1. - (it + self.sched_stop_step) * (1. / self.num_main_iters)
        elif self._lr_sched == 'custom':
            self.scheduler_fn = eval(self.config.lr_sched_custom)
        else:
            raise ValueError("config does not support this LR scheduler.\nCurrently supported LR Schedulers are: "
                             "[ 'resolution dependent', 'linear decay', 'custom' ]")
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            self.scheduler_gen = torch.optim.lr_scheduler.LambdaLR(self.opt_gen, self.scheduler_fn, last_epoch=-1)
            self.scheduler_disc = torch.optim.lr_scheduler.LambdaLR(self.opt_disc, self.scheduler_fn, last_epoch=-1)

    def _reset_opt_and_sched(self):
        self._set_optimizer()
        if self.sched_bool:
            self.sched_stop_step += self.scheduler_gen._step_count if self.scheduler_gen is not None else 0
            self._set_scheduler()

    def get_smoothing_ewma_beta(self, half_life):
        assert isinstance(half_life, float)
        # the half-life is in IMAGES (progan/learner.py:1124-1127): under data parallelism one G iteration sees
        # batch_size * world_size of them
        return ewma_beta(self.batch_size * parallel.world_size(), self.config.gen_bs_mult, half_life)

    @property
    def progressively_grow(self):
        return self._progressively_grow

    # ------------------------------------------------------------------------------------------------
    # the hot path: one D iteration, one G iteration
    # ------------------------------------------------------------------------------------------------
    def _gen_forward(self, zb, **kw):
        return self.gen_model(zb, **kw)

    @property
    def step_graph(self):
        """The iteration's HIP-graph executor (graphs.GraphedStep), built on first use."""
        if getattr(self, '_step_graph', None) is None:
            from ..graphs import GraphedStep
            self._step_graph = GraphedStep(self)
        return self._step_graph

    def fade_in_real(self, xb):
        """Real images follow the generator's fade-in: up(down(x))*(1-alpha) + x*alpha
        (progan/learner.py:771-779, the non-bit-exact branch), on the device."""
        if not self.gen_model.fade_in_phase:
            return xb
        return ops.lerp(ops.k_up2(ops.k_pool2(xb, 0.25), 1.0), xb, self.gen_model.alpha)

    def _pair_critic_batches(self, xgenb, xb):
        """May the critic score the generated and the real batch in one pass (the penalty-free / WGAN-GP branch of d_step)?
        Only when the minibatch-stddev groups stay inside each half: batch a multiple of the group size.
        GANLAB_CRITIC_PAIR=0: A/B."""
        import os
        if os.environ.get('GANLAB_CRITIC_PAIR') == '0' or xgenb.numel() != xb.numel():
            return False
        key, per_sample = getattr(self, '_pairable', (None, False))
        if key != id(self.disc_model):
            per_sample = not any(isinstance(m, torch.nn.modules.batchnorm._BatchNorm) for m in self.disc_model.modules())
            self._pairable = (id(self.disc_model), per_sample)
        g = getattr(self.disc_model, 'mbstd_group_size', 4)
        n = xgenb.shape[0]
        return per_sample and (g == -1 or (n >= g and n % g == 0))

    def d_step(self, xb, zb=None, defer_update=False, gen_kwargs=None, eps_interp=None):
        """One discriminator iteration (progan/learner.py:734-816).  ``xb``: real batch on the device,
        already at the current resolution.  Returns the (device) loss scalar."""
        c = self.config
        self.arena_d.zero_grad()
        if zb is None:
            zb = gen_rand_latent_vars(num_samples=self.batch_size, length=c.len_latent,
                                      distribution=self.latent_distribution, device=c.dev)
        with torch.no_grad():                    # the generator is frozen in the D step (:752-753)
            xgenb = self._gen_forward(zb, **(gen_kwargs or {}))
        xb = self.fade_in_real(xb)
        gp = self.gradient_penalty
        xr = None
        if self.share_gp_forward and gp in ('r1', 'r2'):
            # R1 penalises the gradient at the real batch itself (R2: at the fake batch): D(real) of the
            # adversarial term and D(real) of the penalty are the SAME forward, so evaluate it once and
            # let one backward sweep carry both cotangents (identical result, 3 fewer D passes).
            xr = (xb if gp == 'r1' else xgenb).detach().view(-1, 3, xb.shape[2], xb.shape[3]).requires_grad_(True)
            d_pen = self.disc_model(xr)
            d_gen, d_real = (self.disc_model(xgenb), d_pen) if gp == 'r1' else (d_pen, self.disc_model(xb))
            loss = self.loss_func_disc(d_gen, d_real) + bp.gp_from_output(d_pen, xr, gp, c.lda, c.gamma)
        else:
            if self._pair_critic_batches(xgenb, xb):
                # one critic pass over [generated; real]: the only layer that looks across samples, the minibatch-stddev
                # statistic, works on CONTIGUOUS groups of 4 (custom_layers.py:117-140: x.view(G, group_size, ...)), which a
                # batch that is a multiple of 4 keeps inside its own half - the outputs are the two separate passes'
                # (progan/learner.py:786-800) and every critic parameter gets one gradient contribution from the pair
                n = xgenb.shape[0]
                out = self.disc_model(torch.cat((xgenb, xb.reshape(xgenb.shape))))
                d_gen, d_real = out[:n], out[n:]
            else:
                d_gen, d_real = self.disc_model(xgenb), self.disc_model(xb)
            loss = self.loss_func_disc(d_gen, d_real)
            if gp is not None:
                loss = loss + self.calc_gp(xgenb, xb, eps_interp=eps_interp)
        if self.eps:
            loss = loss + bp.drift_loss(d_real, c.eps_drift)
        # RCCL mean all-reduce of the flat gradient arena: each ~32 MiB bucket is launched from inside the backward sweep
        # when its last gradient has landed (parallel.GradReducer), the rest by start(); the generator forward of the
        # G step runs while they are in flight and _finish_d_update() waits
        self.reducer.arm(self.arena_d)
        # first-use parameter gradients land in the arena directly (no AccumulateGrad add per parameter)
        with ops.no_grad_towards(xr), ops.direct_param_grads(ops.direct_grads_enabled()):
            loss.backward()            # (no_grad_towards: d loss / d (real batch) is nobody's input - skip that kernel)
        self.reducer.start(self.arena_d.gflat)
        if not defer_update:
            self._finish_d_update()
        return loss.detach()

    def _finish_d_update(self):
        self.reducer.finish()
        self.opt_disc.step()

    def g_step(self, zb=None, d_update_pending=False, gen_kwargs=None):
        """One generator iteration (progan/learner.py:857-916) + EWMA shadow update."""
        c = self.config
        self.arena_g.zero_grad()
        if zb is None:
            zb = gen_rand_latent_vars(num_samples=self.batch_size * c.gen_bs_mult, length=c.len_latent,
                                      distribution=self.latent_distribution, device=c.dev)
        fake = self._gen_forward(zb, **(gen_kwargs or {}))   # needs G weights only -> overlaps the D all-reduce
        if d_update_pending:
            self._finish_d_update()
        loss = self.loss_func_gen(self.disc_model(fake))
        self.reducer.arm(self.arena_g)           # buckets go out while the backward is still producing the others
        with ops.direct_param_grads(ops.direct_grads_enabled()):
            loss.backward()
        self.reducer.start(self.arena_g.gflat)
        self.reducer.finish()                    # Adam needs the averaged gradients; the EWMA follows the update
        self.opt_gen.step()
        if c.use_ewma_gen:
            self.ewma.update(self.beta)
        return loss.detach()

    def set_requires_grad_disc(self, flag):
        for p in self.disc_model.parameters():
            p.requires_grad_(flag)

    # ------------------------------------------------------------------------------------------------
    def _grow(self):
        """Growth event (progan/learner.py:562-685)."""
        old_lagged = dict(self.lagged_params) if self.lagged_params is not None else None
        self.gen_model.increase_scale()
        self.disc_model.increase_scale()
        assert self.gen_model.cls_base.__dict__ == self.disc_model.cls_base.__dict__
        self.gen_model.to(self.config.dev)
        self.disc_model.to(self.config.dev)
        self._make_arenas(old_lagged=old_lagged)
        self.batch_size = self.config.bs_dict[self.gen_model.curr_res]
        self._reset_opt_and_sched()
        self._set_loss()
        self.gen_model.alpha = 0
        if self.config.use_ewma_gen and COMPUTE_EWMA_VIA_HALFLIFE:
            self.beta = self.get_smoothing_ewma_beta(half_life=EWMA_SMOOTHING_HALFLIFE)

    def _bump_loader(self, dl):
        """train_dl duck type: assignable batch size + a Resize in the transform list."""
        if dl is None:
            return
        dl.batch_sampler.batch_size = self.batch_size
        tf = getattr(getattr(getattr(dl, 'dataset', None), 'transforms', None), 'transform', None)
        if tf is not None and hasattr(self, 'increase_real_data_res'):
            tf.transforms = self.increase_real_data_res(transforms_lst=tf.transforms)

    def increase_real_data_res(self, transforms_lst: list):
        """Swap the first Resize for one at the current resolution (progan/learner.py:1099-1112)."""
        res = self.gen_model.curr_res
        n_rsz = 0
        for n, tr in enumerate(transforms_lst):
            if tr.__class__.__name__ == 'Resize':
                n_rsz += 1
                if n_rsz < 2:
                    kw = dict(size=(res, res,))
                    interp = getattr(self.data_config, 'dataset_downsample_type', None) if self.data_config else None
                    if interp is not None:
                        kw['interpolation'] = interp
                    transforms_lst[n] = tr.__class__(**kw)
                else:
                    raise RuntimeWarning('Warning: More than 1 `Resize` transform found; only resized the first '
                                         '`Resize` in transforms list.')
        return transforms_lst

    def _apply_phase_events(self, sched, train_dl=None, valid_dl=None, z_valid_dl=None):
        """Host side of a phase boundary (progan/learner.py:560-729): ask the phase machine what happens before this
        main iteration and do it - grow both networks (+ arenas, replica re-broadcast, fresh Adam / LR schedule,
        EWMA re-keying, loader bump), or reset the optimisers at a stabilise / final boundary."""
        for ev in sched.begin_iter():
            if ev == GROW:
                prev = self.gen_model.curr_res
                self._grow()
                self._bump_loader(train_dl)
                self._bump_loader(valid_dl)
                if z_valid_dl is not None:
                    z_valid_dl.batch_sampler.batch_size = self.batch_size
                if parallel.rank() == 0:
                    print(f'\n\n\nRESOLUTION INCREASED FROM {prev}x{prev} to {self.gen_model.curr_res}x'
                          f'{self.gen_model.curr_res}\n\nFADING IN {self.gen_model.curr_res}x'
                          f'{self.gen_model.curr_res} RESOLUTION...\n')
            elif ev == STABILISE:
                self._reset_opt_and_sched()
                if parallel.rank() == 0:
                    print('\nSTABILIZING...\n')
            elif ev == FINAL:
                self._reset_opt_and_sched()
                self._progressively_grow = False
                if parallel.rank() == 0:
                    print('\nSTABILIZING (FINAL)...\n')
        self.curr_phase_num = sched.curr_phase_num
        assert sched.curr_res == self.gen_model.curr_res and sched.batch_size == self.batch_size

    # ------------------------------------------------------------------------------------------------
    def train(self, train_dl, valid_dl=None, z_valid_dl=None, num_main_iters=None, num_gen_iters=None,
              num_disc_iters=None):
        """Progressive GAN training (see the module docstring); re-entrant like the reference."""
        c = self.config
        num_main_iters = c.num_main_iters if num_main_iters is None else num_main_iters
        num_gen_iters = c.num_gen_iters if num_gen_iters is None else num_gen_iters
        num_disc_iters = c.num_disc_iters if num_disc_iters is None else num_disc_iters
        self.num_main_iters = num_main_iters
        self.dataset_sz = len(train_dl.dataset)
        self.gen_model.to(c.dev).train()
        self.disc_model.to(c.dev).train()
        if self.not_trained_yet:
            self.sched = PhaseSchedule(self.gen_model.curr_res, self.gen_model.final_res, c.bs_dict,
                                       c.nimg_transition, num_disc_iters, world_size=parallel.world_size())
            self.beta = None
            if c.use_ewma_gen:
                self.beta = self.get_smoothing_ewma_beta(half_life=EWMA_SMOOTHING_HALFLIFE) \
                    if COMPUTE_EWMA_VIA_HALFLIFE else EWMA_SMOOTHING_BETA
            self.train_dataiter = iter(train_dl)
            if parallel.rank() == 0:
                print('STARTING FROM ITERATION 0:\n')
        else:
            if self.sched is None:      # checkpoint without phase bookkeeping: start the phases at this resolution
                self.sched = PhaseSchedule(self.gen_model.curr_res, self.gen_model.final_res, c.bs_dict,
                                           c.nimg_transition, num_disc_iters, world_size=parallel.world_size())
            if self.pretrained_model and getattr(self, 'train_dataiter', None) is None:
                self._bump_loader(train_dl)         # resume at the checkpoint's resolution / batch size
                self._bump_loader(valid_dl)
                self.train_dataiter = iter(train_dl)
            if parallel.rank() == 0:
                print('CONTINUING FROM WHERE YOU LEFT OFF:\n')
        if self.sched_bool:
            if self.scheduler_gen is None and not self.pretrained_model:
                self.sched_stop_step = 0
            self._set_scheduler()
        sched = self.sched
        self.nimg_transition_lst = sched.nimg_transition_lst

        try:
            for itr in range(num_main_iters):
                self.set_requires_grad_disc(True)
                self._apply_phase_events(sched, train_dl, valid_dl, z_valid_dl)

                valid_due = (itr + 1) % c.num_iters_valid == 0 or itr == 0
                metrics_due = valid_due and z_valid_dl is not None and (bool(c.gen_metrics) or
                                                                        (valid_dl is not None and bool(c.disc_metrics)))
                if self.use_step_graph and num_disc_iters == 1 and num_gen_iters == 1 and not metrics_due and \
                        self.step_graph.eligible():
                    # stabilised phase, single process: the whole iteration replayed as HIP graphs (graphs.GraphedStep)
                    batch = next(self.train_dataiter, None)
                    if batch is None:
                        self.curr_epoch_num += 1
                        self.train_dataiter = iter(train_dl)
                        batch = next(self.train_dataiter)
                    xb = batch[0].to(c.dev, non_blocking=True).float()
                    loss_d, loss_g = self.step_graph(xb)
                    self.curr_dataset_batch_num += 1
                    sched.after_d_iter()
                    self.curr_img_num = sched.curr_img_num
                else:
                    # ------------------------- TRAIN DISCRIMINATOR -------------------------
                    for disc_iter in range(num_disc_iters):
                        batch = next(self.train_dataiter, None)
                        if batch is None:
                            self.curr_epoch_num += 1
                            self.train_dataiter = iter(train_dl)
                            batch = next(self.train_dataiter)
                        xb = batch[0].to(c.dev, non_blocking=True).float()
                        last = disc_iter == num_disc_iters - 1
                        valid_now = last and ((itr + 1) % c.num_iters_valid == 0 or itr == 0)
                        d_metrics = valid_now and z_valid_dl is not None and valid_dl is not None and bool(c.disc_metrics)
                        loss_d = self.d_step(xb, defer_update=last and num_gen_iters > 0 and not d_metrics)
                        if d_metrics:       # validation metrics of the just-updated discriminator (:822-832)
                            vals = self.compute_metrics(metrics=c.disc_metrics, metrics_type='Discriminator',
                                                        z_valid_dl=z_valid_dl, valid_dl=valid_dl)
                            if parallel.rank() == 0:
                                print('|\n', 'Discriminator Validation Metrics:\n', *vals)
                        self.curr_dataset_batch_num += 1
                        sched.after_d_iter()
                        self.curr_img_num = sched.curr_img_num

                    # --------------------------- TRAIN GENERATOR ---------------------------
                    self.set_requires_grad_disc(False)
                    loss_g = None
                    for gen_iter in range(num_gen_iters):
                        loss_g = self.g_step(d_update_pending=(gen_iter == 0))
                        if gen_iter == num_gen_iters - 1 and z_valid_dl is not None and c.gen_metrics and \
                                ((itr + 1) % c.num_iters_valid == 0 or itr == 0):     # (:921-928)
                            vals = self.compute_metrics(metrics=c.gen_metrics, metrics_type='Generator',
                                                        z_valid_dl=z_valid_dl, valid_dl=None)
                            if parallel.rank() == 0:
                                print('|\n', 'Generator Validation Metrics:\n', *vals)
                    if num_gen_iters == 0:
                        self._finish_d_update()

                # alpha, LR schedule (progan/learner.py:951-956)
                sched.end_iter()
                if self.gen_model.fade_in_phase:
                    if sched.fade_in_phase:
                        self.gen_model.alpha = sched.alpha
                    else:
                        self.gen_model.alpha = 1      # snaps and leaves the fade-in phase for both networks
                if self.sched_bool:
                    with warnings.catch_warnings():
                        warnings.simplefilter('ignore')
                        self.scheduler_gen.step()
                        self.scheduler_disc.step()
                self.not_trained_yet = False
                if self.log_every and (itr % self.log_every == 0 or itr == num_main_iters - 1):
                    self.last_losses = dict(itr=itr, loss_d=float(loss_d), loss_g=float(loss_g) if loss_g is not None
                                            else None, res=self.gen_model.curr_res, alpha=float(self.gen_model.alpha),
                                            fade_in=bool(self.gen_model.fade_in_phase), batch=self.batch_size)
                    if parallel.rank() == 0:
                        print(('%9s' * 6) % (f'{self.curr_epoch_num}', f'{self.gen_model.curr_res}X'
                                             f'{self.gen_model.curr_res}',
                                             'Fade In' if self.gen_model.fade_in_phase else 'Stab.',
                                             '%.4g' % self.last_losses['loss_d'],
                                             '%.4g' % (self.last_losses['loss_g'] or 0.), itr))
                if (itr + 1) % c.num_iters_save_model == 0:
                    self.save_model(c.save_model_dir / (self.model.casefold().replace(' ', '') + '_model.tar'))
        except KeyboardInterrupt:
            # progan/learner.py:986-1013: Ctrl-C saves the latest checkpoint before the run ends.  The signal reaches the
            # ranks at different points of the step (one may already have its gradient all-reduce in flight), so NO
            # collective runs here: pending reductions are dropped, rank 0 writes atomically, nobody waits at a barrier
            self.set_requires_grad_disc(True)
            self.reducer.abandon()
            if not self.not_trained_yet:
                self.save_model(c.save_model_dir / (self.model.casefold().replace(' ', '') + '_model.tar'), sync=False)
                if parallel.rank() == 0:
                    print(f'\nTraining interrupted. Saved latest checkpoint into "{c.save_model_dir}/".\n')
            raise
        self.set_requires_grad_disc(True)

    # ------------------------------------------------------------------------------------------------
    # validation metrics and sample grids: inference-only forwards
    # ------------------------------------------------------------------------------------------------
    def _update_gen_lagged(self):
        """progan/learner.py:234-242: refresh ``gen_model_lagged`` from the EWMA shadow (train mode, like upstream)."""
        g = self.materialize_lagged_generator()
        g.to(self.config.dev).train()
        return g

    @torch.no_grad()
    def compute_metrics(self, metrics, metrics_type, z_valid_dl, valid_dl=None):
        """Metric evaluation over the validation latents / images (progan/learner.py:249-416), same definitions:
        every metric is ``sum over the (batch, n_batches) table / len(z_valid_dl.dataset)``, i.e. 'fake realness' /
        'real realness' are mean logits and the two losses are batch-size-weighted means of per-batch losses
        (the gradient penalty is not included, :393-395).  Networks run in eval mode on the training device
        (``config.metrics_dev`` is ignored: there is no CPU path) and return to train mode.  Returns the
        reference's list of formatted lines; the raw numbers are kept in ``self.last_metrics``."""
        c = self.config
        metrics_type = metrics_type.casefold()
        if metrics_type not in ('generator', 'critic', 'discriminator',):
            raise Exception('Invalid metrics_type. Only "generator", "critic", or "discriminator" are accepted.')
        metrics = [m.casefold() for m in metrics]
        want_grid = 'image grid' in metrics and metrics_type == 'generator'
        if want_grid and (self.ds_mean is None or self.data_config is None):
            self._update_data_config(raise_exception=True)
        self.disc_model.eval()
        if c.use_ewma_gen and metrics_type == 'generator':
            self.gen_model.train()
            self._update_gen_lagged()
            self.gen_model_lagged.eval()
        self.gen_model.eval()
        try:
            n_batches, n_z = len(z_valid_dl), len(z_valid_dl.dataset)
            valid_iter = iter(valid_dl) if valid_dl is not None else None
            if want_grid and not self.grid_inputs_constructed:
                assert c.img_grid_sz ** 2 <= n_z
                self.rand_idxs = torch.multinomial(torch.ones(n_z), num_samples=c.img_grid_sz ** 2,
                                                   replacement=False)
                self._grid_fill = 0
            self._img_grid_constructed = False
            table = {m: torch.zeros(self.batch_size, n_batches, device=c.dev) for m in metrics}
            for n, zbatch in enumerate(z_valid_dl):
                zb = zbatch[0].to(c.dev).float()
                gen_labels = zbatch[1].cpu() if len(zbatch) > 1 else None
                k = len(zb)
                xgen = self.gen_model(zb)
                y_fake = None
                if 'fake realness' in metrics:
                    y_fake = self.disc_model(xgen)
                    table['fake realness'][:k, n] = y_fake
                if metrics_type == 'generator':
                    if 'generator loss' in metrics:
                        if y_fake is None:
                            y_fake = self.disc_model(xgen)
                        table['generator loss'][:k, n] = self.loss_func_gen(y_fake)
                    if want_grid:
                        self._collect_grid_inputs(zb, gen_labels, n)
                        if self.grid_inputs_constructed and not self._img_grid_constructed:
                            self._save_metric_grids()
                            self._img_grid_constructed = True
                elif valid_iter is not None:
                    xb = next(valid_iter)[0].to(c.dev).float()
                    xb = self.fade_in_real(xb)          # reals follow the generator's fade-in (:361-371)
                    y_real = None
                    if 'real realness' in metrics:
                        y_real = self.disc_model(xb)
                        table['real realness'][:k, n] = y_real
                    if 'discriminator loss' in metrics:
                        if y_fake is None:
                            y_fake = self.disc_model(xgen)
                        if y_real is None:
                            y_real = self.disc_model(xb)
                        table['discriminator loss'][:k, n] = self.loss_func_disc(y_fake, y_real)
            if metrics_type == 'generator':
                self.gen_metrics_num += 1
            else:
                self.disc_metrics_num += 1
            vals = {m: float(t.sum() / n_z) for m, t in table.items() if m != 'image grid'}
        finally:
            self.gen_model.train()
            self.disc_model.train()
        self.last_metrics[metrics_type] = vals
        width = '%-' + str(max(len(m) for m in metrics) + 3) + 's'
        return ['    ' + (width % (m + ':')) + '%.4g' % vals[m] + '\n' for m in metrics if m != 'image grid']

    def _collect_grid_inputs(self, zb, gen_labels, n):
        """Pick the img_grid_sz^2 randomly chosen validation latents (fixed across calls, :311-330)."""
        c = self.config
        if self.valid_z is None:
            self.valid_z = torch.empty(c.img_grid_sz ** 2, zb.shape[1], device=c.dev)
            if gen_labels is not None and c.img_grid_show_labels:
                self.valid_label = torch.zeros(c.img_grid_sz ** 2, dtype=torch.long)
        if self.grid_inputs_constructed:
            return
        chosen = set(self.rand_idxs.tolist())
        for o in range(len(zb)):
            if n * self.batch_size + o in chosen:
                self.valid_z[self._grid_fill] = zb[o]
                if self.valid_label is not None and gen_labels is not None:
                    self.valid_label[self._grid_fill] = gen_labels[o]
                self._grid_fill += 1
        if self._grid_fill == c.img_grid_sz ** 2:
            self.grid_inputs_constructed = True

    def _save_metric_grids(self):
        """samples/<model>/<dataset>/image_grid/{time_averaged,original}/<n>.png (:332-349)."""
        c = self.config
        root = c.save_samples_dir / self.model.casefold().replace(' ', '') / self.data_config.dataset / 'image_grid'
        kinds = (('time_averaged', True),) if c.use_ewma_gen else ()
        for sub, avg in kinds + (('original', False),):
            (root / sub).mkdir(parents=True, exist_ok=True)
            self.make_image_grid(zs=self.valid_z, labels=self.valid_label, time_average=avg,
                                 save_path=str(root / sub / (str(self.gen_metrics_num) + '.png')))

    def _check_sample_latents(self, zs):
        if self.ds_mean is None or self.ds_std is None:
            raise ValueError("This model does not hold any information about your dataset's mean and/or std.\n"
                             "Please provide these (either from your current data configuration or from your "
                             "pretrained model).")
        want = self.config.len_latent + (self.num_classes_gen if self.cond_gen else 0)
        if zs.shape[-1] != want:
            raise IndexError(f'Input latent vector must be of size {want}.')

    @torch.no_grad()
    def generate(self, zs, time_average=True, graph=False, **gen_kwargs):
        """Samples in [0, 1] image space, (N, 3, R, R) on the device: ``G(z) * ds_std + ds_mean`` from the EWMA
        generator (``time_average``) or the snapshot one - what plot_sample / make_image_grid display
        (:1148-1234).  The networks' modes are left as the caller set them.  ``graph=True`` (eval-mode generator,
        at most 16 latents, no extra forward arguments) replays a captured hipGraph of the forward
        (gan_lab_amd/graphs.py): the latency-bound serving path."""
        self._check_sample_latents(zs)
        if time_average and self.gen_model_lagged is None:
            self._update_gen_lagged()
            self.gen_model_lagged.eval()
        gen = self.gen_model_lagged if time_average else self.gen_model
        dev = self.config.dev
        std, mean = self.ds_std.to(dev).view(1, -1, 1, 1), self.ds_mean.to(dev).view(1, -1, 1, 1)
        if graph and not gen_kwargs and not gen.training and len(zs) <= 16:
            from ..graphs import GraphedGenerator
            key = (id(gen), len(zs), gen.curr_res, bool(gen.fade_in_phase))
            cache = self.__dict__.setdefault('_graphed_generators', {})
            if key not in cache:
                cache.clear()                    # one live capture: a grown / re-materialised generator replaces it
                cache[key] = GraphedGenerator(gen, len(zs), len_z=zs.shape[-1])
            return cache[key](zs.to(dev).float()) * std + mean
        out = [gen(zs[i:i + 16].to(dev).float(), **gen_kwargs) * std + mean for i in range(0, len(zs), 16)]
        return torch.cat(out)

    @torch.no_grad()
    def make_image_grid(self, zs, labels=None, time_average=True, save_path=None):
        """sqrt(len(zs)) x sqrt(len(zs)) grid of generated samples as a uint8 (H, W, 3) array, saved as a PNG
        when ``save_path`` is given (progan/learner.py:1187-1234; the matplotlib figure of the reference is
        replaced by the pixel grid itself, one generated pixel per grid pixel, labels are not drawn)."""
        import numpy as np
        if not zs.dim() == 2:
            raise IndexError('Incorrect dimensions of input latent vector. Must be `dim == 2`.')
        self._check_sample_latents(zs)
        if np.sqrt(len(zs)) % 1 != 0:
            raise ValueError('Argument `zs` must be a perfect square-length in order to make image grid.')
        sz = int(np.sqrt(len(zs)))
        x = self.generate(zs, time_average=time_average).clamp_(0., 1.)
        r = x.shape[-1]
        grid = (x.view(sz, sz, 3, r, r).permute(0, 3, 1, 4, 2).reshape(sz * r, sz * r, 3) * 255.).round()
        grid = grid.to(torch.uint8).cpu().numpy()
        if save_path is not None:
            from PIL import Image
            Image.fromarray(grid).save(save_path)
        return grid

    # ------------------------------------------------------------------------------------------------
    def materialize_lagged_generator(self):
        """EWMA generator as a module (progan/learner.py:223-242): a deep copy of G with the lagged
        parameter values loaded."""
        g = copy.deepcopy(self.gen_model)
        if self.config.use_ewma_gen and self.lagged_params is not None:
            sd = g.state_dict()
            for k, v in self.lagged_params.items():
                sd[k] = v.detach().clone()
            g.load_state_dict(sd)
        self.gen_model_lagged = g
        return g

    # -- checkpoints ---------------------------------------------------------------------------------------------
    def _param_name_sets(self):
        """Names the optimisers cover, in the reference's ``most_parameters`` order (progan/learner.py:1064-1095)."""
        fade = self.gen_model.fade_in_phase
        return ([k for k, _ in self.gen_model.named_parameters() if fade or k not in _EXCL_G],
                [k for k, _ in self.disc_model.named_parameters() if fade or k not in _EXCL_D])

    def _extra_checkpoint_fields(self):
        """Family-specific additions to the checkpoint dict (StyleGAN: truncation-trick state)."""
        return {}

    def _restore_extra_fields(self, ck):
        pass

    def _sched_steps(self):
        """Main iterations the live LambdaLR has stepped through (a fresh scheduler's ``_step_count`` is 1)."""
        return max(self.scheduler_gen._step_count - 1, 0) if (self.sched_bool and self.scheduler_gen is not None) else 0

    def save_model(self, save_path, reference_format=False, sync=True):
        """Checkpoint.  Default: this package's plain-data format (tensors + builtin types only; key names follow
        progan/learner.py:1257-1298).  ``reference_format=True``: the dict the reference's own ``save_model`` writes -
        same key set, ``config`` / ``lagged_params`` pickled under the reference's class names, torch-Adam optimiser
        state dicts, the ``nl`` / resampler modules - so the reference's ``load_model`` (:1305-1448) reads it
        (``checkpoint.reference_checkpoint_dict``).  Under data parallelism only rank 0 writes (replicas are identical),
        through a temporary file + ``os.replace``; every rank waits for the file to be complete (``sync=False``: no
        barrier - the interrupt path, where the ranks are not at the same point of the step)."""
        from .. import checkpoint as ckpt
        if self.not_trained_yet and reference_format:
            raise Exception('Please train your model for atleast 1 iteration before saving.')
        if parallel.rank() == 0:
            g_names, d_names = self._param_name_sets()
            if reference_format:
                ck = ckpt.reference_checkpoint_dict(self, g_names, d_names, extra=self._extra_checkpoint_fields())
                ckpt.save_atomic(ck, save_path, foreign=True)
            else:
                ckpt.save_atomic(self._plain_checkpoint_dict(), save_path)
        if sync:
            parallel.barrier()

    def _plain_checkpoint_dict(self):
        cpu = lambda sd: {k: v.detach().cpu() for k, v in sd.items()}  # noqa: E731
        lagged = self.materialize_lagged_generator() if self.config.use_ewma_gen else None
        tcpu = lambda v: None if v is None else v.detach().cpu()  # noqa: E731
        ck = {
            'config': {k: v for k, v in vars(self.config).items() if not k.startswith('_') and
                       isinstance(v, (int, float, str, bool, dict, list, tuple, type(None)))},
            'curr_res': self.gen_model.curr_res,
            'alpha': self.gen_model.alpha,
            'gen_model_state_dict': cpu(self.gen_model.state_dict()),
            'disc_model_state_dict': cpu(self.disc_model.state_dict()),
            'gen_model_lagged_state_dict': None if lagged is None else cpu(lagged.state_dict()),
            'opt_gen_state_dict': self.opt_gen.export_moments(self.gen_model.named_parameters()),
            'opt_disc_state_dict': self.opt_disc.export_moments(self.disc_model.named_parameters()),
            'batch_size': self.batch_size,
            # LambdaLR bookkeeping: steps taken by the live scheduler are folded in, so a 'linear decay' schedule resumes
            # where it stopped (the reference keeps the scheduler state dicts for this, :1249-1252)
            'sched_stop_step': (self.sched_stop_step or 0) + self._sched_steps() if self.sched_bool else
            self.sched_stop_step,
            'lr_sched': self.lr_sched, 'optimizer': self.optimizer,
            'loss': self.loss, 'gradient_penalty': self.gradient_penalty,
            'latent_distribution': self.latent_distribution,
            'curr_dataset_batch_num': self.curr_dataset_batch_num, 'curr_epoch_num': self.curr_epoch_num,
            'tot_num_epochs': self.tot_num_epochs, 'dataset_sz': getattr(self, 'dataset_sz', None),
            'progressively_grow': self.progressively_grow,
            'curr_img_num': self.curr_img_num,
            'curr_phase_num': self.curr_phase_num,
            'nimg_transition_lst': list(self.sched.nimg_transition_lst) if self.sched else None,
            'not_trained_yet': self.not_trained_yet,
            'ds_mean': tcpu(self.ds_mean), 'ds_std': tcpu(self.ds_std),
            'valid_z': tcpu(self.valid_z), 'valid_label': tcpu(self.valid_label), 'rand_idxs': tcpu(self.rand_idxs),
            'grid_inputs_constructed': self.grid_inputs_constructed,
            'gen_metrics_num': self.gen_metrics_num, 'disc_metrics_num': self.disc_metrics_num,
            # device Philox stream (latents, per-layer noise): a resumed run continues the sequence instead of replaying
            # the start of the original one
            'rng_state': dict(rng._STATE),
        }
        ck.update({k: (tcpu(v) if torch.is_tensor(v) else v) for k, v in self._extra_checkpoint_fields().items()})
        return ck

    def load_model(self, load_path, dev_of_saved_model='cpu'):
        """Restore networks (replaying ``increase_scale`` up to the saved resolution, progan/learner.py:
        1348-1360), EWMA shadow, Adam moments, the phase machine, dataset statistics and validation-grid state.  Reads
        this package's plain-data checkpoints AND files written by the reference's own ``save_model``
        (``checkpoint.py``)."""
        from .. import checkpoint as ckpt
        ck = ckpt.load_checkpoint(load_path, dev_of_saved_model)
        ref = ckpt.is_reference_format(ck)
        ckpt.check_architecture(ckpt.config_dict(ck), self.config)
        self._family.reset_state()
        self.gen_model, self.disc_model = self._build_networks()
        import numpy as np
        for _ in range(int(np.log2(ck['curr_res'])) - 2):
            self.gen_model.increase_scale()
            self.disc_model.increase_scale()
        self.gen_model.load_state_dict(ck['gen_model_state_dict'])
        self.disc_model.load_state_dict(ck['disc_model_state_dict'])
        fade = ck['alpha'] != 1
        self.gen_model.fade_in_phase = fade
        self.gen_model.alpha = ck['alpha'] if fade else 1
        self.gen_model.to(self.config.dev)
        self.disc_model.to(self.config.dev)
        self.batch_size = ck.get('batch_size', self.config.bs_dict[self.gen_model.curr_res])
        self._make_arenas(first=True)
        lag = ck.get('lagged_params') if ref else None
        if lag is None and ck.get('gen_model_lagged_state_dict') is not None:
            lag = ck['gen_model_lagged_state_dict']
        if lag is not None and self.lagged_params is not None:
            with torch.no_grad():
                for k, v in lag.items():
                    if k in self.lagged_params:
                        self.lagged_params[k].copy_(v.to(self.config.dev))
        self._restore_extra_fields(ck)
        for attr in ('loss', 'gradient_penalty'):
            if ck.get(attr) is not None:
                setattr(self, attr, ck[attr])
        self._set_optimizer()
        g_names, d_names = self._param_name_sets()
        mg, md = ck.get('opt_gen_state_dict'), ck.get('opt_disc_state_dict')
        if ref:
            mg, md = ckpt.moments_from_torch_adam(mg, g_names), ckpt.moments_from_torch_adam(md, d_names)
        if mg is not None and mg.get('exp_avg'):
            self.opt_gen.import_moments(self.gen_model.named_parameters(), mg)
            self.opt_disc.import_moments(self.disc_model.named_parameters(), md)
        for k in ('sched_stop_step', 'curr_dataset_batch_num', 'curr_epoch_num', 'tot_num_epochs', 'dataset_sz',
                  'curr_img_num', 'curr_phase_num', 'not_trained_yet', 'latent_distribution', 'grid_inputs_constructed',
                  'gen_metrics_num', 'disc_metrics_num', 'rand_idxs', 'valid_label'):
            if ck.get(k) is not None:
                setattr(self, k, ck[k])
        # (reference-written files: ``sched_stop_step`` is taken as stored.  The reference also stores the LambdaLR
        # state dicts but its ``_set_scheduler`` overwrites them with the fresh scheduler's before loading them back
        # (:1055-1062), so its own resume restarts the step count at ``sched_stop_step`` too.)
        if ck.get('valid_z') is not None:
            self.valid_z = ck['valid_z'].to(self.config.dev)
        # dataset statistics: the pretrained model's by default (:1431-1436), also pushed into a live data config
        if ck.get('ds_mean') is not None and ck.get('ds_std') is not None:
            self.ds_mean, self.ds_std = ck['ds_mean'].float().cpu(), ck['ds_std'].float().cpu()
            if self._is_data_configed and self.data_config is not None:
                self.data_config.ds_mean = self.ds_mean.squeeze().tolist()
                self.data_config.ds_std = self.ds_std.squeeze().tolist()
        if ck.get('nimg_transition_lst') is not None:
            self.sched = PhaseSchedule(self.gen_model.curr_res, self.gen_model.final_res, self.config.bs_dict,
                                       self.config.nimg_transition, self.config.num_disc_iters,
                                       world_size=parallel.world_size())
            self.sched.restore(self.gen_model.curr_res, ck['curr_img_num'], ck['curr_phase_num'],
                               ck['nimg_transition_lst'], self.gen_model.alpha,
                               ck.get('progressively_grow', True))
            self.sched.batch_size = self.batch_size
            self._progressively_grow = self.sched.progressively_grow
        self.beta = None
        if self.config.use_ewma_gen:
            self.beta = self.get_smoothing_ewma_beta(half_life=EWMA_SMOOTHING_HALFLIFE) \
                if COMPUTE_EWMA_VIA_HALFLIFE else EWMA_SMOOTHING_BETA
        self.scheduler_gen = self.scheduler_disc = None
        self.train_dataiter = None
        self.pretrained_model = True
        st = ck.get('rng_state')
        if st is not None:
            # the learner owns the device stream: the saved offset always resumes; the seed is rank 0's, so it is only
            # taken over by a single process (data-parallel ranks keep their own rank-mixed seed of this construction)
            rng._STATE['offset'] = int(st['offset'])
            if parallel.world_size() == 1:
                rng._STATE['seed'] = int(st['seed'])
