"""GANLearner on the HIP path (drop-in surface of gan_lab/resnetgan/learner.py).

Two roles, as in the reference:
  * base class of ProGANLearner / StyleGANLearner (:87-301, :780-946): supervision flags, resampler /
    nonlinearity selection, loss + gradient-penalty + optimiser + LR-scheduler plumbing;
  * the learner of the non-progressive ResNet GANs (BASELINE config #5, ``config.model ==
    'ResNet GAN'``): 32 / 64 pixel BatchNorm generator + LayerNorm critic, ``train()`` = per main
    iteration ``num_gen_iters`` generator iterations with the critic frozen, then ``num_disc_iters``
    critic iterations (:463-700), WGAN + WGAN-GP by default.
Every tensor op of the step runs in the hand-written kernels (gan_lab_amd.ops); parameters, gradients
and Adam moments live in flat arenas (optim.py); with torch.distributed initialised the gradients are
mean-all-reduced over RCCL (BatchNorm statistics stay per rank, like DDP without SyncBN).
Validation metrics, image grids and plotting (:249-461, :950-1046) are outside the hot path."""
import os
import warnings

import torch

from .. import _lib, ops, parallel
from .._int import FMAP_SAMPLES, LearnerConfigCopy, get_current_configuration  # noqa: F401
from ..optim import ParamArena
from ..utils import backprop_utils as bp
from ..utils.backprop_utils import configure_adam_for_gan
from ..utils.custom_layers import LeakyReLU, Tanh, make_downsampler, make_upsampler
from ..utils.latent_utils import gen_rand_latent_vars

NONREDEFINABLE_ATTRS = ('model', 'res_samples', 'res_dataset', 'len_latent', 'num_classes', 'class_condition',
                        'use_auxiliary_classifier', 'model_upsample_type', 'model_downsample_type',
                        'align_corners', 'blur_type', 'nonlinearity', 'use_equalized_lr',)
REDEFINABLE_FROM_LEARNER_ATTRS = ('batch_size', 'loss', 'gradient_penalty', 'optimizer', 'lr_sched',)


class GANLearner(object):
    def __init__(self, config):
        super().__init__()
        self._model = config.model
        self.pretrained_model = False
        dev = torch.device(config.dev)
        # GANLAB_HOST_LOGIC_ONLY=1 lets the CPU tests drive the learner's HOST logic (construction, growth events,
        # arenas, optimiser / EWMA bookkeeping, the data-parallel parameter broadcast over gloo).  It does not add a
        # CPU compute path: every tensor op of a step still raises in gan_lab_amd.ops for a non-GPU tensor.
        if dev.type != 'cuda' and os.environ.get('GANLAB_HOST_LOGIC_ONLY') != '1':
            raise RuntimeError(f"gan_lab_amd runs on the MI355X only (config.dev={config.dev!r}); there is no CPU "
                               f"path - use the reference or the test oracle for CPU runs")
        _lib.lib()  # fail now, loudly, if the kernel library is missing
        from .. import ops
        ops.set_compute_dtype(getattr(config, 'compute_dtype', 'f32'))   # 'bf16': BASELINE config #2

        self.curr_dataset_batch_num = 0
        self.curr_epoch_num = 1
        # supervised / unsupervised selection (resnetgan/learner.py:122-138)
        self.num_classes = 0
        self.cond_gen = self.cond_disc = self.ac = False
        self.num_classes_gen = self.num_classes_disc = 0
        if config.use_auxiliary_classifier or config.class_condition:
            # SURVEY.md §8f item 4.  No variant of these options runs in the reference (every combination raises in its
            # constructor or first iteration: tests/golden/probe_conditional.py -> tests/golden/conditional_probe.json),
            # so there is no behaviour to be in parity with.
            raise NotImplementedError('class_condition / use_auxiliary_classifier: no variant of these options runs in '
                                      'the reference (tests/golden/conditional_probe.json); not provided')
        if not (config.res_samples <= config.res_dataset):
            raise ValueError(f'Resolution of generated images (config.res_samples = {config.res_samples}) must be '
                             f'less than\nor equal to resolution of dataset (config.res_dataset = '
                             f'{config.res_dataset}) at all times.\nPlease set config.res_samples <= '
                             f'config.res_dataset.')
        # resamplers (:147-176)
        self.gen_model_upsampler = make_upsampler(config.model_upsample_type, config.align_corners)
        self.disc_model_downsampler = make_downsampler(config.model_downsample_type, config.align_corners)
        # nonlinearity (:178-184)
        nl = config.nonlinearity.casefold()
        if nl == 'leaky relu':
            self.nl = LeakyReLU(negative_slope=config.leakiness)
        elif nl == 'relu':
            self.nl = LeakyReLU(negative_slope=0.)
        elif nl == 'tanh':
            self.nl = Tanh()
        else:
            raise ValueError("config does not support this nonlinearity.\nSupported nonlinearities are: "
                             "[ 'leaky relu', 'relu', 'tanh' ]")

        self.gen_model = None
        self.disc_model = None
        self._gradient_penalty = config.gradient_penalty
        self._optimizer = config.optimizer.casefold()
        self.opt_gen = self.opt_disc = None
        self._lr_sched = None
        self.sched_bool = False
        self.sched_stop_step = None
        self.scheduler_gen = self.scheduler_disc = None
        if config.lr_sched is not None:
            self._lr_sched = config.lr_sched.casefold()
            self.sched_bool = True
            self.sched_stop_step = 0
        self.valid_z = None
        self.curr_img_num = 0
        self.tot_num_epochs = None
        self.not_trained_yet = True
        self.ds_mean = self.ds_std = None
        self.data_config = None
        if self._model == 'ResNet GAN':
            self._init_resnet(config)

    # -- the non-progressive ResNet GAN (resnetgan/learner.py:98-120, :186-300) -------------------------
    def _init_resnet(self, config):
        from .architectures import (FMAP_D, FMAP_G, Discriminator32PixResnet, Discriminator64PixResnet,
                                    Generator32PixResnet, Generator64PixResnet)
        self.config = LearnerConfigCopy(config, self.__class__.__name__, NONREDEFINABLE_ATTRS,
                                        REDEFINABLE_FROM_LEARNER_ATTRS)
        self._is_data_configed = False
        self._update_data_config(raise_exception=False)
        self.batch_size = self.config.batch_size
        c = self.config
        if c.res_samples == 64:
            gen_cls, disc_cls, fmap_g, fmap_d = Generator64PixResnet, Discriminator64PixResnet, FMAP_G, FMAP_D
        elif c.res_samples == 32:
            gen_cls, disc_cls, fmap_g, fmap_d = Generator32PixResnet, Discriminator32PixResnet, FMAP_G * 2, FMAP_D * 2
        else:
            raise ValueError('GANLearner currently only supports 32 pixel and 64 pixel GAN architectures.\n'
                             'If a different generated sample resolution is desired, please use the\n'
                             'ProGAN or StyleGAN models featured in this package instead.')
        fmap_g = getattr(config, 'fmap_g', fmap_g)      # width override (tests / small runs)
        fmap_d = getattr(config, 'fmap_d', fmap_d)
        self.gen_model = gen_cls(len_latent=c.len_latent, fmap=fmap_g, upsampler=self.gen_model_upsampler,
                                 blur_type=c.blur_type, nl=self.nl, num_classes=self.num_classes_gen,
                                 equalized_lr=c.use_equalized_lr)
        self.disc_model = disc_cls(fmap=fmap_d, pooler=self.disc_model_downsampler, blur_type=c.blur_type,
                                   nl=self.nl, num_classes=self.num_classes_disc, equalized_lr=c.use_equalized_lr)
        self.gen_model.to(c.dev)
        self.disc_model.to(c.dev)
        from .. import rng
        rng.seed_from_config(c.random_seed)
        assert self.gen_model.res == self.disc_model.res
        self.latent_distribution = c.latent_distribution
        self.reducer = parallel.GradReducer()
        self.log_every = getattr(config, 'log_every', 50)
        self.last_losses = {}
        self._loss = config.loss.casefold()
        self._set_loss()
        self._make_arenas()
        self._set_optimizer()
        if parallel.rank() == 0:
            print('-------- Initialized Model Configuration --------')
            print(self.config)
            print('-------------------------------------------------')
            print('\n    Ready to train!\n')

    def _make_arenas(self):
        self._graph_gen = getattr(self, '_graph_gen', 0) + 1    # captured step graphs point into the old arenas (graphs.py)
        self.arena_g = ParamArena(self.gen_model.named_parameters(), self.config.dev)
        self.arena_d = ParamArena(self.disc_model.named_parameters(), self.config.dev)
        parallel.broadcast_params(self.arena_g.flat)
        parallel.broadcast_params(self.arena_d.flat)

    def _set_optimizer(self):
        """resnetgan/learner.py:884-908: Adam through configure_adam_for_gan; the others are not implemented
        upstream either."""
        if self._optimizer != 'adam':
            if self._optimizer in ('rmsprop', 'momentum', 'sgd'):
                raise NotImplementedError(f'{self._optimizer} optimizer not yet implemented.')
            raise ValueError("config does not support this optimizer.\nSupported Optimizers are: "
                             "[ 'adam', 'rmsprop', 'momentum', 'sgd' ]")
        c = self.config
        adam_gan = configure_adam_for_gan(lr_base=c.lr_base, betas=(c.beta1, c.beta2), eps=c.eps, wd=c.wd)
        self._graph_gen = getattr(self, '_graph_gen', 0) + 1
        self.opt_gen = adam_gan(params=list(self.gen_model.parameters()))
        self.opt_disc = adam_gan(params=list(self.disc_model.parameters()))

    def _set_scheduler(self):
        """resnetgan/learner.py:849-864."""
        if self._lr_sched == 'linear decay':
            self.scheduler_fn = lambda main_iter: 1. - (main_iter + self.sched_stop_step) * (1. / self.num_main_iters)
        elif self._lr_sched == 'custom':
            self.scheduler_fn = eval(self.config.lr_sched_custom)
        else:
            raise ValueError("config does not support this LR scheduler.\n"
                             "Currently supported LR Schedulers are: [ 'linear decay', 'custom' ]")
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            self.scheduler_gen = torch.optim.lr_scheduler.LambdaLR(self.opt_gen, self.scheduler_fn, last_epoch=-1)
            self.scheduler_disc = torch.optim.lr_scheduler.LambdaLR(self.opt_disc, self.scheduler_fn, last_epoch=-1)

    def set_requires_grad_disc(self, flag):
        for p in self.disc_model.parameters():
            p.requires_grad_(flag)

    @property
    def use_step_graph(self):
        """config.use_step_graph / GANLAB_STEP_GRAPH: 1 = replay eligible iterations as HIP graphs (graphs.GraphedStep: the
        progressive learners), 0 = never, unset = where the step is launch-bound (resolutions up to 256).  The ResNet GAN's
        iteration is bound by its kernels (measured: 158.9 ms eager, 159.0 ms replayed) and always steps eagerly."""
        import os
        v = os.environ.get('GANLAB_STEP_GRAPH', getattr(self.config, 'use_step_graph', None))
        if v in (None, '', 'auto'):
            res = getattr(self.gen_model, 'curr_res', None) or self.config.res_samples
            return res <= 256
        return str(v).lower() not in ('0', 'false', 'no')

    # -- the hot path: one generator iteration, one critic iteration ------------------------------------
    def g_step(self, zb=None):
        """resnetgan/learner.py:545-597 (critic parameters frozen by the caller)."""
        c = self.config
        self.arena_g.zero_grad()
        if zb is None:
            zb = gen_rand_latent_vars(num_samples=self.batch_size * c.gen_bs_mult, length=c.len_latent,
                                      distribution=self.latent_distribution, device=c.dev)
        out = self.disc_model(self.gen_model(zb))
        # :573-578 - the minimax generator loss here is -BCE(D(G(z)), 0), like backprop_utils
        loss = self.loss_func_gen(out)
        self.reducer.arm(self.arena_g)
        with ops.direct_param_grads(ops.direct_grads_enabled()):      # first-use gradients land in the arena directly
            loss.backward()
        self.reducer.allreduce(self.arena_g.gflat)
        self.opt_gen.step()
        return loss.detach()

    def _pair_critic_batches(self, xgenb, xb):
        """May the critic see the generated and the real batch as one?  Only when no critic layer couples samples (a
        BatchNorm would take its statistics over both) and the output is the plain score.  GANLAB_RESNET_PAIR=0: A/B."""
        if os.environ.get('GANLAB_RESNET_PAIR') == '0' or xgenb.numel() != xb.numel():
            return False
        key, ok = getattr(self, '_pairable', (None, False))
        if key != id(self.disc_model):
            ok = not any(isinstance(m, torch.nn.modules.batchnorm._BatchNorm) for m in self.disc_model.modules()) and \
                not hasattr(self.disc_model, 'linear_aux')
            self._pairable = (id(self.disc_model), ok)
        return ok

    def d_step(self, xb, zb=None, eps_interp=None):
        """resnetgan/learner.py:606-672: generator frozen but in train mode (its BatchNorm running
        statistics keep moving, :621-622); no drift term on this path."""
        c = self.config
        self.arena_d.zero_grad()
        if zb is None:
            zb = gen_rand_latent_vars(num_samples=self.batch_size, length=c.len_latent,
                                      distribution=self.latent_distribution, device=c.dev)
        with torch.no_grad():
            xgenb = self.gen_model(zb)
        if self._pair_critic_batches(xgenb, xb):
            # one critic pass over [generated; real]: every critic layer is per-sample (LayerNorm), so the outputs are the
            # two separate passes' (resnetgan/learner.py:640-651) and each parameter gets ONE gradient contribution from
            # the pair instead of two - half the launches of the first-order critic work at this launch-bound size
            n = xgenb.shape[0]
            out = self.disc_model(torch.cat((xgenb, xb.reshape(xgenb.shape))))
            loss = self.loss_func_disc(out[:n], out[n:])
        else:
            loss = self.loss_func_disc(self.disc_model(xgenb), self.disc_model(xb))
        if self.gradient_penalty is not None:
            loss = loss + self.calc_gp(xgenb, xb, eps_interp=eps_interp)
        self.reducer.arm(self.arena_d)
        with ops.direct_param_grads(ops.direct_grads_enabled()):
            loss.backward()
        self.reducer.allreduce(self.arena_d.gflat)
        self.opt_disc.step()
        return loss.detach()

    def train(self, train_dl, valid_dl=None, z_valid_dl=None, num_main_iters=None, num_gen_iters=None,
              num_disc_iters=None):
        """GAN training, generator first (resnetgan/learner.py:463-700); re-entrant like the reference."""
        c = self.config
        num_main_iters = c.num_main_iters if num_main_iters is None else num_main_iters
        num_gen_iters = c.num_gen_iters if num_gen_iters is None else num_gen_iters
        num_disc_iters = c.num_disc_iters if num_disc_iters is None else num_disc_iters
        self.num_main_iters = num_main_iters
        self.dataset_sz = len(train_dl.dataset)
        self._update_data_config(raise_exception=False)
        self.gen_model.to(c.dev).train()
        self.disc_model.to(c.dev).train()
        if self.sched_bool:
            if not self.pretrained_model:
                self.sched_stop_step = 0
            self._set_scheduler()
        if self.not_trained_yet or self.pretrained_model:
            self.train_dataiter = iter(train_dl)
        if parallel.rank() == 0:
            print('STARTING FROM ITERATION 0:\n' if self.not_trained_yet else 'CONTINUING FROM WHERE YOU LEFT OFF:\n')
        if self.tot_num_epochs is None:
            per_epoch = max(self.dataset_sz // self.batch_size * self.batch_size, 1)
            self.tot_num_epochs = num_main_iters * self.batch_size * num_disc_iters // per_epoch + 1
        loss_d = loss_g = None
        try:
            for itr in range(num_main_iters):
                # ---------------------------- TRAIN GENERATOR ----------------------------
                self.set_requires_grad_disc(False)
                for _ in range(num_gen_iters):
                    loss_g = self.g_step()
                # -------------------------- TRAIN DISCRIMINATOR --------------------------
                self.set_requires_grad_disc(True)
                for _ in range(num_disc_iters):
                    batch = next(self.train_dataiter, None)
                    if batch is None:
                        self.curr_epoch_num += 1
                        self.train_dataiter = iter(train_dl)
                        batch = next(self.train_dataiter)
                    xb = batch[0].to(c.dev, non_blocking=True).float()
                    loss_d = self.d_step(xb)
                    self.curr_dataset_batch_num += 1
                    self.curr_img_num += self.batch_size
                if self.sched_bool:
                    with warnings.catch_warnings():
                        warnings.simplefilter('ignore')
                        self.scheduler_gen.step()
                        self.scheduler_disc.step()
                self.not_trained_yet = False
                if self.log_every and (itr % self.log_every == 0 or itr == num_main_iters - 1):
                    self.last_losses = dict(itr=itr, loss_d=float(loss_d) if loss_d is not None else None,
                                            loss_g=float(loss_g) if loss_g is not None else None,
                                            res=c.res_samples, batch=self.batch_size)
                    if parallel.rank() == 0:
                        print(('%9s' * 5) % (f'{self.curr_epoch_num}/{self.tot_num_epochs}',
                                             f'{c.res_samples}X{c.res_samples}',
                                             '%.4g' % (self.last_losses['loss_d'] or 0.),
                                             '%.4g' % (self.last_losses['loss_g'] or 0.), itr))
                if (itr + 1) % c.num_iters_save_model == 0:
                    self.save_model(c.save_model_dir / (self.model.casefold().replace(' ', '') + '_model.tar'))
        except KeyboardInterrupt:
            # resnetgan/learner.py: Ctrl-C saves the latest checkpoint before the run ends
            self.set_requires_grad_disc(True)
            self.reducer.abandon()          # no collective in the interrupt path (ranks are at different points)
            if not self.not_trained_yet:
                self.save_model(c.save_model_dir / (self.model.casefold().replace(' ', '') + '_model.tar'), sync=False)
                if parallel.rank() == 0:
                    print(f'\nTraining interrupted. Saved latest checkpoint into "{c.save_model_dir}/".\n')
            raise

    def save_model(self, save_path, sync=True):
        """Checkpoint as plain data (key names follow resnetgan/learner.py:1076-1140).  ``sync=False``: no barrier
        behind rank 0's write (the interrupt path)."""
        if self.not_trained_yet:
            raise Exception('Please train your model for atleast 1 iteration before saving.')
        from .. import checkpoint as ckpt
        tcpu = lambda v: None if v is None else v.detach().cpu()  # noqa: E731
        sched_steps = max(self.scheduler_gen._step_count - 1, 0) if (self.sched_bool and self.scheduler_gen) else 0
        ck = {
            'config': {k: v for k, v in vars(self.config).items() if not k.startswith('_') and
                       isinstance(v, (int, float, str, bool, dict, list, tuple, type(None)))},
            'gen_model_state_dict': {k: v.detach().cpu() for k, v in self.gen_model.state_dict().items()},
            'disc_model_state_dict': {k: v.detach().cpu() for k, v in self.disc_model.state_dict().items()},
            'opt_gen_state_dict': self.opt_gen.export_moments(self.gen_model.named_parameters()),
            'opt_disc_state_dict': self.opt_disc.export_moments(self.disc_model.named_parameters()),
            'sched_stop_step': (self.sched_stop_step or 0) + sched_steps if self.sched_bool else self.sched_stop_step,
            'lr_sched': self.lr_sched, 'optimizer': self.optimizer,
            'loss': self.loss, 'gradient_penalty': self.gradient_penalty, 'batch_size': self.batch_size,
            'curr_dataset_batch_num': self.curr_dataset_batch_num, 'curr_epoch_num': self.curr_epoch_num,
            'tot_num_epochs': self.tot_num_epochs, 'curr_img_num': self.curr_img_num,
            'not_trained_yet': self.not_trained_yet,
            'ds_mean': tcpu(self.ds_mean), 'ds_std': tcpu(self.ds_std), 'valid_z': tcpu(self.valid_z),
        }
        if parallel.rank() == 0:            # replicas are identical: one writer, atomically; everyone waits for the file
            ckpt.save_atomic(ck, save_path)
        if sync:
            parallel.barrier()

    def load_model(self, load_path, dev_of_saved_model='cpu'):
        from .. import checkpoint as ckpt
        ck = ckpt.load_checkpoint(load_path, dev_of_saved_model)
        self.gen_model.load_state_dict(ck['gen_model_state_dict'])
        self.disc_model.load_state_dict(ck['disc_model_state_dict'])
        self.gen_model.to(self.config.dev)
        self.disc_model.to(self.config.dev)
        self._make_arenas()
        self._set_optimizer()
        self.opt_gen.import_moments(self.gen_model.named_parameters(), ck['opt_gen_state_dict'])
        self.opt_disc.import_moments(self.disc_model.named_parameters(), ck['opt_disc_state_dict'])
        for k in ('sched_stop_step', 'batch_size', 'curr_dataset_batch_num', 'curr_epoch_num', 'tot_num_epochs',
                  'curr_img_num', 'not_trained_yet'):
            setattr(self, k, ck[k])
        if ck.get('ds_mean') is not None and ck.get('ds_std') is not None:
            self.ds_mean, self.ds_std = ck['ds_mean'].float().cpu(), ck['ds_std'].float().cpu()
        if ck.get('valid_z') is not None:
            self.valid_z = ck['valid_z'].to(self.config.dev)
        self.pretrained_model = True

    # -- gradient penalty (resnetgan/learner.py:780-827) ------------------------------------------------
    def calc_gp(self, gen_data, real_data, eps_interp=None):
        """Method that takes care of all gradient regularizers (double backward through HIP kernels)."""
        return bp.calc_gp(self.disc_model, self.gradient_penalty, gen_data, real_data, lda=self.config.lda,
                          gamma=self.config.gamma, eps_interp=eps_interp)

    # -- redefinable-from-learner properties (:831-946) -------------------------------------------------
    @property
    def lr_sched(self):
        return self._lr_sched

    @lr_sched.setter
    def lr_sched(self, new_lr_sched):
        self._lr_sched = None
        self.sched_bool = False
        self.scheduler_gen = self.scheduler_disc = None
        if new_lr_sched is not None:
            self._lr_sched = new_lr_sched.casefold()
            self.sched_bool = True
            if not self.pretrained_model:
                self.sched_stop_step = 0

    @property
    def optimizer(self):
        return self._optimizer

    @optimizer.setter
    def optimizer(self, new_optimizer):
        self._optimizer = new_optimizer.casefold()
        self._set_optimizer()

    @property
    def gradient_penalty(self):
        return self._gradient_penalty.casefold() if self._gradient_penalty is not None else None

    @gradient_penalty.setter
    def gradient_penalty(self, new_gradient_penalty):
        self._gradient_penalty = new_gradient_penalty.casefold() if new_gradient_penalty is not None else None

    @property
    def loss(self):
        return self._loss

    @loss.setter
    def loss(self, new_loss):
        self._loss = new_loss.casefold()
        self._set_loss()

    def _set_loss(self):
        if self._loss not in ('wgan', 'nonsaturating', 'minimax',):
            raise ValueError("config does not support this loss.\nCurrently supported Loss Functions are: "
                             "[ 'wgan', 'nonsaturating', 'minimax' ]")
        # the BCE targets are constants folded into the loss kernels (no cached ones/zeros tensors,
        # cf. resnetgan/learner.py:937-938)
        self.loss_func_gen = lambda outb: bp.loss_gen(self._loss, outb)
        self.loss_func_disc = lambda outb, yb: bp.loss_disc(self._loss, outb, yb)

    @property
    def model(self):
        return self._model

    @model.setter
    def model(self, new_model):
        raise AttributeError(
            f"{self.__class__.__name__}().model attribute cannot be changed once {self.__class__.__name__} is "
            f"instantiated.\nInstead, please run 'python config.py {new_model}' and then instantiate a new "
            f"{self.__class__.__name__}.")

    # -- data config (resnetgan/learner.py:1051-1072) ----------------------------------------------------
    def _update_data_config(self, raise_exception=True):
        dc = get_current_configuration('data_config', raise_exception=raise_exception)
        if dc is not None:
            self.data_config = dc
            if not self.pretrained_model and getattr(dc, 'ds_mean', None) is not None:
                self.ds_mean = torch.FloatTensor(dc.ds_mean).unsqueeze(dim=1).unsqueeze(dim=2)
                self.ds_std = torch.FloatTensor(dc.ds_std).unsqueeze(dim=1).unsqueeze(dim=2)
        self._is_data_configed = self.data_config is not None
