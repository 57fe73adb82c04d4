"""Base learner (drop-in surface of gan_lab/resnetgan/learner.py:87-301, :780-946) for the HIP path:
supervision flags, resampler / nonlinearity selection, loss + gradient-penalty + optimiser + LR
scheduler plumbing shared by ProGANLearner / StyleGANLearner.  The non-progressive ResNet GAN
architectures themselves (BASELINE config #5) are a later hot-path row and raise here."""
import torch

from .. import _lib
from .._int import FMAP_SAMPLES, LearnerConfigCopy, get_current_configuration  # noqa: F401
from ..utils import backprop_utils as bp
from ..utils.custom_layers import AvgPool2x, LeakyReLU, Upsample2x


class GANLearner(object):
    def __init__(self, config):
        super().__init__()
        self._model = config.model
        self.pretrained_model = False
        if self._model == 'ResNet GAN':
            raise NotImplementedError('the ResNet GAN path (BASELINE config #5) has no HIP kernels yet '
                                      '(BatchNorm / LayerNorm residual blocks: SURVEY.md §8a row A19)')
        dev = torch.device(config.dev)
        if dev.type != 'cuda':
            raise RuntimeError(f"gan_lab_amd runs on the MI355X only (config.dev={config.dev!r}); there is no CPU "
                               f"path - use the reference or the test oracle for CPU runs")
        _lib.lib()  # fail now, loudly, if the kernel library is missing

        self.curr_dataset_batch_num = 0
        self.curr_epoch_num = 1
        # supervised / unsupervised selection (resnetgan/learner.py:122-138)
        self.num_classes = 0
        self.cond_gen = self.cond_disc = self.ac = False
        self.num_classes_gen = self.num_classes_disc = 0
        if config.use_auxiliary_classifier or config.class_condition:
            raise NotImplementedError('class conditioning / auxiliary classifier: SURVEY.md §8f item 4 (next)')
        if not (config.res_samples <= config.res_dataset):
            raise ValueError(f'Resolution of generated images (config.res_samples = {config.res_samples}) must be '
                             f'less than\nor equal to resolution of dataset (config.res_dataset = '
                             f'{config.res_dataset}) at all times.\nPlease set config.res_samples <= '
                             f'config.res_dataset.')
        # resamplers (:147-176): only the hot-path choices have kernels
        if config.model_upsample_type.casefold() != 'nearest':
            raise ValueError("config does not support this model_upsample_type on the HIP path.\n"
                             "Supported Upsampling Types are: [ 'nearest' ]")
        self.gen_model_upsampler = Upsample2x()
        if config.model_downsample_type.casefold() not in ('average', 'box',):
            raise ValueError("config does not support this model_downsample_type on the HIP path.\n"
                             "Supported Downsampling Types are: [ 'average', 'box' ]")
        self.disc_model_downsampler = AvgPool2x()
        # nonlinearity (:178-184)
        nl = config.nonlinearity.casefold()
        if nl == 'leaky relu':
            self.nl = LeakyReLU(negative_slope=config.leakiness)
        elif nl == 'relu':
            self.nl = LeakyReLU(negative_slope=0.)
        else:
            raise ValueError("config does not support this nonlinearity on the HIP path: [ 'leaky relu', 'relu' ]")

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
