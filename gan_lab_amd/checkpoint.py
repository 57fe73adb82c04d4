"""Checkpoint wire-format compatibility with the reference.

``ProGANLearner.save_model`` of the reference (gan_lab/progan/learner.py:1238-1298; StyleGAN adds nothing
structural, stylegan/learner.py:452-501) ``torch.save``s a dict that holds, besides tensors and plain values,
a few *objects*: ``config`` (an ``_int.LearnerConfigCopy``), ``lagged_params`` (an ``indexed.
IndexedOrderedDict``), and torch modules (``nl``, the resamplers).  ``load_checkpoint`` reads such a file
without the reference on the path: the two foreign classes are mapped onto local stand-ins while unpickling,
everything else is torch / stdlib.  Helper functions translate ``torch.optim.Adam.state_dict()`` into the
flat-arena moments of ``optim.FusedAdam`` and restore the phase machine.  Pure host code (no kernels)."""
import pickle
import types
from collections import OrderedDict

import torch
from torch import nn

from ._int import LearnerConfigCopy


class IndexedOrderedDict(OrderedDict):
    """Stand-in for ``indexed.IndexedOrderedDict`` (list-returning ``keys()`` / ``values()``)."""

    def values(self):
        return list(super().values())

    def keys(self):
        return list(super().keys())


class _ConfigShell(LearnerConfigCopy):
    """``_int.LearnerConfigCopy`` as unpickled: state goes straight into ``__dict__`` (no ctor, no guards)."""

    def __init__(self):  # noqa: D401 - never called by pickle
        pass

    def __setattr__(self, name, value):
        object.__setattr__(self, name, value)


class _PoolShell(nn.Module):
    """``utils.custom_layers.NearestPool2d`` / ``BilinearPool2d`` as unpickled from a reference-written file: the
    metadata entry only names the pooler (``custom_layers.own_resampler`` maps it by class name); the learner rebuilds its
    resamplers from the config."""


class NearestPool2d(_PoolShell):
    pass


class BilinearPool2d(_PoolShell):
    pass


_FOREIGN = {('_int', 'LearnerConfigCopy'): _ConfigShell, ('indexed', 'IndexedOrderedDict'): IndexedOrderedDict,
            ('utils.custom_layers', 'NearestPool2d'): NearestPool2d,
            ('utils.custom_layers', 'BilinearPool2d'): BilinearPool2d}


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        hit = _FOREIGN.get((module, name))
        return hit if hit is not None else super().find_class(module, name)


_pickle_module = types.ModuleType('gan_lab_amd._ckpt_pickle')
_pickle_module.Unpickler = _Unpickler
_pickle_module.load = lambda f, **kw: _Unpickler(f, **kw).load()
_pickle_module.__dict__.update({k: getattr(pickle, k) for k in ('HIGHEST_PROTOCOL', 'DEFAULT_PROTOCOL', 'Pickler',
                                                                 'dump', 'dumps', 'loads', 'PickleError',
                                                                 'UnpicklingError')})


def load_checkpoint(path, map_location='cpu'):
    """``torch.load`` that also understands reference-written files (``weights_only=False`` is inherent: the file
    holds pickled objects; only load checkpoints you trust, exactly as with the reference)."""
    return torch.load(str(path), map_location=map_location, pickle_module=_pickle_module, weights_only=False)


def is_reference_format(ck):
    return isinstance(ck, dict) and 'gen_model_metadata' in ck and not isinstance(ck.get('config'), dict)


def config_dict(ck):
    c = ck['config']
    return dict(c) if isinstance(c, dict) else {k: v for k, v in vars(c).items() if not k.startswith('_')}


def moments_from_torch_adam(opt_state_dict, ordered_names):
    """``torch.optim.Adam.state_dict()`` -> the plain ``{'step', 'exp_avg', 'exp_avg_sq'}`` record of
    ``FusedAdam.import_moments``.  The reference builds its optimisers from ``most_parameters(...)``
    (progan/learner.py:1064-1095), so state index ``i`` is the ``i``-th entry of ``ordered_names``."""
    idx = list(opt_state_dict['param_groups'][0]['params'])
    if len(idx) != len(ordered_names):
        raise ValueError(f'optimizer state holds {len(idx)} parameters, the network exposes {len(ordered_names)}')
    out = dict(step=0, exp_avg={}, exp_avg_sq={})
    for i, name in zip(idx, ordered_names):
        st = opt_state_dict['state'].get(i)
        if st is None:
            continue
        out['step'] = max(out['step'], int(st['step']))
        out['exp_avg'][name] = st['exp_avg'].detach().float().cpu()
        out['exp_avg_sq'][name] = st['exp_avg_sq'].detach().float().cpu()
    return out


# ---------------------------------------------------------------------------------------------------------------
# writing a file the REFERENCE's load_model accepts (gan_lab/progan/learner.py:1257-1360, stylegan/learner.py:452-640)
# ---------------------------------------------------------------------------------------------------------------
class _RefConfig(object):
    """Pickles as ``_int.LearnerConfigCopy`` (the reference unpickles it without calling its constructor: the state
    dict goes straight into ``__dict__``)."""

    def __init__(self, state):
        object.__setattr__(self, '__dict__', dict(state))


class _RefIndexedOrderedDict(OrderedDict):
    """Pickles as ``indexed.IndexedOrderedDict`` in the ``(cls, (items,))`` form both the real package's class and a
    plain OrderedDict subclass rebuild from."""

    def __reduce__(self):
        return (self.__class__, ([[k, v] for k, v in self.items()],))


class _RefNearestPool2d(nn.Module):
    """Pickles as ``utils.custom_layers.NearestPool2d`` (custom_layers.py:59-65)."""


class _RefBilinearPool2d(nn.Module):
    """Pickles as ``utils.custom_layers.BilinearPool2d`` (custom_layers.py:67-75)."""

    def __init__(self, align_corners):
        super().__init__()
        self.align_corners = align_corners


for _cls, _name in ((_RefNearestPool2d, 'NearestPool2d'), (_RefBilinearPool2d, 'BilinearPool2d')):
    _cls.__module__, _cls.__qualname__, _cls.__name__ = 'utils.custom_layers', _name, _name
_RefConfig.__module__, _RefConfig.__qualname__, _RefConfig.__name__ = '_int', 'LearnerConfigCopy', 'LearnerConfigCopy'
_RefIndexedOrderedDict.__module__ = 'indexed'
_RefIndexedOrderedDict.__qualname__ = _RefIndexedOrderedDict.__name__ = 'IndexedOrderedDict'


class _foreign_modules(object):
    """While pickling, ``_int`` / ``indexed`` must resolve to the stand-ins above (pickle verifies that the global it
    writes is importable); modules already imported under those names (e.g. the real ones) are put back afterwards."""

    def __enter__(self):
        import sys
        self._saved = {k: sys.modules.get(k) for k in ('_int', 'indexed', 'utils', 'utils.custom_layers')}
        m1, m2 = types.ModuleType('_int'), types.ModuleType('indexed')
        m3, m4 = types.ModuleType('utils'), types.ModuleType('utils.custom_layers')
        m1.LearnerConfigCopy, m2.IndexedOrderedDict = _RefConfig, _RefIndexedOrderedDict
        m4.NearestPool2d, m4.BilinearPool2d, m3.custom_layers = _RefNearestPool2d, _RefBilinearPool2d, m4
        sys.modules['_int'], sys.modules['indexed'] = m1, m2
        sys.modules['utils'], sys.modules['utils.custom_layers'] = m3, m4

    def __exit__(self, *exc):
        import sys
        for k, v in self._saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        return False


def _ref_upsampler(m):
    """This package's upsampler module -> what the reference pickles (resnetgan/learner.py:147-158)."""
    if type(m).__name__ == 'BilinearUpsample2x':
        return nn.Upsample(scale_factor=2, mode='bilinear', align_corners=m.align_corners)
    return nn.Upsample(scale_factor=2, mode='nearest')


def _ref_downsampler(m):
    """This package's pooler module -> what the reference pickles (resnetgan/learner.py:160-173)."""
    name = type(m).__name__
    if name == 'NearestPool2x':
        return _RefNearestPool2d()
    if name == 'BilinearPool2x':
        return _RefBilinearPool2d(m.align_corners)
    return nn.AvgPool2d(kernel_size=2, stride=2)


def torch_adam_state_dict(fused_adam, named_params, ordered_names):
    """``FusedAdam`` flat-arena moments -> ``torch.optim.Adam.state_dict()`` (state index i = the i-th of
    ``ordered_names``, the reference's ``most_parameters(...)`` order, progan/learner.py:1064-1095)."""
    mom = fused_adam.export_moments(named_params)
    group = dict(fused_adam.param_groups[0])
    template = torch.optim.Adam([torch.zeros(1, requires_grad=True)], lr=group['lr'], betas=group['betas'],
                                eps=group['eps'], weight_decay=group['weight_decay']).state_dict()['param_groups'][0]
    pg = dict(template)
    pg.update(lr=group['lr'], params=list(range(len(ordered_names))))
    if 'initial_lr' in group:
        pg['initial_lr'] = group['initial_lr']
    state = {}
    if mom['step'] > 0:
        for i, name in enumerate(ordered_names):
            if name in mom['exp_avg']:
                state[i] = {'step': torch.tensor(float(mom['step'])), 'exp_avg': mom['exp_avg'][name].clone(),
                            'exp_avg_sq': mom['exp_avg_sq'][name].clone()}
    return {'state': state, 'param_groups': [pg]}


def reference_checkpoint_dict(learner, g_names, d_names, extra=None):
    """The dict ``ProGANLearner.save_model`` of the reference writes (progan/learner.py:1257-1298; ``extra``: the
    StyleGAN additions, stylegan/learner.py:455-464), built from a product learner.  Tensors are moved to the CPU;
    the module-valued fields (``nl``, the resamplers) are the torch modules the reference itself constructs
    (resnetgan/learner.py:147-184)."""
    from torch import nn
    c = learner.config
    cpu = lambda sd: OrderedDict((k, v.detach().cpu().clone()) for k, v in sd.items())  # noqa: E731
    cfg_state = {k: v for k, v in vars(c).items()}
    nl = {'leaky relu': lambda: nn.LeakyReLU(negative_slope=c.leakiness), 'tanh': nn.Tanh}.get(c.nonlinearity.casefold(),
                                                                                             nn.ReLU)()
    lagged = learner.materialize_lagged_generator() if c.use_ewma_gen else None
    lagged_params = None
    if c.use_ewma_gen and learner.lagged_params is not None:
        lagged_params = _RefIndexedOrderedDict((k, v.detach().cpu().clone()) for k, v in learner.lagged_params.items())
    n_grid = c.img_grid_sz ** 2
    valid_z = learner.valid_z if learner.valid_z is not None else torch.zeros(n_grid, c.len_latent)
    sched = learner.sched_bool and learner.scheduler_gen is not None
    ck = {
        'config': _RefConfig(cfg_state),
        'curr_res': learner.gen_model.curr_res,
        'alpha': learner.gen_model.alpha,
        'gen_model_metadata': {'gen_model_upsampler': _ref_upsampler(learner.gen_model_upsampler),
                               'num_classes_gen': learner.num_classes_gen},
        'gen_model_state_dict': cpu(learner.gen_model.state_dict()),
        'gen_model_lagged_state_dict': cpu(lagged.state_dict()) if lagged is not None else None,
        'disc_model_metadata': {'disc_model_downsampler': _ref_downsampler(learner.disc_model_downsampler),
                                'num_classes_disc': learner.num_classes_disc},
        'disc_model_state_dict': cpu(learner.disc_model.state_dict()),
        'nl': nl,
        'sched_stop_step': learner.sched_stop_step,
        'lr_sched': learner.lr_sched,
        'scheduler_gen_state_dict': learner.scheduler_gen.state_dict() if sched else None,
        'scheduler_disc_state_dict': learner.scheduler_disc.state_dict() if sched else None,
        'optimizer': learner.optimizer,
        'opt_gen_state_dict': torch_adam_state_dict(learner.opt_gen, list(learner.gen_model.named_parameters()), g_names),
        'opt_disc_state_dict': torch_adam_state_dict(learner.opt_disc, list(learner.disc_model.named_parameters()),
                                                     d_names),
        'loss': learner.loss,
        'gradient_penalty': learner.gradient_penalty,
        'batch_size': learner.batch_size,
        'curr_dataset_batch_num': learner.curr_dataset_batch_num,
        'curr_epoch_num': learner.curr_epoch_num,
        'tot_num_epochs': learner.tot_num_epochs,
        'dataset_sz': getattr(learner, 'dataset_sz', None),
        'ac': learner.ac, 'cond_gen': learner.cond_gen, 'cond_disc': learner.cond_disc,
        'valid_z': valid_z.detach().to('cpu'),
        'valid_label': learner.valid_label,
        'grid_inputs_constructed': learner.grid_inputs_constructed,
        'rand_idxs': learner.rand_idxs,
        'gen_metrics_num': learner.gen_metrics_num,
        'disc_metrics_num': learner.disc_metrics_num,
        'curr_img_num': learner.curr_img_num,
        'nimg_transition_lst': list(learner.sched.nimg_transition_lst) if learner.sched is not None else
        [learner.config.nimg_transition],
        'not_trained_yet': learner.not_trained_yet,
        'ds_mean': learner.ds_mean, 'ds_std': learner.ds_std,
        'latent_distribution': learner.latent_distribution,
        'curr_phase_num': learner.curr_phase_num,
        'lagged_params': lagged_params,
        'progressively_grow': learner.progressively_grow,
    }
    if extra:
        ck.update(extra)
    return ck


def save_atomic(obj, path, foreign=False):
    """``torch.save`` to ``path`` through a temporary file in the same directory + ``os.replace`` (a reader never
    sees a torn file)."""
    import os
    path = str(path)
    os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
    tmp = f'{path}.tmp{os.getpid()}'
    try:
        if foreign:
            with _foreign_modules():
                torch.save(obj, tmp)
        else:
            torch.save(obj, tmp)
        os.replace(tmp, path)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


ARCH_FIELDS = ('model', 'res_samples', 'len_latent', 'blur_type', 'nonlinearity', 'use_equalized_lr', 'normalize_z',
               'use_pixelnorm', 'mbstd_group_size', 'num_classes', 'len_dlatent', 'mapping_num_fcs', 'use_noise',
               'use_instancenorm')


def check_architecture(ck_cfg, my_cfg):
    """The receiving learner was built from its own config: the fields that shape the networks must agree."""
    bad = [(k, ck_cfg[k], getattr(my_cfg, k)) for k in ARCH_FIELDS
           if k in ck_cfg and hasattr(my_cfg, k) and ck_cfg[k] != getattr(my_cfg, k)]
    if bad:
        raise ValueError('checkpoint / learner architecture mismatch: ' +
                         ', '.join(f'{k}: checkpoint {a!r} vs config {b!r}' for k, a, b in bad))
