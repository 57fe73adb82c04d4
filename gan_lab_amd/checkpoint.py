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


_FOREIGN = {('_int', 'LearnerConfigCopy'): _ConfigShell, ('indexed', 'IndexedOrderedDict'): IndexedOrderedDict}


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
