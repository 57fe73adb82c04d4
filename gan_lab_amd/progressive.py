"""Progressive-growing state shared by a generator/discriminator pair.

Reference behaviour (gan_lab/{progan,stylegan}/base.py:21-173): ``alpha``, ``curr_res``,
``scale_stage``, ``fmap``, ``fmap_prev``, ``fade_in_phase`` ... are CLASS-level attributes of the
family base class, so ``gen_model.alpha = x`` moves the discriminator too and both nets grow in
lock-step.  Here each family (``ProGAN`` / ``StyleGAN``) owns ONE mutable ``_FamilyState`` record
that all of its subclasses' instances read and write through properties - the observable behaviour
is the same, without re-creating classes at run time.
"""
import types

import numpy as np
from torch import nn

FMAP_BASE = 8192
FMAP_MAX = 512


def _fmap(scale_stage):
    return min(int(FMAP_BASE / (2 ** scale_stage)), FMAP_MAX)


class _FamilyState(object):
    def __init__(self):
        self.final_res = None
        self.reset()

    def reset(self):
        self.alpha = 1
        self.alpha_tol = 1.e-8
        self.prev_res = None
        self.curr_res = 4
        self.scale_stage = int(np.log2(self.curr_res)) - 1
        self.fmap_prev = None
        self.fmap = _fmap(self.scale_stage)
        self.scale_inc_metadata_updated = False
        self.fade_in_phase = False

    def as_dict(self):
        return dict(self.__dict__)


def _shared(name):
    def get(self):
        return getattr(self._state, name)

    def set_(self, value):
        setattr(self._state, name, value)

    return property(get, set_)


class ProgressiveBase(nn.Module):
    """Common machinery; concrete families bind ``_state``."""
    _state = None

    @classmethod
    def reset_state(cls):
        """Call this to start a new G/D pair from 4x4 (base.py:40-53)."""
        cls._state.reset()

    def __init__(self, final_res):
        super().__init__()
        self.final_res = final_res
        assert self.curr_res <= self.final_res

    @property
    def cls_base(self):
        # reference code compares `gen_model.cls_base.__dict__ == disc_model.cls_base.__dict__`
        return types.SimpleNamespace(**self._state.as_dict())

    def increase_scale(self):
        """Metadata update of a growth step (base.py:62-74)."""
        self.prev_res = self.curr_res
        self.curr_res = int(2 ** (int(np.log2(self.curr_res)) + 1))
        self.scale_stage = int(np.log2(self.curr_res)) - 1
        self.fmap_prev = self.fmap
        self.fmap = self.get_fmap(scale_stage=self.scale_stage)
        self.scale_inc_metadata_updated = True
        self.fade_in_phase = True

    def get_fmap(self, scale_stage):
        return _fmap(scale_stage)

    def most_parameters(self, recurse=True, excluded_params: list = []):
        """``parameters()`` minus the named ones (base.py:79-83)."""
        for name, params in self.named_parameters(recurse=recurse):
            if name not in excluded_params:
                yield params

    fade_in_phase = _shared('fade_in_phase')
    scale_inc_metadata_updated = _shared('scale_inc_metadata_updated')
    fmap = _shared('fmap')
    fmap_prev = _shared('fmap_prev')
    scale_stage = _shared('scale_stage')
    curr_res = _shared('curr_res')
    final_res = _shared('final_res')
    prev_res = _shared('prev_res')
    alpha_tol = _shared('alpha_tol')

    @property
    def alpha(self):
        return self._state.alpha

    @alpha.setter
    def alpha(self, new_alpha):
        # base.py:161-170: clamp check, snap to 1 within alpha_tol and leave the fade-in phase
        if not (0. <= new_alpha < 1. + self.alpha_tol):
            raise ValueError('Input alpha parameter must be in the range [0,1].')
        if 1. - self.alpha_tol < new_alpha < 1. + self.alpha_tol:
            self.fade_in_phase = False
            self._state.alpha = 1
        else:
            self._state.alpha = new_alpha


class ProGAN(ProgressiveBase):
    _state = _FamilyState()


class StyleGAN(ProgressiveBase):
    _state = _FamilyState()
