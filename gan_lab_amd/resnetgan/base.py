"""Base architecture class for the non-progressive GANs (drop-in for gan_lab/resnetgan/base.py:15-37)."""
from abc import ABC, abstractmethod

from torch import nn


class GAN(nn.Module, ABC):
    def __init__(self, res):
        super().__init__()
        self._res = res

    def most_parameters(self, recurse=True, excluded_params: list = []):
        """nn.Module.parameters() with the option to exclude parameters by name."""
        for name, params in self.named_parameters(recurse=recurse):
            if name not in excluded_params:
                yield params

    @property
    def res(self):
        return self._res

    @res.setter
    def res(self, new_res):
        raise AttributeError(f'GAN().res cannot be changed, as {self.__class__.__name__} only permits one '
                             f'resolution: {self._res}.')

    @abstractmethod
    def forward(self, x):
        raise NotImplementedError('Can only call `forward` on valid subclasses.')
