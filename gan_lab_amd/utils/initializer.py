"""Equalised-LR / He initialisation constants (reference: gan_lab/utils/initializer.py:27-80).

Only the std / runtime-scale computation is needed by the hot path; it is what gets folded into
the packed weights of the HIP conv kernels."""
import math


class Initializer(object):
    def __init__(self, init, init_type='default', gain_sq_base=2., equalized_lr=False):
        self.init = init.casefold()
        self.init_type = init_type.casefold()
        self.gain_sq_base = gain_sq_base
        self.equalized_lr = equalized_lr

    def get_init_bound_layer(self, tensor, distribution_type, stride=1):
        distribution_type = distribution_type.casefold()
        if distribution_type not in ('uniform', 'normal',):
            raise ValueError('Only uniform and normal distributions are supported.')
        fan_in, fan_out = self._calculate_fan_in_fan_out(tensor, stride)
        std = self._calculate_init_weight_std(fan_in, fan_out)
        return math.sqrt(3) * std if distribution_type == 'uniform' else std

    def _calculate_fan_in_fan_out(self, tensor, stride=1):
        if tensor.dim() < 2:
            raise ValueError('Fan in and fan out cannot be computed for tensor with fewer than 2 dimensions.')
        progressive = self.init_type in ('progan', 'stylegan',)   # fan-in only (initializer.py:35-37)
        if tensor.dim() == 2:
            fan_in = tensor.size(1)
            fan_out = None if progressive else tensor.size(0)
        else:
            rf = tensor[0][0].numel()
            fan_in = tensor.size(1) * rf
            fan_out = None if progressive else tensor.size(0) * rf / stride ** 2
        return fan_in, fan_out

    def _calculate_init_weight_std(self, fan_in=None, fan_out=None):
        gain_sq = self.gain_sq_base / 2.
        if fan_out is not None and fan_in is not None:
            fan = fan_in + fan_out
            gain_sq *= 2
        elif fan_in is not None:
            fan = fan_in
        else:
            fan = fan_out
        if self.init == 'he':
            gain_sq = 2. * gain_sq
        return math.sqrt(gain_sq / fan)
