"""gan_lab_amd - MI355X-native G+D training-step hot path for sidward14/gan-lab (StyleGAN / ProGAN).

Layout (mirrors the reference package `gan_lab/` for the path it replaces):
  csrc/                 hand-written gfx950 HIP kernels + C-ABI (include/ganlab_hip.h)
  _lib.py, ops.py       ctypes binding and double-differentiable autograd ops over the C-ABI
  utils/                custom_layers / initializer / backprop_utils / latent_utils
  stylegan/, progan/    architectures (+ learners) with the reference's module tree and API
  optim.py, parallel.py fused Adam/EWMA over flat parameter arenas; RCCL data parallelism

Importing the package never touches the GPU and never builds anything; every op raises if the HIP
library is missing (no CPU / PyTorch fallback).
"""
from ._int import get_current_configuration  # noqa: F401

__version__ = '0.1.0'
