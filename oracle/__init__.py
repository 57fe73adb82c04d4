"""CPU oracle for the gan-lab G+D training-step hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch (CPU, fp32) *restatement* of the reference's algorithm for the hot
path: equalised-LR convs/linears, binomial blur, PixelNorm,
InstanceNorm + AdaIN, noise injection, minibatch-stddev, the StyleGAN / ProGAN generator and
discriminator graphs driven from a reference-layout ``state_dict``, the GAN losses, the R1 / R2 /
WGAN-GP gradient penalties (double backward through ``torch.autograd``) and the Adam + EWMA update.
Every function cites the reference file:line it follows (paths relative to /root/reference/gan_lab).

It is pinned against golden vectors captured from the imported reference itself
(``tests/golden/*.npz`` written by ``tests/golden/make_golden.py``; checked in
``tests/test_oracle_golden.py``) - the reference has no tests or fixtures of its own.

Allowed importers: ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg,
as the checker / baseline only.  The product package ``gan_lab_amd`` never imports it, and has no
CPU fallback: its ops raise when the HIP library is missing.

There is no C restatement here: the reference's arithmetic *is* PyTorch ATen fp32 on CPU, so the
oracle keeps a torch fp32 reference (floating-point kernels), as the task's tier rules allow.
"""
from . import ops, nets, step  # noqa: F401
