"""Synthetic ``train_dl`` for benchmarks and tests (the reference's torchvision/PIL pipeline,
gan_lab/utils/data_utils.py, is host-side I/O outside the hot path - SURVEY.md §2 rows 13-14).

``SyntheticImageLoader`` is the duck type ``ProGANLearner.train`` needs:
``len(dl.dataset)``, assignable ``dl.batch_sampler.batch_size``, a ``Resize`` in
``dl.dataset.transforms.transform.transforms`` that the learner swaps on growth, and an iterator of
``(xb, label)`` with ``xb`` float32 NCHW in [-1, 1] at the CURRENT resolution."""
import types

import torch


class Resize(object):
    """Stand-in for torchvision.transforms.Resize (only records the target size)."""

    def __init__(self, size, interpolation=None):
        self.size = size
        self.interpolation = interpolation


class _Dataset(object):
    def __init__(self, n, transforms):
        self.n = n
        self.transforms = transforms

    def __len__(self):
        return self.n


class SyntheticImageLoader(object):
    def __init__(self, num_images, batch_size, res, device='cpu', seed=0, channels=3):
        self.batch_sampler = types.SimpleNamespace(batch_size=batch_size)
        tf = types.SimpleNamespace(transforms=[Resize(size=(res, res))])
        self.dataset = _Dataset(num_images, types.SimpleNamespace(transform=tf))
        self.device = device
        self.channels = channels
        self.gen = torch.Generator(device='cpu').manual_seed(seed)
        self.served = []

    @property
    def res(self):
        return self.dataset.transforms.transform.transforms[0].size[0]

    def __iter__(self):
        i = 0
        while i + self.batch_sampler.batch_size <= len(self.dataset):
            bs, res = self.batch_sampler.batch_size, self.res
            xb = torch.rand(bs, self.channels, res, res, generator=self.gen) * 2 - 1
            self.served.append((bs, res))
            yield xb.to(self.device), torch.zeros(bs, dtype=torch.int64)
            i += bs
