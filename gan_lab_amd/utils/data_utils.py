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
            if str(self.device) != 'cpu':     # synthetic reals drawn on the device: no host RNG / PCIe in the loop
                xb = torch.rand(bs, self.channels, res, res, device=self.device) * 2 - 1
            else:
                xb = torch.rand(bs, self.channels, res, res, generator=self.gen) * 2 - 1
            if not self.served or self.served[-1] != (bs, res):
                self.served.append((bs, res))
            yield xb, torch.zeros(bs, dtype=torch.int64)
            i += bs


class DeviceImageLoader(object):
    """``train_dl`` over a uint8 NHWC image array resident on the GPU: each batch is
    gathered, box-downsampled to the CURRENT resolution and normalised by one HIP kernel
    (``ops.decode_u8``: PIL ``Image.resize(BOX)`` -> ``ToTensor`` -> ``Normalize`` of data_config.py:307-341,
    bit-exact in the uint8 stage) instead of PIL on the host + a 4x larger fp32 PCIe copy.  Same duck type as the
    reference's DataLoader as far as ``train()`` uses it: ``len(dl.dataset)``, assignable
    ``dl.batch_sampler.batch_size``, and a ``Resize`` in ``dl.dataset.transforms.transform.transforms`` that the
    learner swaps on growth (progan/learner.py:608-612, :1099-1112)."""

    def __init__(self, images_u8_nhwc, batch_size, res, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5), mirror=False,
                 shuffle=True, seed=0, labels=None, device='cuda'):
        if images_u8_nhwc.dtype != torch.uint8 or images_u8_nhwc.dim() != 4:
            raise TypeError('images must be a (N,H,W,C) uint8 tensor')
        self.images = images_u8_nhwc.to(device)
        self.labels = labels
        self.batch_sampler = types.SimpleNamespace(batch_size=batch_size)
        tf = types.SimpleNamespace(transforms=[Resize(size=(res, res))])
        self.dataset = _Dataset(self.images.shape[0], types.SimpleNamespace(transform=tf))
        self.mean, self.std, self.mirror, self.shuffle = tuple(mean), tuple(std), mirror, shuffle
        self.gen = torch.Generator(device='cpu').manual_seed(seed)
        self.served = []

    @property
    def res(self):
        return self.dataset.transforms.transform.transforms[0].size[0]

    def __len__(self):
        return len(self.dataset) // self.batch_sampler.batch_size

    def __iter__(self):
        from .. import ops
        n = len(self.dataset)
        order = torch.randperm(n, generator=self.gen) if self.shuffle else torch.arange(n)
        i = 0
        while i + self.batch_sampler.batch_size <= n:
            bs, res = self.batch_sampler.batch_size, self.res
            idx = order[i:i + bs]
            flip = (torch.rand(bs, generator=self.gen) < 0.5) if self.mirror else None
            xb = ops.decode_u8(self.images[idx.to(self.images.device)], res, self.mean, self.std, flip)
            lab = self.labels[idx] if self.labels is not None else torch.zeros(bs, dtype=torch.int64)
            self.served.append((bs, res))
            yield xb, lab
            i += bs
