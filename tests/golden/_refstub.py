"""Import shim for the upstream reference (used ONLY by make_golden.py, in the build container).

The reference (`/root/reference/gan_lab`) imports `torchvision` and `indexed`, neither of which is
installed here.  This module pre-seeds `sys.modules` with minimal stand-ins for the *symbols the
reference touches at import time* so that its pure-PyTorch hot path (custom layers, architectures,
learners) can be imported and driven as a golden-vector generator.  Nothing here restates any
reference code; the stubs only have to exist, they are never exercised by the hot path.

Never imported by the product package, by `-m gpu` tests, by `smoke()` or by `bench.py`
(`/root/reference` does not exist on the GPU box).
"""
import sys
import types
from collections import OrderedDict

REF_ROOT = '/root/reference/gan_lab'


class _Anything:
    """A do-nothing class used for every torchvision symbol the reference names."""

    def __init__(self, *a, **k):
        self.args, self.kwargs = a, k
        for key, val in k.items():
            setattr(self, key, val)

    def __call__(self, x):
        return x


class IndexedOrderedDict(OrderedDict):
    """list-returning .values()/.keys() like the `indexed` package (progan/learner.py:228,472); lives at module
    level under the package's own name so that the reference's save_model can pickle it as
    ``indexed.IndexedOrderedDict`` exactly like a real installation would."""

    def values(self):
        return list(super().values())

    def keys(self):
        return list(super().keys())


IndexedOrderedDict.__module__ = 'indexed'
IndexedOrderedDict.__qualname__ = 'IndexedOrderedDict'


def _mk(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    if 'torchvision' in sys.modules and getattr(sys.modules['torchvision'], '_ganlab_stub', False):
        return
    names = ['Resize', 'Normalize', 'Compose', 'ToPILImage', 'ToTensor', 'CenterCrop', 'RandomCrop',
             'RandomHorizontalFlip', 'Lambda', 'Pad', 'Grayscale']
    tfm = _mk('torchvision.transforms', **{n: type(n, (_Anything,), {}) for n in names})
    folder = _mk('torchvision.datasets.folder',
                 IMG_EXTENSIONS=('.jpg', '.jpeg', '.png'),
                 make_dataset=lambda *a, **k: [],
                 default_loader=lambda p: None,
                 DatasetFolder=type('DatasetFolder', (_Anything,), {}),
                 ImageFolder=type('ImageFolder', (_Anything,), {}),
                 has_file_allowed_extension=lambda *a, **k: True,
                 is_image_file=lambda *a, **k: True,
                 pil_loader=lambda p: None,
                 accimage_loader=lambda p: None)
    folder.__all__ = [k for k in folder.__dict__ if not k.startswith('_')]
    dsets = _mk('torchvision.datasets', folder=folder,
                LSUN=type('LSUN', (_Anything,), {}), CIFAR10=type('CIFAR10', (_Anything,), {}),
                DatasetFolder=folder.DatasetFolder, ImageFolder=folder.ImageFolder)
    tv = _mk('torchvision', transforms=tfm, datasets=dsets, _ganlab_stub=True)
    tv.__path__ = []

    _mk('indexed', IndexedOrderedDict=IndexedOrderedDict)


def import_reference():
    """Returns a namespace with the reference modules needed to generate golden vectors."""
    install_stubs()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import matplotlib
    matplotlib.use('Agg')
    ns = types.SimpleNamespace()
    import utils.custom_layers as cl
    import utils.initializer as ini
    import utils.backprop_utils as bp
    import utils.latent_utils as lu
    import stylegan.architectures as sa
    import stylegan.base as sb
    import progan.architectures as pa
    import progan.base as pb
    ns.cl, ns.ini, ns.bp, ns.lu, ns.sa, ns.sb, ns.pa, ns.pb = cl, ini, bp, lu, sa, sb, pa, pb
    return ns


def import_reference_learners():
    ns = import_reference()
    import resnetgan.learner as rl
    import progan.learner as pl
    import stylegan.learner as sl
    import _int
    ns.rl, ns.pl, ns.sl, ns._int = rl, pl, sl, _int
    return ns
