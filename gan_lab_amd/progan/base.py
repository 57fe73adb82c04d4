"""Base architecture class for ProGANs (reference: gan_lab/progan/base.py)."""
from ..progressive import ProGAN, FMAP_BASE, FMAP_MAX  # noqa: F401
