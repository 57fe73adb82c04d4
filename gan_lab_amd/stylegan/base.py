"""Base architecture class for StyleGANs (reference: gan_lab/stylegan/base.py)."""
from ..progressive import StyleGAN, FMAP_BASE, FMAP_MAX  # noqa: F401
