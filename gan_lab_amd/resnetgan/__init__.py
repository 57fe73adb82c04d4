"""Mirror of the reference sub-package gan_lab/resnetgan (base learner only; the ResNet GAN
architectures, BASELINE config #5, are a later hot-path row)."""
