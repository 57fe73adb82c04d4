"""Mirror of the reference sub-package gan_lab/stylegan."""
