"""StyleGANLearner on the HIP path (drop-in surface of gan_lab/stylegan/learner.py:97-236).
Inherits ``train`` unchanged from ProGANLearner, like the reference does; only the network
construction differs (StyleGenerator + the ProDiscriminator body in the StyleGAN family)."""
from ..progan.architectures import StyleDiscriminator
from ..progan.learner import ProGANLearner, REDEFINABLE_FROM_LEARNER_ATTRS  # noqa: F401
from .architectures import StyleGenerator
from .base import StyleGAN

NONREDEFINABLE_ATTRS = ('model', 'init_res', 'res_samples', 'res_dataset', 'len_latent', 'num_classes',
                        'class_condition', 'use_auxiliary_classifier', 'model_upsample_type',
                        'model_downsample_type', 'align_corners', 'blur_type', 'nonlinearity', 'use_equalized_lr',
                        'normalize_z', 'use_pixelnorm', 'mbstd_group_size', 'use_ewma_gen', 'use_instancenorm',
                        'use_noise', 'pct_mixing_reg', 'beta_trunc_trick', 'psi_trunc_trick', 'cutoff_trunc_trick',
                        'len_dlatent', 'mapping_num_fcs', 'mapping_lrmul',)


class StyleGANLearner(ProGANLearner):
    """GAN learner for StyleGAN architectures."""
    _family = StyleGAN
    _nonredefinable = NONREDEFINABLE_ATTRS

    def __init__(self, config):
        super().__init__(config)
        if self.model == 'StyleGAN':
            self._init_progressive(config, self.__class__.__name__)

    def _build_networks(self):
        c = self.config
        gen = StyleGenerator(
            final_res=c.res_samples, latent_distribution=c.latent_distribution, len_latent=c.len_latent,
            len_dlatent=c.len_dlatent, mapping_num_fcs=c.mapping_num_fcs, mapping_lrmul=c.mapping_lrmul,
            use_instancenorm=c.use_instancenorm, use_noise=c.use_noise, upsampler=self.gen_model_upsampler,
            blur_type=c.blur_type, nl=self.nl, num_classes=self.num_classes_gen, equalized_lr=c.use_equalized_lr,
            normalize_z=c.normalize_z, use_pixelnorm=c.use_pixelnorm, pct_mixing_reg=c.pct_mixing_reg,
            truncation_trick_params={'beta': c.beta_trunc_trick, 'psi': c.psi_trunc_trick,
                                     'cutoff_stage': c.cutoff_trunc_trick})
        disc = StyleDiscriminator(final_res=c.res_samples, pooler=self.disc_model_downsampler, blur_type=c.blur_type,
                                  nl=self.nl, num_classes=self.num_classes_disc, equalized_lr=c.use_equalized_lr,
                                  mbstd_group_size=c.mbstd_group_size)
        return gen, disc
