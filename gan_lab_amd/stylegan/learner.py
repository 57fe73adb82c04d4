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

    # -- checkpoint additions (stylegan/learner.py:455-464 on save, :547-556 / :603-607 on load) ------------------
    def _extra_checkpoint_fields(self):
        g = self.gen_model
        lag_w = None
        if self.config.use_ewma_gen:
            # the reference stores the EWMA generator's copy; that copy is a deepcopy of the generator made by the last
            # _update_gen_lagged (or at construction) - the same tensor unless training moved on since
            lag = self.gen_model_lagged
            lag_w = lag.w_ewma if (lag is not None and lag.w_ewma is not None) else g.w_ewma
        cpu = lambda v: None if v is None else v.detach().to('cpu')  # noqa: E731
        return {'use_truncation_trick': g.use_truncation_trick, 'trunc_cutoff_stage': g.trunc_cutoff_stage,
                'w_eval_psi': g.w_eval_psi, 'w_ewma_beta': g.w_ewma_beta, 'w_ewma': cpu(g.w_ewma),
                'w_ewma_lagged': cpu(lag_w), 'trained_with_noise': g._trained_with_noise,
                'pct_mixing_reg': g.pct_mixing_reg}

    def _restore_extra_fields(self, ck):
        if 'w_ewma_beta' not in ck:
            return
        g, dev = self.gen_model, self.config.dev
        g.w_ewma_beta = ck['w_ewma_beta']
        g._w_eval_psi = ck['w_eval_psi']
        g._trunc_cutoff_stage = ck['trunc_cutoff_stage']
        g.use_truncation_trick = ck['use_truncation_trick']
        g.w_ewma = None if ck['w_ewma'] is None else ck['w_ewma'].to(dev)
        g._trained_with_noise = g._use_noise = ck['trained_with_noise']
        g.pct_mixing_reg = ck['pct_mixing_reg']
        g._use_mixing_reg = True if g.pct_mixing_reg else False
        if self.config.use_ewma_gen:
            # exactly the reference's load order: the EWMA generator is a copy made while the generator still holds
            # checkpoint['w_ewma'] (:586-594), then the GENERATOR's w_ewma is overwritten with checkpoint['w_ewma_lagged']
            # (:605)
            self.materialize_lagged_generator().to(dev)
            if ck.get('w_ewma_lagged') is not None:
                g.w_ewma = ck['w_ewma_lagged'].to(dev)

    # -- style-mixing figure (stylegan/learner.py:306-431): the pixel content, without the matplotlib layout ------
    STYLE_MIX_STAGES = (1, 4, 8)    # coarse / middle / fine: generator layer at which source B's w takes over

    def make_stylemixing_grid(self, zs_sourceb, zs_coarse=(), zs_middle=(), zs_fine=(), labels=None,
                              time_average=True, save_path=None, noise=None):
        """Figure 3 of Karras et al. 2019 as a uint8 pixel grid: row 0 = source-B samples, column 0 = the
        coarse / middle / fine source-A samples, cell (r, c) = ``G(z_A[r], x_mixing=z_B[c], style_mixing_stage)``
        with stages 1 / 4 / 8 as upstream (:325-333).  Generators run in eval mode (truncation trick as
        configured); ``noise`` pins the per-layer noise (tests).  Titles / axis labels of the reference's
        matplotlib figure are not drawn; ``labels`` is accepted for signature compatibility."""
        import numpy as np
        import torch
        groups = [g if isinstance(g, torch.Tensor) else torch.empty(0, self.config.len_latent)
                  for g in (zs_coarse, zs_middle, zs_fine)]
        groups = [g.unsqueeze(0) if g.dim() == 1 else g for g in groups]
        zs_sourceb = zs_sourceb.unsqueeze(0) if zs_sourceb.dim() == 1 else zs_sourceb
        assert any(len(g) for g in groups)
        for g in [zs_sourceb] + groups:
            if g.dim() > 2:
                raise IndexError('Incorrect dimensions of input latent vector. Must be either `dim == 1` or `dim == 2`.')
            if len(g):
                self._check_sample_latents(g)
        modes = (self.gen_model.training, None if self.gen_model_lagged is None else self.gen_model_lagged.training)
        if time_average and self.gen_model_lagged is None:
            self._update_gen_lagged()
        gen = self.gen_model_lagged if time_average else self.gen_model
        gen.eval()
        try:
            kw = {} if noise is None else {'noise': noise}
            ncols = 1 + len(zs_sourceb)
            r = self.gen_model.curr_res
            rows = [[torch.ones(3, r, r, device=self.config.dev)] +
                    list(self.generate(zs_sourceb, time_average=time_average, **kw))]
            for stage, zs in zip(self.STYLE_MIX_STAGES, groups):
                for z in zs:
                    cells = [self.generate(z.unsqueeze(0), time_average=time_average, **kw)[0]]
                    for zb in zs_sourceb:
                        cells.append(self.generate(z.unsqueeze(0), time_average=time_average,
                                                   x_mixing=zb.unsqueeze(0).to(self.config.dev),
                                                   style_mixing_stage=stage, **kw)[0])
                    rows.append(cells)
        finally:
            self.gen_model.train(modes[0])
            if modes[1] is not None:
                self.gen_model_lagged.train(modes[1])
        x = torch.stack([torch.stack(c) for c in rows]).clamp_(0., 1.)           # (nrows, ncols, 3, r, r)
        grid = (x.permute(0, 3, 1, 4, 2).reshape(len(rows) * r, ncols * r, 3) * 255.).round().to(torch.uint8)
        grid = grid.cpu().numpy()
        if save_path is not None:
            from PIL import Image
            Image.fromarray(grid).save(save_path)
        return grid
