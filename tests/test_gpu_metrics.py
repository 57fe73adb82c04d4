"""Validation metrics / sample grids of the learners (progan/learner.py:249-416, :1187-1234):
inference-only forwards of the HIP kernels, checked against the CPU oracle evaluating the reference's metric
definitions on the same weights and validation latents.  Tolerance 1e-3 relative (north star)."""
import numpy as np
import pytest
import torch

from util import assert_close

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.fixture(autouse=True)
def _widths():
    from gan_lab_amd import progressive as P
    P.FMAP_BASE, P.FMAP_MAX = 64, 16
    yield
    P.FMAP_BASE, P.FMAP_MAX = 8192, 512


class _DS(object):
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n


class ListLoader(object):
    """Duck type of the z_valid_dl / valid_dl DataLoaders: iterable of tuples, len(), .dataset, .batch_sampler."""

    def __init__(self, batches, n):
        self.batches, self.dataset = batches, _DS(n)
        self.batch_sampler = type('S', (), {'batch_size': len(batches[0][0])})()

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        return iter(self.batches)


def _oracle_metrics(L, zs, xs, loss):
    """The reference's definitions (progan/learner.py:286-404) on the CPU oracle."""
    from oracle import nets, ops as O
    cfg = nets.make_cfg(use_pixelnorm=True)
    sd_g = {k: v.detach().cpu().clone() for k, v in L.gen_model.state_dict().items()}
    sd_d = {k: v.detach().cpu().clone() for k, v in L.disc_model.state_dict().items()}
    fade, alpha = bool(L.gen_model.fade_in_phase), float(L.gen_model.alpha)
    n_z = sum(len(z) for z in zs)
    acc = dict(fake=0., real=0., gl=0., dl=0.)
    with torch.no_grad():
        for zb, xb in zip(zs, xs):
            k = len(zb)
            xg = nets.progen_forward(sd_g, zb, cfg, alpha=alpha, fade_in=fade)
            yf = nets.disc_forward(sd_d, xg, cfg, alpha=alpha, fade_in=fade)
            if fade:
                xb = O.upsample2(O.avgpool2(xb)) * (1. - alpha) + xb * alpha
            yr = nets.disc_forward(sd_d, xb, cfg, alpha=alpha, fade_in=fade)
            acc['fake'] += yf.sum().item()
            acc['real'] += yr.sum().item()
            acc['gl'] += O.loss_gen(loss, yf).item() * k
            acc['dl'] += O.loss_disc(loss, yf, yr).item() * k
    return {k: v / n_z for k, v in acc.items()}


@pytest.mark.parametrize('fade', [False, True], ids=['stabilised', 'fade-in'])
def test_progan_metrics_match_oracle(fade):
    from gan_lab_amd.utils.data_utils import SyntheticImageLoader
    from test_gpu_learner import make_learner
    L = make_learner('progan', 16, init_res=4, batch=4, loss='nonsaturating', gradient_penalty='r1')
    # 6 iterations per phase (nimg_transition 24 / batch 4): 4^2 stab -> 8^2 fade-in -> 8^2 stab
    L.train(SyntheticImageLoader(4096, 4, 4), num_main_iters=9 if fade else 14)
    assert L.gen_model.curr_res == 8 and bool(L.gen_model.fade_in_phase) == fade
    if fade:
        assert 0. < L.gen_model.alpha < 1.
    gen = torch.Generator().manual_seed(5)
    zs = [torch.randn(4, 16, generator=gen), torch.randn(4, 16, generator=gen), torch.randn(2, 16, generator=gen)]
    xs = [torch.rand(len(z), 3, 8, 8, generator=gen) * 2 - 1 for z in zs]
    z_dl = ListLoader([(z,) for z in zs], 10)
    x_dl = ListLoader([(x, torch.zeros(len(x))) for x in xs], 10)
    ref = _oracle_metrics(L, zs, xs, 'nonsaturating')

    lines = L.compute_metrics(['Generator Loss', 'Fake Realness'], 'Generator', z_dl)
    assert len(lines) == 2 and lines[0].strip().startswith('generator loss:')
    m = L.last_metrics['generator']
    assert_close(m['fake realness'], ref['fake'], TOL, 'fake realness')
    assert_close(m['generator loss'], ref['gl'], TOL, 'generator loss')
    L.compute_metrics(['discriminator loss', 'fake realness', 'real realness'], 'Discriminator', z_dl, x_dl)
    m = L.last_metrics['discriminator']
    assert_close(m['fake realness'], ref['fake'], TOL, 'fake realness (D)')
    assert_close(m['real realness'], ref['real'], TOL, 'real realness')
    assert_close(m['discriminator loss'], ref['dl'], TOL, 'discriminator loss')
    assert L.gen_model.training and L.disc_model.training and L.gen_metrics_num == 1 and L.disc_metrics_num == 1
    with pytest.raises(Exception, match='Invalid metrics_type'):
        L.compute_metrics(['fake realness'], 'classifier', z_dl)


def test_image_grid_and_lagged_generator_inference(tmp_path):
    """EWMA-generator inference + grid assembly: pixels equal the oracle's G_lagged(z)*std+mean, one generated
    pixel per grid pixel; the reference's argument checks raise the same exception types."""
    from gan_lab_amd.utils.data_utils import SyntheticImageLoader
    from oracle import nets
    from test_gpu_learner import make_learner
    L = make_learner('progan', 8, init_res=8, batch=4)
    L.train(SyntheticImageLoader(4096, 4, 8), num_main_iters=3)
    zs = torch.randn(4, 16, generator=torch.Generator().manual_seed(2))
    with pytest.raises(ValueError, match='mean and/or std'):
        L.make_image_grid(zs)
    L.ds_mean = torch.tensor([0.5, 0.4, 0.6]).view(3, 1, 1)
    L.ds_std = torch.tensor([0.5, 0.3, 0.2]).view(3, 1, 1)
    with pytest.raises(ValueError, match='perfect square'):
        L.make_image_grid(zs[:3])
    with pytest.raises(IndexError):
        L.make_image_grid(zs[:, :7])
    path = tmp_path / 'grid.png'
    grid = L.make_image_grid(zs, time_average=True, save_path=str(path))
    assert grid.shape == (16, 16, 3) and grid.dtype == np.uint8 and path.exists()
    lag = {k: v.detach().cpu().clone() for k, v in L.gen_model.state_dict().items()}
    lag.update({k: v.detach().cpu().clone() for k, v in L.lagged_params.items()})
    with torch.no_grad():
        ref = nets.progen_forward(lag, zs, nets.make_cfg(use_pixelnorm=True)) * L.ds_std + L.ds_mean
    ref = (ref.clamp(0, 1).view(2, 2, 3, 8, 8).permute(0, 3, 1, 4, 2).reshape(16, 16, 3) * 255).round()
    assert np.abs(grid.astype(np.int32) - ref.numpy().astype(np.int32)).max() <= 1
    snap = L.make_image_grid(zs, time_average=False)
    assert snap.shape == (16, 16, 3) and (snap != grid).any()     # snapshot generator differs from the EWMA one
    from PIL import Image
    assert np.array_equal(np.asarray(Image.open(path)), grid)


def test_train_runs_validation_like_the_reference(capsys):
    """train(train_dl, valid_dl, z_valid_dl): metrics are evaluated at iteration 0 and every num_iters_valid
    (progan/learner.py:822-832, :921-928) without disturbing the training state machine."""
    from gan_lab_amd.utils.data_utils import SyntheticImageLoader
    from test_gpu_learner import make_learner
    L = make_learner('stylegan', 8, init_res=8, batch=4, loss='nonsaturating', gradient_penalty='r1',
                     num_iters_valid=2, gen_metrics=['generator loss', 'fake realness'],
                     disc_metrics=['discriminator loss', 'fake realness', 'real realness'])
    gen = torch.Generator().manual_seed(9)
    z_dl = ListLoader([(torch.randn(4, 16, generator=gen),) for _ in range(2)], 8)
    x_dl = ListLoader([(torch.rand(4, 3, 8, 8, generator=gen) * 2 - 1, torch.zeros(4)) for _ in range(2)], 8)
    L.train(SyntheticImageLoader(4096, 4, 8), valid_dl=x_dl, z_valid_dl=z_dl, num_main_iters=4)
    out = capsys.readouterr().out
    assert out.count('Generator Validation Metrics') == 3 and out.count('Discriminator Validation Metrics') == 3
    assert L.gen_metrics_num == 3 and L.disc_metrics_num == 3
    assert all(np.isfinite(v) for m in L.last_metrics.values() for v in m.values())
    assert L.gen_model.training and L.disc_model.training


def test_stylemixing_grid_cells_are_mixed_samples():
    """make_stylemixing_grid (stylegan/learner.py:306-431, pixel content): cell (r, c) is the eval-mode generator
    with source B's style taking over at stage 1 / 4 - compared with the oracle's style-mixing forward."""
    from gan_lab_amd.utils.data_utils import SyntheticImageLoader
    from oracle import nets
    from test_gpu_learner import make_learner
    L = make_learner('stylegan', 32, init_res=32, batch=4, loss='nonsaturating', gradient_penalty='r1',
                     use_ewma_gen=False)
    L.train(SyntheticImageLoader(4096, 4, 32), num_main_iters=2)
    L.ds_mean, L.ds_std = torch.full((3, 1, 1), 0.5), torch.full((3, 1, 1), 0.5)
    gen = torch.Generator().manual_seed(4)
    zb, za = torch.randn(2, 16, generator=gen), torch.randn(2, 16, generator=gen)
    nl = len(L.gen_model.gen_layers)
    noise = [torch.randn(1, 1, 4 * 2 ** (n // 2), 4 * 2 ** (n // 2), generator=gen) for n in range(nl)]
    grid = L.make_stylemixing_grid(zb, zs_coarse=za[:1], zs_middle=za[1:], time_average=False,
                                   noise=[n.cuda() for n in noise])
    assert grid.shape == (3 * 32, 3 * 32, 3) and L.gen_model.training
    sd = {k: v.detach().cpu().clone() for k, v in L.gen_model.state_dict().items()}
    cfg = nets.make_cfg()

    def cell(r, c):
        return torch.from_numpy(grid[r * 32:(r + 1) * 32, c * 32:(c + 1) * 32].astype(np.float32)).permute(2, 0, 1)
    with torch.no_grad():
        for r, (z, stage) in enumerate(((za[:1], 1), (za[1:], 4)), start=1):
            for c in (1, 2):
                ref = nets.stylegen_forward(sd, z, noise, cfg, z_mix=zb[c - 1:c], cutoff_idx=stage)
                ref = ((ref[0] * 0.5 + 0.5).clamp(0, 1) * 255).round()
                assert (cell(r, c) - ref).abs().max() <= 1, (r, c)


@pytest.mark.parametrize('kind', ['stylegan', 'progan'])
def test_graphed_generator_matches_eager(kind, capsys):
    """HIP-graph replay of the eval-mode generator (gan_lab_amd/graphs.py): bit-identical to the eager forward for
    the same latents / noise, follows in-place weight updates, and is faster per call at batch 1."""
    import time
    from gan_lab_amd.graphs import GraphedGenerator
    from gan_lab_amd.utils.data_utils import SyntheticImageLoader
    from test_gpu_learner import make_learner
    kw = dict(loss='nonsaturating', gradient_penalty='r1') if kind == 'stylegan' else {}
    L = make_learner(kind, 32, init_res=32, batch=4, **kw)
    dl = SyntheticImageLoader(4096, 4, 32)
    L.train(dl, num_main_iters=2)
    g = L.gen_model
    g.eval()
    z0 = torch.randn(2, 16).cuda()
    n0 = [torch.randn(2, 1, 4 * 2 ** (n // 2), 4 * 2 ** (n // 2)).cuda() for n in range(len(g.gen_layers))] \
        if kind == 'stylegan' else None
    with torch.no_grad():
        before = (g(z0, noise=n0) if n0 is not None else g(z0)).clone()
    gg = GraphedGenerator(g, batch=1)
    # an EAGER forward between the capture and the first replay must not pick up the packed-weight tensors the capture
    # left in the cache (graph-pool memory no kernel has written yet)
    with torch.no_grad():
        after = g(z0, noise=n0) if n0 is not None else g(z0)
    assert torch.equal(before, after)
    gen = torch.Generator().manual_seed(0)
    for it in range(3):
        z = torch.randn(1, 16, generator=gen).cuda()
        noise = None
        if gg.noise is not None:
            noise = [torch.randn(1, 1, *b.shape[2:], generator=gen).cuda() for b in gg.noise]
        out = gg(z, noise=noise).clone()
        with torch.no_grad():
            ref = g(z, noise=noise) if noise is not None else g(z)
        assert torch.equal(out, ref), (kind, it, (out - ref).abs().max().item())
        if it == 0:                       # one more training iteration rewrites the weights in place
            g.train()
            L.train(dl, num_main_iters=1)
            g.eval()

    def per_call(fn, n=30):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    z = torch.randn(1, 16).cuda()
    t_graph = per_call(lambda: gg(z, redraw_noise=False))
    with torch.no_grad():
        t_eager = per_call(lambda: g(z, noise=gg.noise) if gg.noise is not None else g(z))
    with capsys.disabled():
        print(f'\n{kind} 32^2 batch-1 sample: eager {t_eager:.3f} ms, hipGraph replay {t_graph:.3f} ms')
    assert t_graph < t_eager
    g.train()
