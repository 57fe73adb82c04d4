"""GPU tests of the real-image input path: uint8 NHWC -> box downsample -> fp32 NCHW.
Bit-exact (integer stage and the fp32 normalisation) against the PIL-made fixture and the numpy oracle."""
import numpy as np
import pytest
import torch

from util import load_golden

pytestmark = pytest.mark.gpu


def test_decode_matches_pil_fixture_bit_exact():
    from gan_lab_amd import ops
    g = load_golden('data_box.npz')
    imgs = torch.from_numpy(g['images']).cuda()
    for res in (64, 32, 16, 8, 4):
        x = ops.decode_u8(imgs, res, g['mean'], g['std'])
        assert x.shape == (6, 3, res, res)
        assert np.array_equal(x.cpu().numpy(), g[f'x_{res}']), res
        x2 = ops.decode_u8(imgs, res, g['mean2'], g['std2'])
        assert np.array_equal(x2.cpu().numpy(), g[f'x2_{res}']), res


@pytest.mark.parametrize('shape,res', [((5, 256, 256, 3), 32), ((3, 128, 128, 1), 128), ((2, 1024, 1024, 3), 8),
                                       ((32, 64, 64, 3), 64)])
def test_decode_matches_oracle_with_flip(shape, res):
    from gan_lab_amd import ops
    from oracle import data
    rng = np.random.default_rng(sum(shape) + res)
    imgs = rng.integers(0, 256, shape, dtype=np.uint8)
    flip = rng.integers(0, 2, shape[0]).astype(bool)
    c = shape[3]
    mean, std = rng.uniform(0.3, 0.6, c).astype(np.float32), rng.uniform(0.2, 0.6, c).astype(np.float32)
    x = ops.decode_u8(torch.from_numpy(imgs).cuda(), res, mean, std, torch.from_numpy(flip))
    assert np.array_equal(x.cpu().numpy(), data.decode(imgs, res, mean, std, flip))


def test_decode_rejects_bad_input():
    from gan_lab_amd import ops
    from gan_lab_amd._lib import GanlabLibraryError
    with pytest.raises(TypeError):
        ops.decode_u8(torch.zeros(2, 8, 8, 3, dtype=torch.uint8), 8, [0.5] * 3, [0.5] * 3)      # CPU tensor
    with pytest.raises(ValueError):
        ops.decode_u8(torch.zeros(2, 12, 12, 3, dtype=torch.uint8).cuda(), 8, [0.5] * 3, [0.5] * 3)
    with pytest.raises(GanlabLibraryError):
        ops.decode_u8(torch.zeros(2, 24, 24, 3, dtype=torch.uint8).cuda(), 8, [0.5] * 3, [0.5] * 3)   # factor 3


def test_device_loader_feeds_progressive_training():
    """The learner swaps the loader's Resize on growth; the device loader then serves that resolution."""
    from gan_lab_amd import progressive as P
    from gan_lab_amd.config import make_config
    from gan_lab_amd.progan.learner import ProGANLearner
    from gan_lab_amd.utils.data_utils import DeviceImageLoader
    P.FMAP_BASE, P.FMAP_MAX = 64, 16
    try:
        cfg = make_config('progan', dev='cuda', pin_memory=False, res_samples=16, res_dataset=64, init_res=4,
                          batch_size=4, len_latent=16, nimg_transition=24, num_iters_save_model=10 ** 9, log_every=1)
        L = ProGANLearner(cfg)
        imgs = torch.randint(0, 256, (64, 64, 64, 3), dtype=torch.uint8)
        dl = DeviceImageLoader(imgs, 4, 4, mirror=True)
        L.train(dl, num_main_iters=32)
        assert L.gen_model.curr_res == 16
        assert [r for _, r in dl.served] == sorted(r for _, r in dl.served) and dl.served[-1] == (4, 16)
        assert np.isfinite(L.last_losses['loss_d']) and np.isfinite(L.last_losses['loss_g'])
        xb, _ = next(iter(dl))
        assert xb.shape == (4, 3, 16, 16) and xb.is_cuda and xb.min() >= -1 and xb.max() <= 1
    finally:
        P.FMAP_BASE, P.FMAP_MAX = 8192, 512
