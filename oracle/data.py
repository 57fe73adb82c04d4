"""Real-image input path oracle (numpy, integer arithmetic).  TEST INFRASTRUCTURE - see oracle/__init__.py.

The reference resizes every training image on the host with torchvision ``Resize(interpolation=Image.BOX)``
(= PIL ``Image.resize``; gan_lab/data_config.py:307-331, swapped per resolution at
gan_lab/progan/learner.py:1099-1112), then ``ToTensor`` (``/255``) and ``Normalize`` (``(x-mean)/std``,
data_config.py:335-341).  PIL (pinned by the reference only as ``pillow`` in requirements.txt, any version;
checked here against 12.2.0) resamples 8-bit images in two passes - horizontal then vertical - each with 22-bit
fixed-point coefficients and round-half-up to uint8.  For a power-of-two factor f the box coefficients 2^22/f
are exact, so each pass is ``floor(sum/f + 1/2)``.  Pinned by tests/golden/data_box.npz (made with PIL)."""
import numpy as np


def box_resize_u8(images_nhwc, res):
    a = np.asarray(images_nhwc)
    assert a.dtype == np.uint8 and a.ndim == 4
    n, hs, ws, c = a.shape
    f = hs // res
    assert hs == res * f and ws == res * f and f & (f - 1) == 0, 'power-of-two box factors only'
    coef = (1 << 22) // f
    h = a.reshape(n, hs, res, f, c).astype(np.int64).sum(axis=3)
    h = (h * coef + (1 << 21)) >> 22                      # horizontal pass -> uint8
    v = h.reshape(n, res, f, res, c).sum(axis=2)
    v = (v * coef + (1 << 21)) >> 22                      # vertical pass -> uint8
    return v.astype(np.uint8)


def decode(images_nhwc, res, mean, std, flip=None):
    """(N,Hs,Ws,C) uint8 -> (N,C,res,res) float32, the arithmetic of Resize -> [flip] -> ToTensor -> Normalize."""
    u8 = box_resize_u8(images_nhwc, res)
    if flip is not None:
        u8 = np.where(np.asarray(flip, dtype=bool)[:, None, None, None], u8[:, :, ::-1, :], u8)
    x = u8.astype(np.float32) / np.float32(255.0)
    x = (x - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2)).astype(np.float32)
